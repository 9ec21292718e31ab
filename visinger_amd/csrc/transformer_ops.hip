// transformer_ops.hip -- the non-conv kernels of the windowed relative-position transformer on gfx950:
//   vs_relattn_fwd       MultiHeadAttention.attention      (reference modules/rel_transformer.py:148-179 + 181-243)
//   vs_layernorm_c_fwd   LayerNorm over channels (+ fused residual add / conditioning add / mask)
//                                                         (rel_transformer.py:24-42, 305-307, 314-316, 297-299)
//
// Attention is a streaming-softmax ("flash") kernel on the exact-fp32 matrix instruction: the [T, T] score
// tensor, its three padded/skewed copies (rel_transformer.py:214-243) and the stored `self.attn` never exist.
// The relative-key bias and the relative-value term are non-zero only for |i - j| <= window, so they are computed
// by index arithmetic: QR[i][r] = q_i . rel_k[r] (9 dots per query) is added on the diagonal tiles, and the 9
// in-window scores of each query are kept aside so that sum_r p[i, i+r] * rel_v[r] is added once at the end.
//
// Orientation: the kernel computes S^T = K^T Q (keys on the MFMA rows, queries on the lanes), so a query's row
// statistics are in-register reductions plus one cross-half exchange, and the probability tile is -- register for
// register -- the B operand of the P.V product (k-index order chosen to match the accumulator layout): no LDS
// round trip, no shuffles between the two GEMMs.
#include "attn_common.h"

#include <cstdlib>

extern unsigned long long *g_stamp_buf;   // conv_engine.hip (vs_debug_set_stamp_buffer)

namespace vs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// DT = number of 32-wide tiles of the head dimension (dk <= 32*DT).
// One workgroup = 4 waves = 4 query tiles of 32 rows of one (batch, head); the K/V tiles of 32 keys are shared through
// LDS, double-buffered: the global loads of tile t+1 are issued before the MFMAs of tile t and written to the other
// buffer after them (one barrier per tile).  The pre-scaled query tile lives in registers (it is the B operand of every
// S^T MFMA), which keeps the LDS footprint at 74 KB for dk = 96 -> two workgroups per CU.
//
// WIDE (heads of 129..256 channels, BASELINE config 5: hidden 512, 2 heads): one workgroup per CU, so that a wave may use the
// whole 512-register budget (128 for the query tile, 128 for the output accumulators, 64 for the K/V tile in flight); the K/V
// tile is single-buffered in LDS (double-buffered it would need 167 KB) with the next tile prefetched in registers.
template <int DT, int NWV, bool WIDE>
__global__ void __launch_bounds__(64 * NWV, WIDE ? 1 : 2) relattn_kernel(const AttnParams p) {
    constexpr int DKR = DT * 32;                   // padded head dim (rows of the K / V tiles in LDS)
    constexpr int NTHR = 64 * NWV;
    constexpr int KPT = DKR * 32 / NTHR;           // K (and V) tile elements staged per thread
    constexpr int NBUF = WIDE ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int i0 = (blockIdx.x * NWV + wave) * 32;
    const int dk = p.dk, T = p.T;
    const int nrel = (p.ws >= 0 && p.rel_k) ? 2 * p.ws + 1 : 0;

    float *Ks = smem;                              // [NBUF][DKR][32]
    float *Vs = Ks + NBUF * DKR * 32;              // [NBUF][DKR][33]
    float *Ms = Vs + NBUF * DKR * 33;              // [2][32]   key mask of the tile
    float *QRs = Ms + 64;                          // [NWV][32][ATT_QRS]   rel-key logits
    float *Sws = QRs + NWV * 32 * ATT_QRS;         // [NWV][32][ATT_QRS]   in-window raw scores
    float *RVs = Sws + NWV * 32 * ATT_QRS;         // [ATT_MAXREL][dk] relative value embeddings


    const float *qb = p.q + (long long)b * p.bs + (long long)h * dk * T;
    const float *kb = p.k + (long long)b * p.bs + (long long)h * dk * T;
    const float *vb = p.v + (long long)b * p.bs + (long long)h * dk * T;
    const float *maskb = p.mask ? p.mask + (long long)b * T : nullptr;
    const float *relk = nrel ? p.rel_k + (long long)(p.nh_rel == 1 ? 0 : h) * nrel * dk : nullptr;
    const float *relv = nrel ? p.rel_v + (long long)(p.nh_rel == 1 ? 0 : h) * nrel * dk : nullptr;

    // ---- this lane's slice of the query tile, pre-scaled: B operand of S^T = K^T Q is Q[d = 2kk+half][i = l31] ----
    const int qi = i0 + l31;                       // this lane's query
    const int qic = min(qi, T - 1);
    float qreg[DT * 16];
#pragma unroll
    for (int kk = 0; kk < DT * 16; ++kk) {
        const int d = 2 * kk + half;
        const float v = qb[(long long)min(d, dk - 1) * T + qic];
        qreg[kk] = (d < dk && qi < T) ? v * p.scale : 0.f;
    }
    auto qv = [&](int kk) __attribute__((always_inline)) { return qreg[kk]; };
    for (int e = tid; e < nrel * dk; e += NTHR) RVs[e] = relv[e];
    float *QRw = QRs + wave * 32 * ATT_QRS;
    float *Sww = Sws + wave * 32 * ATT_QRS;
    for (int e = lane; e < 32 * ATT_QRS; e += 64) Sww[e] = -INFINITY;
    if (nrel) {
        // QR[i][r] = (q_i * scale) . rel_k[r]: each lane holds half of the d's -> partial dots, summed across halves
        float qr[ATT_MAXREL];
#pragma unroll
        for (int r = 0; r < ATT_MAXREL; ++r) qr[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < DT * 16; ++kk) {
            const int d = min(2 * kk + half, dk - 1);      // qreg is 0 beyond dk
#pragma unroll
            for (int r = 0; r < ATT_MAXREL; ++r)
                if (r < nrel) qr[r] += qv(kk) * relk[r * dk + d];
        }
#pragma unroll
        for (int r = 0; r < ATT_MAXREL; ++r) {
            const float tot = qr[r] + __shfl_xor(qr[r], 32);
            if (half == 0) QRw[l31 * ATT_QRS + r] = tot;
        }
    }

    // ---- K/V tile staging (registers -> LDS), thread t owns elements e = t + 256*i of the [DKR][32] tile ----
    float kst[KPT], vst[KPT], mst = 1.f;
    auto tile_load = [&](int jt) __attribute__((always_inline)) {
        const int j0 = jt * 32;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int e = tid + NTHR * i;
            const int d = e >> 5, jj = e & 31;
            const long long off = (long long)min(d, dk - 1) * T + min(j0 + jj, T - 1);
            kst[i] = kb[off];
            vst[i] = vb[off];
        }
        if (tid < 32) mst = maskb ? maskb[min(j0 + tid, T - 1)] : 1.f;
    };
    auto tile_store = [&](int jt, int buf) __attribute__((always_inline)) {
        const int j0 = jt * 32;
        float *Kb = Ks + buf * DKR * 32, *Vb = Vs + buf * DKR * 33;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int e = tid + NTHR * i;
            const int d = e >> 5, jj = e & 31;
            const bool ok = (d < dk) && (j0 + jj < T);
            Kb[e] = ok ? kst[i] : 0.f;
            Vb[d * 33 + jj] = ok ? vst[i] : 0.f;
        }
        if (tid < 32) Ms[buf * 32 + tid] = (j0 + tid < T) ? mst : 1.f;
    };

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_half = 0.f;
    const float mi = (maskb && qi < T) ? maskb[qi] : 1.f;

    const int ntiles = (T + 31) / 32;
    tile_load(0);
    tile_store(0, 0);
    __syncthreads();
    for (int jt = 0; jt < ntiles; ++jt) {
        const int j0 = jt * 32;
        const int buf = WIDE ? 0 : (jt & 1);
        const float *Kb = Ks + buf * DKR * 32, *Vb = Vs + buf * DKR * 33, *Mb = Ms + buf * 32;
        if (jt + 1 < ntiles) tile_load(jt + 1);

        // ---- S^T tile: rows = keys, cols (lanes) = queries ----
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < DT * 16; ++kk) {
            const float a = Kb[(2 * kk + half) * 32 + l31];
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qv(kk), s, 0, 0, 0);
        }
        const bool near_diag = nrel && (j0 + 31 >= i0 - p.ws) && (j0 <= i0 + 31 + p.ws);
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = acc_row(r, half);
            const int j = j0 + jj;
            float sv = s[r];
            if (near_diag) {
                const int rel = j - qi;
                if (rel >= -p.ws && rel <= p.ws) sv += QRw[l31 * ATT_QRS + rel + p.ws];
            }
            if (mi * Mb[jj] == 0.f) sv = -1e4f;          // masked_fill(mask == 0, -1e4)
            if (j >= T) sv = -INFINITY;                  // beyond the sequence: not part of the softmax
            if (near_diag) {
                const int rel = j - qi;
                if (rel >= -p.ws && rel <= p.ws && j < T) Sww[l31 * ATT_QRS + rel + p.ws] = sv;
            }
            s[r] = sv;
            tmax = fmaxf(tmax, sv);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = expf(s[r] - m_new);     // exp(-inf) = 0 for excluded keys
            s[r] = pv;
            psum += pv;
        }
        l_half = l_half * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

        // ---- O^T += V P^T : k-step ks sums over key acc_row(ks, half), i.e. exactly the key held in s[ks] ----
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int jj = acc_row(ks, half);
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const float a = Vb[(t * 32 + l31) * 33 + jj];
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[ks], o[t], 0, 0, 0);
            }
        }
        if constexpr (WIDE) {
            __syncthreads();                         // every wave is done with the (single) K/V buffer
            if (jt + 1 < ntiles) tile_store(jt + 1, 0);
            __syncthreads();
        } else {
            if (jt + 1 < ntiles) tile_store(jt + 1, buf ^ 1);
            __syncthreads();
        }
    }

    // ---- finish: normalise, add the relative-value term, store ----
    const float l_tot = l_half + __shfl_xor(l_half, 32);
    const float inv = 1.0f / l_tot;
    float pw[ATT_MAXREL];
#pragma unroll
    for (int r = 0; r < ATT_MAXREL; ++r) pw[r] = (r < nrel) ? expf(Sww[l31 * ATT_QRS + r] - m_run) * inv : 0.f;
    float *ob = p.out + (long long)b * p.out_bs + (long long)h * dk * T;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = t * 32 + acc_row(r, half);
            float val = o[t][r] * inv;
            if (d < dk) {
#pragma unroll
                for (int rr = 0; rr < ATT_MAXREL; ++rr)
                    if (rr < nrel) val += pw[rr] * RVs[rr * dk + d];
                if (qi < T) ob[(long long)d * T + qi] = val;
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------------------------
// y[b,c,t] = ((LN_c(a + r) * gamma + beta) + g[b,c,t*g_ts]) * mask[b,t]
// block = G channel groups x 64 frames; each thread keeps its C/G channel values in registers
struct LnParams {
    const float *a, *r;       // r may be null
    const float *gamma, *beta;
    const float *g;           // optional post-add (conditioning of the NEXT layer), time stride g_ts (0 or 1)
    long long g_bs;
    int g_ts;
    const float *mask;        // optional [B, T]
    float *y;
    int B, C, T;
    float eps;
};

template <int G, int PT>
__global__ void __launch_bounds__(64 * G) layernorm_c_kernel(const LnParams p) {
    __shared__ float red[G][64];
    const int tl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + tl;
    const int tc = min(t, p.T - 1);
    const float *ab = p.a + (long long)b * p.C * p.T + tc;
    const float *rb = p.r ? p.r + (long long)b * p.C * p.T + tc : nullptr;
    float v[PT];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int c = grp + G * i;
        const int cc = min(c, p.C - 1);
        float x = ab[(long long)cc * p.T];
        if (rb) x += rb[(long long)cc * p.T];
        v[i] = (c < p.C) ? x : 0.f;
        sum += v[i];
    }
    red[grp][tl] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int gI = 0; gI < G; ++gI) tot += red[gI][tl];
    const float mean = tot / (float)p.C;
    __syncthreads();
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int c = grp + G * i;
        const float d = (c < p.C) ? v[i] - mean : 0.f;
        sq += d * d;
    }
    red[grp][tl] = sq;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int gI = 0; gI < G; ++gI) var += red[gI][tl];
    var /= (float)p.C;
    const float rs = rsqrtf(var + p.eps);
    const float mval = p.mask ? p.mask[(long long)b * p.T + tc] : 1.f;
    if (t < p.T) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int c = grp + G * i;
            if (c < p.C) {
                float o = (v[i] - mean) * rs * p.gamma[c] + p.beta[c];
                if (p.g) o += p.g[(long long)b * p.g_bs + (long long)c * (p.g_ts ? p.T : 1) + (p.g_ts ? t : 0)];
                p.y[((long long)b * p.C + c) * p.T + t] = o * mval;
            }
        }
    }
}

}  // namespace vs

using namespace vs;

template <int DT, int NWV, bool WIDE>
static int launch_attn(const AttnParams &p, hipStream_t s) {
    constexpr int NBUF = WIDE ? 1 : 2;
    const size_t lds = sizeof(float) * ((size_t)NBUF * DT * 32 * 32 + (size_t)NBUF * DT * 32 * 33 + 64 + 2 * NWV * 32 * ATT_QRS +
                                        (size_t)ATT_MAXREL * DT * 32);
    auto kern = relattn_kernel<DT, NWV, WIDE>;
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (lds > 160 * 1024) { set_error("vs_relattn_fwd: head dim %d needs %zu B of LDS", p.dk, lds); return VS_EUNSUPPORTED; }
    dim3 grid((unsigned)ceil_div(p.T, 32 * NWV), (unsigned)p.nh, (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NWV), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    set_last_kernel("relattn_kernel<%d, %d, %s>", DT, NWV, WIDE ? "true" : "false");
    return VS_OK;
}

extern "C" {

static int relattn_fwd_impl(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k,
                            const float *rel_v, const float *mask, float *out, int64_t out_batch_stride, int64_t B, int n_heads,
                            int k_channels, int64_t T, int window_size, int n_heads_rel, int math, float *work, int ksplit,
                            void *kv_work, size_t kv_work_bytes, void *stream);

int vs_relattn_fwd(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k,
                   const float *rel_v, const float *mask, float *out, int64_t out_batch_stride, int64_t B, int n_heads,
                   int k_channels, int64_t T, int window_size, int n_heads_rel, int math, void *stream) {
    return relattn_fwd_impl(q, k, v, qkv_batch_stride, rel_k, rel_v, mask, out, out_batch_stride, B, n_heads, k_channels, T, window_size,
                            n_heads_rel, math, nullptr, 0, nullptr, 0, stream);
}

int vs_relattn_fwd_ksplit(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k,
                          const float *rel_v, const float *mask, float *out, int64_t out_batch_stride, int64_t B, int n_heads,
                          int k_channels, int64_t T, int window_size, int n_heads_rel, int math, float *work, int ksplit, void *stream) {
    VS_REQUIRE(ksplit >= 1 && ksplit <= 16 && (ksplit == 1 || work), "vs_relattn_fwd_ksplit: ksplit in 1..16, work buffer for ksplit > 1");
    return relattn_fwd_impl(q, k, v, qkv_batch_stride, rel_k, rel_v, mask, out, out_batch_stride, B, n_heads, k_channels, T, window_size,
                            n_heads_rel, math, work, ksplit, nullptr, 0, stream);
}

size_t vs_relattn_kv_work_bytes(int64_t B, int n_heads, int k_channels, int64_t T, int math) {
    if (math != VS_MATH_BF16 || B <= 0 || n_heads <= 0 || k_channels <= 0 || T <= 0 || opt(OPT_NO_BF16_ATTN)) return 0;
    return attn_kv_work_bytes(B, n_heads, k_channels, T);
}

int vs_relattn_fwd_work(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k,
                        const float *rel_v, const float *mask, float *out, int64_t out_batch_stride, int64_t B, int n_heads,
                        int k_channels, int64_t T, int window_size, int n_heads_rel, int math, float *work, int ksplit,
                        void *kv_work, size_t kv_work_bytes, void *stream) {
    VS_REQUIRE(ksplit >= 1 && ksplit <= 16 && (ksplit == 1 || work), "vs_relattn_fwd_work: ksplit in 1..16, work buffer for ksplit > 1");
    VS_REQUIRE(!kv_work || (reinterpret_cast<uintptr_t>(kv_work) & 15u) == 0, "vs_relattn_fwd_work: kv_work must be 16-byte aligned");
    return relattn_fwd_impl(q, k, v, qkv_batch_stride, rel_k, rel_v, mask, out, out_batch_stride, B, n_heads, k_channels, T, window_size,
                            n_heads_rel, math, work, ksplit, kv_work, kv_work_bytes, stream);
}

static int relattn_fwd_impl(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k,
                            const float *rel_v, const float *mask, float *out, int64_t out_batch_stride, int64_t B, int n_heads,
                            int k_channels, int64_t T, int window_size, int n_heads_rel, int math, float *work, int ksplit,
                            void *kv_work, size_t kv_work_bytes, void *stream) {
    VS_REQUIRE(q && k && v && out, "vs_relattn_fwd: NULL tensor");
    VS_REQUIRE(B > 0 && B <= 65535 && n_heads > 0 && k_channels > 0 && T > 0, "vs_relattn_fwd: bad dims");
    VS_REQUIRE(window_size < 0 || (rel_k && rel_v), "vs_relattn_fwd: window given but relative embeddings are NULL");
    VS_REQUIRE(window_size < 0 || 2 * window_size + 1 <= ATT_MAXREL, "vs_relattn_fwd: window_size %d too large", window_size);
    VS_REQUIRE(n_heads_rel == 1 || n_heads_rel == n_heads, "vs_relattn_fwd: bad n_heads_rel");
    AttnParams p;
    p.q = q; p.k = k; p.v = v;
    p.bs = qkv_batch_stride ? qkv_batch_stride : (long long)n_heads * k_channels * T;
    p.rel_k = window_size >= 0 ? rel_k : nullptr;
    p.rel_v = window_size >= 0 ? rel_v : nullptr;
    p.mask = mask; p.out = out;
    p.out_bs = out_batch_stride ? out_batch_stride : (long long)n_heads * k_channels * T;
    p.B = (int)B; p.nh = n_heads; p.dk = k_channels; p.T = (int)T; p.ws = window_size; p.nh_rel = n_heads_rel;
    p.scale = 1.0f / sqrtf((float)k_channels);
    p.part = nullptr; p.ksplit = 0; p.kvimg = nullptr; p.stamps = g_stamp_buf;
    hipStream_t s = as_stream(stream);
    VS_REQUIRE(math == VS_MATH_F32 || math == VS_MATH_BF16 || math == VS_MATH_SPLIT6 || math == VS_MATH_SPLIT3, "vs_relattn_fwd: unknown arithmetic %d", math);
    // VS_MATH_SPLIT3 (the default arithmetic of the path): q / sqrt(dk), k, v and the probabilities as two f16 planes under power-of-two scales
    // (per query, per K tile, running per V tile), three cross products per product on the f16 matrix instruction -- fp32-class like
    // VS_MATH_SPLIT6 at half the executed matrix work (round 6; heads of up to 128 channels, else the split-bf16 x6 / fp32 kernels below)
    if (math == VS_MATH_SPLIT3) {
        if (work && ksplit > 1 && T / 64 >= ksplit) { p.part = work; p.ksplit = ksplit; }
        if (attn_bf16_supported(p, 3) && !opt(OPT_ATTN_SPLIT6) && !opt(OPT_NO_SPLIT_ATTN)) return launch_attn_bf16(p, 3, s);
        p.part = nullptr; p.ksplit = 0;
        math = VS_MATH_SPLIT6;
    }
    // VS_MATH_BF16: both GEMMs on the bf16 matrix instruction (attention_bf16.hip); any other arithmetic, and shapes that kernel does
    // not take (T % 4 != 0, unaligned rows), run the exact-fp32 MFMA kernel below
    // VS_MATH_SPLIT6 (the default of rounds 3-5, selectable): the same kernel with every operand split exactly into three bf16 planes and
    // six cross products per product -- fp32-class scores and outputs at 16/6 of the fp32 matrix rate; VS_MATH_F32: the kernel below
    // (key split: only the bf16-pipe kernels take it; the number of ranges is capped by the key tiles of the kernel, 32 keys each at least)
    if (work && ksplit > 1 && T / 64 >= ksplit) { p.part = work; p.ksplit = ksplit; }
    if (math == VS_MATH_BF16 && attn_bf16_supported(p, 1) && !opt(OPT_NO_BF16_ATTN)) {
        // pre-packed K / V tile images when the caller brought the scratch for them (vs_relattn_kv_work_bytes)
        const size_t need = attn_kv_work_bytes(B, n_heads, k_channels, T);
        if (kv_work && need && kv_work_bytes >= need) p.kvimg = static_cast<unsigned *>(kv_work);
        return launch_attn_bf16(p, 1, s);
    }
    if (math == VS_MATH_SPLIT6 && attn_bf16_supported(p, 6) && !opt(OPT_NO_SPLIT_ATTN)) return launch_attn_bf16(p, 6, s);
    p.part = nullptr; p.ksplit = 0;
    const int DT = (int)ceil_div(k_channels, 32);
    switch (DT) {
        case 1: return launch_attn<1, 4, false>(p, s);
        case 2: return launch_attn<2, 4, false>(p, s);
        case 3: return launch_attn<3, 4, false>(p, s);
        case 4: return launch_attn<4, 4, false>(p, s);
        case 5: return launch_attn<5, 4, true>(p, s);
        case 6: return launch_attn<6, 4, true>(p, s);
        case 7: return launch_attn<7, 4, true>(p, s);
        case 8: return launch_attn<8, 4, true>(p, s);
        default: set_error("vs_relattn_fwd: k_channels %d > 256 not supported", k_channels); return VS_EUNSUPPORTED;
    }
}

int vs_layernorm_c_fwd(const float *a, const float *r, const float *gamma, const float *beta, const float *g,
                       int64_t g_batch_stride, int g_time_stride, const float *mask, float *y, int64_t B, int64_t C,
                       int64_t T, float eps, void *stream) {
    VS_REQUIRE(a && gamma && beta && y && B > 0 && B <= 65535 && C > 0 && T > 0, "vs_layernorm_c_fwd: bad arguments");
    LnParams p;
    p.a = a; p.r = r; p.gamma = gamma; p.beta = beta; p.g = g; p.g_bs = g_batch_stride; p.g_ts = g_time_stride;
    p.mask = mask; p.y = y; p.B = (int)B; p.C = (int)C; p.T = (int)T; p.eps = eps;
    hipStream_t s = as_stream(stream);
    dim3 grid((unsigned)ceil_div(T, 64), (unsigned)B);
    if (C <= 4 * 16) hipLaunchKernelGGL((layernorm_c_kernel<4, 16>), grid, dim3(256), 0, s, p);
    else if (C <= 16 * 16) hipLaunchKernelGGL((layernorm_c_kernel<16, 16>), grid, dim3(1024), 0, s, p);     // (16 groups of <= 16 channels: 4x the threads of <4, 64> per 64-frame block)
    else if (C <= 16 * 32) hipLaunchKernelGGL((layernorm_c_kernel<16, 32>), grid, dim3(1024), 0, s, p);
    else if (C <= 16 * 64) hipLaunchKernelGGL((layernorm_c_kernel<16, 64>), grid, dim3(1024), 0, s, p);
    else { set_error("vs_layernorm_c_fwd: C=%lld > 1024 unsupported", (long long)C); return VS_EUNSUPPORTED; }
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

}  // extern "C"
