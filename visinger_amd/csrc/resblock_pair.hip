// resblock_pair.hip -- one launch for a whole residual pair of the HiFi-GAN MRF blocks on the narrow stages (32 / 64 channels):
//
//     y = conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 + x   [+ acc] [* scale]          (reference decoder.py:92-101: xt = lrelu(x);
//                                                                                  xt = c1(xt); xt = lrelu(xt); xt = c2(xt); x = xt + x)
//
// At C = 32 the two convs of a pair are HBM-bound as separate launches (k=3: 16 FLOP/B, measured 4.2 TB/s): x is read twice (input
// and residual), the intermediate is written and read back -- 5 tensor passes of 1.07 GB.  Here the intermediate never leaves
// the CU: a workgroup computes conv1 on a time tile widened by conv2's halo (the same exact-fp32 MFMA main loop as
// conv_engine.hip, x staged per 16-channel chunk through LDS), applies bias + leaky-relu in registers, lays the tile out in LDS as
// the B operand of conv2 ([channel][time], zero outside the sequence: conv2's own zero padding), and runs conv2 straight from
// there -- no staging, no barrier in the second loop.  2 passes instead of 5, one prologue / epilogue instead of two.
//
// Tile: C rows x BN columns of the intermediate (BN = 512 at C = 32: four waves side by side; 256 at C = 64: 2 x 2), of which
// NOUT = BN - 12 are final outputs (k <= 13); the staging buffers are reused for the intermediate tile, so the LDS footprint
// stays at that of the plain conv (73 KB -> two workgroups per CU).
#include "vs_internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace vs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PCK = 16;      // input channels per LDS chunk (as the conv engine: the packed weight layout depends on it)
constexpr int PHALO = 12;    // intermediate columns that are not final outputs (>= k - 1, multiple of 4)

struct PairParams {
    const float *x;
    long long x_bs;
    const float *wp1, *bias1, *wp2, *bias2;   // packed by conv_engine.hip (Wp[m_tile][tap][chunk][quad][64][4], biasp[row])
    float *y;
    const float *res, *acc;
    long long y_bs, res_bs, acc_bs;
    float scale;
    int B, C, T, K, d1, nchunks;
    int W1;          // staged x columns: BN + (K - 1) * d1
    int fast_epi;    // T % 4 == 0 and 16-byte aligned tensors
};

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(256, 2) respair_kernel(const PairParams p) {
    constexpr int NW = WAVES_M * WAVES_N;          // 4
    constexpr int NT_W = 4;
    constexpr int BN = 32 * NT_W * WAVES_N;
    constexpr int NOUT = BN - PHALO;
    constexpr int WT = BN + PHALO;                 // row pitch of the intermediate tile in LDS
    constexpr int RPW = PCK / NW;
    constexpr int CIT = (BN + 64 + 63) / 64;       // (K - 1) * d1 <= 64
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int lhalf = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * NOUT;              // first final output of this tile
    const int pad1 = p.d1 * (p.K - 1) / 2, pad2 = (p.K - 1) / 2;
    const int tg0 = n0 - pad2;                     // sequence position of intermediate column 0
    const int xg0 = tg0 - pad1;                    // sequence position of staged x column 0
    const int W = p.W1;
    float *const buf0 = smem, *const buf1 = smem + PCK * W;
    const float *const xb = p.x + (long long)b * p.x_bs;

    // ------------------------------------------------------------------------------------------- phase 1: conv1(lrelu(x))
    f32x16 acc[NT_W];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = p.bias1[wm * 32 + acc_row(r, lhalf)];
#pragma unroll
        for (int j = 0; j < NT_W; ++j) acc[j][r] = bv;
    }
    float st[RPW][CIT];
    const __amdgpu_buffer_rsrc_t xsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)((long long)p.C * p.T * 4), 0x00020000);
    auto stage_load = [&](int chunk) __attribute__((always_inline)) {
        const int nbase = xg0 + lane;
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int voff = ((chunk * PCK + wave + NW * j) * p.T + nbase) * 4;
#pragma unroll
            for (int i = 0; i < CIT; ++i)
                st[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, voff + i * 256, 0, 0));
        }
    };
    const bool time_edge = (xg0 < 0) || (xg0 + W > p.T);
    auto stage_store = [&](float *buf) __attribute__((always_inline)) {
        auto run = [&](auto edge_tag) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
            for (int i = 0; i < CIT; ++i) {
                const int col = lane + 64 * i;
                const int n = xg0 + col;
                const bool okn = (n >= 0) && (n < p.T);
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    float v = st[j][i];
                    if constexpr (EDGE) v = okn ? v : 0.f;
                    v = fmaxf(v, 0.1f * v);
                    if (64 * (i + 1) <= BN || col < W) buf[(wave + NW * j) * W + col] = v;
                }
            }
        };
        if (time_edge) run(std::true_type{});
        else run(std::false_type{});
    };

    const int KT = p.K;
    const int nsteps = p.nchunks * KT;
    float a0[PCK / 2], a1[PCK / 2], a2[PCK / 2];
    const float *wbase = p.wp1 + (long long)wm * KT * p.nchunks * (PCK / 2) * 64 + lane * 4;
    auto load_a = [&](float (&dst)[PCK / 2], int chunk, int tap) __attribute__((always_inline)) {
        const float *src = wbase + ((long long)tap * p.nchunks + chunk) * (PCK / 2) * 64;
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
            const float4 t = *reinterpret_cast<const float4 *>(src + qd * 256);
            dst[qd * 4 + 0] = t.x; dst[qd * 4 + 1] = t.y; dst[qd * 4 + 2] = t.z; dst[qd * 4 + 3] = t.w;
        }
    };
    int pc = 0, pt = 0;
    auto advance = [&]() __attribute__((always_inline)) { if (++pt == KT) { pt = 0; ++pc; } };
    load_a(a0, pc, pt); advance();
    if (nsteps > 1) { load_a(a1, pc, pt); advance(); }

    stage_load(0);
    stage_store(buf0);
    if (p.nchunks > 1) stage_load(1);
    __syncthreads();

    int chunk = 0, tap = 0, s = 0;
    // one (chunk, tap) step: 8 channel pairs x NT_W MFMAs; `rows` = LDS base of the chunk, `pitch` its row pitch, `col0` the column
    // of this lane's first operand
    auto mma_step = [&](const float (&acur)[PCK / 2], const float *xs, int pitch) __attribute__((always_inline)) {
        float bf[NT_W], bn[NT_W];
#pragma unroll
        for (int j = 0; j < NT_W; ++j) bf[j] = xs[j * 32];
#pragma unroll
        for (int cp = 0; cp < PCK / 2; ++cp) {
            if (cp + 1 < PCK / 2) {
#pragma unroll
                for (int j = 0; j < NT_W; ++j) bn[j] = xs[(cp + 1) * 2 * pitch + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT_W; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[cp], bf[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT_W; ++j) bf[j] = bn[j];
        }
    };
    auto step1 = [&](float (&acur)[PCK / 2], float (&apre)[PCK / 2]) __attribute__((always_inline)) {
        const float *cur = (chunk & 1) ? buf1 : buf0;
        if (s + 2 < nsteps) { load_a(apre, pc, pt); advance(); }
        if (tap == 0) {
            if (chunk + 1 < p.nchunks) stage_store((chunk & 1) ? buf0 : buf1);
            if (chunk + 2 < p.nchunks) stage_load(chunk + 2);
        }
        mma_step(acur, cur + lhalf * W + wn * (NT_W * 32) + l31 + tap * p.d1, W);
        if (++tap == KT) {
            __syncthreads();
            tap = 0;
            ++chunk;
        }
        ++s;
    };
    while (s < nsteps) {
        step1(a0, a2);
        if (s < nsteps) step1(a1, a0);
        if (s < nsteps) step1(a2, a1);
    }
    // (the barrier after the last tap of the last chunk: every wave is done with the staging buffers)

    // ------------------------------------------------------------------ intermediate tile -> LDS, as conv2's B operand
    float *const Tb = smem;                                     // [C][WT]
#pragma unroll
    for (int j = 0; j < NT_W; ++j) {
        const int col = wn * (NT_W * 32) + j * 32 + l31;
        const int gpos = tg0 + col;
        const bool inside = (gpos >= 0) && (gpos < p.T);        // conv2 pads the SEQUENCE with zeros, not the tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[j][r];
            v = fmaxf(v, 0.1f * v);
            Tb[(wm * 32 + acc_row(r, lhalf)) * WT + col] = inside ? v : 0.f;
        }
    }
    // columns BN .. WT-1 feed only discarded outputs, but must be finite numbers: NaN * 0 weights would not matter, garbage may be NaN
    for (int e = tid; e < p.C * PHALO; e += 256) Tb[(e / PHALO) * WT + BN + (e % PHALO)] = 0.f;
    __syncthreads();

    // ------------------------------------------------------------------------------------------- phase 2: conv2 from LDS
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = p.bias2[wm * 32 + acc_row(r, lhalf)];
#pragma unroll
        for (int j = 0; j < NT_W; ++j) acc[j][r] = bv;
    }
    wbase = p.wp2 + (long long)wm * KT * p.nchunks * (PCK / 2) * 64 + lane * 4;
    pc = 0; pt = 0; chunk = 0; tap = 0; s = 0;
    load_a(a0, pc, pt); advance();
    if (nsteps > 1) { load_a(a1, pc, pt); advance(); }
    auto step2 = [&](float (&acur)[PCK / 2], float (&apre)[PCK / 2]) __attribute__((always_inline)) {
        if (s + 2 < nsteps) { load_a(apre, pc, pt); advance(); }
        mma_step(acur, Tb + (chunk * PCK + lhalf) * WT + wn * (NT_W * 32) + l31 + tap, WT);
        if (++tap == KT) { tap = 0; ++chunk; }
        ++s;
    };
    while (s < nsteps) {
        step2(a0, a2);
        if (s < nsteps) step2(a1, a0);
        if (s < nsteps) step2(a2, a1);
    }
    __syncthreads();                                            // the tile in LDS is consumed: its space becomes the epilogue's

    // ------------------------------------------------------------------------------------------- epilogue: + x [+ acc] [* scale]
    const int tile_row0 = wm * 32;
    const bool has_res = p.res != nullptr, has_acc = p.acc != nullptr;
    float *const yb = p.y + (long long)b * p.y_bs;
    const float *const resp = has_res ? p.res + (long long)b * p.res_bs : nullptr;
    const float *const accp = has_acc ? p.acc + (long long)b * p.acc_bs : nullptr;
    if (p.fast_epi && n0 + NOUT <= p.T) {
        constexpr int CW = 32 * NT_W, LPR = CW / 4, RPI = 64 / LPR, NIT = 8 / RPI;
        float *const Lw = smem + wave * 8 * CW;
        const int lrow = lane / LPR, c4 = (lane % LPR) * 4;
        const int ctile = wn * CW + c4;                          // column within the tile
        const bool live = ctile < NOUT;                          // NOUT % 4 == 0: a float4 is all in or all out
        const int colg = n0 + ctile;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            float4 r4[NIT], a4[NIT];
            long long goff[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                goff[it] = (long long)(tile_row0 + 8 * ps + it * RPI + lrow) * p.T + colg;
                if (live && has_res) r4[it] = *reinterpret_cast<const float4 *>(resp + goff[it]);
                if (live && has_acc) a4[it] = *reinterpret_cast<const float4 *>(accp + goff[it]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < NT_W; ++j) Lw[(q + 4 * lhalf) * CW + 32 * j + l31] = acc[j][4 * ps + q];
            if (live) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    float4 v = *reinterpret_cast<const float4 *>(Lw + (it * RPI + lrow) * CW + c4);
                    if (has_res) { v.x += r4[it].x; v.y += r4[it].y; v.z += r4[it].z; v.w += r4[it].w; }
                    if (has_acc) { v.x += a4[it].x; v.y += a4[it].y; v.z += a4[it].z; v.w += a4[it].w; }
                    if (p.scale != 1.f) { v.x *= p.scale; v.y *= p.scale; v.z *= p.scale; v.w *= p.scale; }
                    *reinterpret_cast<float4 *>(yb + goff[it]) = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT_W; ++j) {
            const int ctile = wn * (NT_W * 32) + j * 32 + l31;
            const int n = n0 + ctile;
            const bool okc = (ctile < NOUT) && (n < p.T);
            const int nc = min(n, p.T - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long off = (long long)(tile_row0 + acc_row(r, lhalf)) * p.T + nc;
                float v = acc[j][r];
                if (has_res) v += resp[off];
                if (has_acc) v += accp[off];
                v *= p.scale;
                if (okc) yb[off] = v;
            }
        }
    }
}

template <int WAVES_M, int WAVES_N>
static int launch_pair(const PairParams &p, hipStream_t s) {
    constexpr int BN = 128 * WAVES_N, NOUT = BN - PHALO, WT = BN + PHALO;
    auto kern = respair_kernel<WAVES_M, WAVES_N>;
    const size_t lds = sizeof(float) * std::max<size_t>((size_t)2 * PCK * p.W1, (size_t)p.C * WT);
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.T, NOUT), 1, (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    set_last_kernel("respair_kernel<%d, %d>", WAVES_M, WAVES_N);
    return VS_OK;
}

}  // namespace vs

using namespace vs;

extern "C" {

int vs_respair_supported(const vs_conv_t *c1, const vs_conv_t *c2) {
    if (!c1 || !c2) return 0;
    const int C = c1->c_in;
    return c1->kind == VS_CONV1D && c2->kind == VS_CONV1D && (C == 32 || C == 64) && c1->c_out == C && c2->c_in == C && c2->c_out == C &&
           c1->k == c2->k && (c1->k & 1) && c1->k <= PHALO + 1 && c2->dil == 1 && c1->pad == c1->dil * (c1->k - 1) / 2 &&
           c2->pad == (c2->k - 1) / 2 && (c1->k - 1) * c1->dil <= 64 && c1->flags == 0 && c2->flags == 0 &&
           c1->math != VS_MATH_SPLIT3 && c2->math != VS_MATH_SPLIT3;       // (the fused pair has no split-f16 instance yet)
}

int vs_respair_forward(vs_conv_t *c1, vs_conv_t *c2, const vs_conv_io_t *io, void *stream) {
    VS_REQUIRE(!io || (io->x_dtype == io->y_dtype && (io->x_dtype == VS_DTYPE_F32 || (io->x_dtype == VS_DTYPE_BF16 && c1 && c2 &&
               c1->math == VS_MATH_BF16 && c2->math == VS_MATH_BF16))),
               "vs_respair_forward: x and y share one element type; bf16-resident tensors need the plain-bf16 arithmetic on both convs");
    VS_REQUIRE(c1 && c2 && io, "vs_respair_forward: NULL argument");
    VS_REQUIRE(vs_respair_supported(c1, c2), "vs_respair_forward: unsupported pair of convs");
    VS_REQUIRE(c1->weights_set && c2->weights_set, "vs_respair_forward: weights not set");
    VS_REQUIRE(io->x && io->out[0].y && io->B > 0 && io->B <= 65535 && io->T > 0, "vs_respair_forward: bad io");
    VS_REQUIRE(io->in_act == VS_IN_LRELU && !io->mask && !io->bias_b && !io->split_row && io->out[0].mode == VS_OUT_LINEAR &&
                   io->out[0].out_act == VS_OUT_NONE && !io->out[0].out_mask,
               "vs_respair_forward: only the unmasked leaky-relu residual form is fused");
    const int C = c1->c_in;
    VS_REQUIRE((long long)C * io->T * 4 < (1ll << 31), "vs_respair_forward: item exceeds the 2 GiB buffer-descriptor range");
    PairParams p;
    memset(&p, 0, sizeof(p));
    const long long dflt = (long long)C * io->T;
    p.x = io->x; p.x_bs = io->x_bs ? io->x_bs : dflt;
    p.wp1 = c1->wp.as<float>(); p.bias1 = c1->biasp.as<float>();
    p.wp2 = c2->wp.as<float>(); p.bias2 = c2->biasp.as<float>();
    const vs_conv_out_t &o = io->out[0];
    p.y = o.y; p.res = o.res; p.acc = o.acc;
    p.y_bs = o.y_bs ? o.y_bs : dflt; p.res_bs = o.res_bs ? o.res_bs : dflt; p.acc_bs = o.acc_bs ? o.acc_bs : dflt;
    p.scale = (o.scale == 0.f) ? 1.f : o.scale;
    p.B = (int)io->B; p.C = C; p.T = (int)io->T; p.K = c1->k; p.d1 = c1->dil; p.nchunks = c1->nchunks;
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    p.fast_epi = (io->T % 4 == 0) && al16(p.y) && (p.y_bs % 4 == 0) && (!p.res || (al16(p.res) && p.res_bs % 4 == 0)) &&
                 (!p.acc || (al16(p.acc) && p.acc_bs % 4 == 0));
    hipStream_t s = as_stream(stream);
    VS_REQUIRE(c1->math == c2->math, "vs_respair_forward: the two convs of a pair must use the same arithmetic");
    if (c1->math != VS_MATH_F32) return vs_respair_split_launch(c1, c2, io, p.fast_epi, s);
    if (C == 32) { p.W1 = 512 + (p.K - 1) * p.d1; return launch_pair<1, 4>(p, s); }
    p.W1 = 256 + (p.K - 1) * p.d1;
    return launch_pair<2, 2>(p, s);
}

}  // extern "C"
