// attention_bf16.hip -- MultiHeadAttention.attention (reference modules/rel_transformer.py:148-179 + 181-243) on the bf16 matrix
// instruction v_mfma_f32_32x32x16_bf16, for the VS_MATH_BF16 arithmetic (BASELINE.json's long-form configuration: T_mel 4096,
// hidden 512 -> 2 heads of 256 channels, "bf16 activations/weights with fp32 accumulate").  Same algorithm as relattn_kernel
// (transformer_ops.hip: streaming softmax over key tiles, -1e4 mask fill, banded relative-key / relative-value terms by index
// arithmetic, no [T, T] tensor); what changes is the arithmetic of the two GEMMs and everything that follows from the 16-deep
// bf16 fragments:
//   * Q (pre-scaled by 1/sqrt(dk) in fp32), K, V and the probabilities P are rounded to bf16 (RNE); S = K^T Q and O = V P^T
//     accumulate in fp32; row max / sum, the rescale and the relative terms stay fp32;
//   * key tiles of 64: two S^T accumulator tiles per wave (keys on the rows, the wave's 32 queries on the lanes);
//   * K tile in LDS as [d/8][key][8 bf16]: an A fragment (8 consecutive d of one key) is one conflict-free ds_read_b128;
//   * V tile in LDS as [d][64 keys] with the keys of every 16-group stored in the order (0-3, 8-11 | 4-7, 12-15) and a row
//     pitch of 144 B: the bf16 pairs of S^T registers 8s .. 8s+7 ARE the B fragment of P^T for k-step s (k index
//     16s + 8(j>>2) + 4h + (j&3): cdna_hip_programming.md 3, "an accumulator tile as the next MFMA's operand"), and with that key
//     order the matching A fragment of V is one conflict-free ds_read_b128 per lane -- no LDS round trip for P, no shuffles;
//   * K / V tiles double-buffered in LDS; the next tile's K float4 loads are issued before the S^T MFMAs and rounded / written
//     after them, its V loads are issued then and written after the P V MFMAs; one barrier per tile.  Heads of up to 128 channels: two workgroups per CU; 129..256: one (a wave then
//     holds 128 output accumulators + 64 query registers + 128 staging registers).
#include "attn_common.h"
#include "conv_common.h"

namespace vs {

// AKT = keys per tile: 64 for heads of up to 128 channels (two S^T accumulator tiles per wave), 32 for wider heads (the fp32 staging
// registers of a tile scale with DT * AKT: at 256 channels a 64-key tile does not fit next to 128 output accumulators).
// TERMS = 1: bf16 operands (VS_MATH_BF16).  TERMS = 6: the split-bf16 x6 arithmetic of conv_split.hip -- q / sqrt(dk), k, v and the
// probabilities split EXACTLY into three bf16 planes, six cross products per product: fp32-class scores and outputs at 16/6 of the
// fp32 matrix rate (VS_MATH_SPLIT6); three times the LDS per tile, so 32-key tiles, ONE K / V buffer (two barriers per tile) and two
// workgroups per CU that overlap each other; heads of up to 128 channels.  TERMS = 3 (round 6; VS_MATH_SPLIT3, the default arithmetic
// of the path): the same four operands as TWO f16 planes under power-of-two scales -- one per query (a factor of a column of S^T), one per
// staged K tile (a factor of the score tile, applied in the fma that forms the exponential's argument), a running one for V (the output
// accumulators are rescaled by the exact ratio when a tile raises it), 2^14 for the probabilities -- and three cross products on
// v_mfma_f32_32x32x16_f16: the same fp32 class at half the matrix work; same tile shape and buffer scheme as TERMS = 6.
// PK: the K / V tiles arrive as ready LDS images (AttnParams::kvimg, attn_pack_kv_kernel): a tile is 9 (DT = 8) 16-byte loads and LDS
// writes per thread, no conversion -- in place, the fp32 -> bf16 conversion of a tile (64 values per thread at 256 channels: ~300 VALU
// instructions next to 32 MFMAs per wave, one wave per SIMD) and its 64 KB of fp32 loads were repeated by every one of the T / 128 query
// blocks.
// e^x for the softmax arguments (x <= 0; -inf gives an exact 0) in seven VALU instructions: x log2(e) as a rounded product plus its exact
// residual and the low word of log2(e), v_exp_f32 of the product, a first-order correction for the residual -- within ~1.5 ulp of expf,
// whose library form (range reduction, scaling, the overflow / underflow selects) costs ~16: at 17 exponentials per 32-key tile per lane
// the split kernels below are VALU-bound, not matrix-bound (round 6: 3 300 of a tile's cycles per wave were VALU issue).
template <bool GUARD>          // GUARD: the argument may be -inf
__device__ __forceinline__ float exp_nonpos(float x) {
    if constexpr (GUARD) x = fmaxf(x, -150.f);                               // (2^-216 -> 0: below the fp32 denormals; -inf would give inf - inf)
    const float t = x * 1.44269502162933349609375f;
    const float e = __builtin_fmaf(x, 1.9259629911266174681e-8f, __builtin_fmaf(x, 1.44269502162933349609375f, -t));
    const float y = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(y, e * 0.693147182464599609375f, y);
}

template <int DT, int AKT, int TERMS, bool PK = false>
__global__ void __launch_bounds__(256, (DT <= 4) ? 2 : 1) relattn_bf16_kernel(const AttnParams p) {
    static_assert(TERMS == 1 || TERMS == 3 || TERMS == 6, "plain bf16, split-f16 x3 or split-bf16 x6");
    static_assert(!PK || TERMS == 1, "pre-packed tiles: plain-bf16 arithmetic");
    constexpr bool F16 = (TERMS == 3);
    constexpr int NPL = (TERMS == 6) ? 3 : (F16 ? 2 : 1);
    constexpr int NBUF = (TERMS == 1) ? 2 : 1;
    constexpr int DKR = DT * 32;                     // padded head dim
    constexpr int NKS = DKR / 16;                    // k-steps of S^T = K^T Q
    constexpr int NKT = AKT / 32;                    // S^T accumulator tiles per key tile
    constexpr int KQ = AKT / 4;                      // key quads per tile
    constexpr int KW = (TERMS == 1) ? 4 : 2;         // keys per K staging cell (work per thread x planes: finer cells for the split)
    constexpr int KQW = AKT / KW;                    // K cells along the keys
    constexpr int AVP = AKT / 2 + 4;                 // V row pitch in dwords (AKT keys x 2 B + 16 B: conflict-free ds_read_b128 down a column)
    constexpr int KCELLS = (DKR / 8) * KQW;          // (d8, key group) cells of the K tile, 8 loads of KW floats each
    constexpr int KCPT = (KCELLS + 255) / 256;
    constexpr int VCELLS = DKR * KQ;                 // (d, key quad) cells of the V tile, one float4 each
    constexpr int VCPT = VCELLS / 256;
    constexpr int KPL = (DKR / 8) * AKT * 4;         // dwords per K plane
    constexpr int VPL = DKR * AVP;                   // dwords per V plane
    constexpr int KBUF = NPL * KPL;                  // dwords per K buffer
    constexpr int VBUF = NPL * VPL;                  // dwords per V buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int KSPL = (p.part && p.ksplit > 1) ? p.ksplit : 1;          // key split: blockIdx.x = query block * KSPL + key range
    const int ksi = blockIdx.x % KSPL;
    const int i0 = ((blockIdx.x / KSPL) * 4 + wave) * 32;
    const int dk = p.dk, T = p.T;
    const int nrel = (p.ws >= 0 && p.rel_k) ? 2 * p.ws + 1 : 0;

#ifdef VS_ATTN_STAMPS      // (tools/build_variant.py: shader-clock stamps of workgroup (0, 0, 0), wave 0 -- tools/attn_phase_stamps.py)
#define BSTAMP(k)                                                                                                    \
    do {                                                                                                             \
        if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wave == 0) {                        \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                              \
            if (lane == 0) p.stamps[(k)] = t_;                                                                       \
        }                                                                                                            \
    } while (0)
#else
#define BSTAMP(k) do { } while (0)
#endif
    BSTAMP(0);
    unsigned *Ks = reinterpret_cast<unsigned *>(smem);      // [NBUF][plane][DKR/8][AKT keys][4 dwords]
    unsigned *Vs = Ks + NBUF * KBUF;                         // [NBUF][plane][DKR][AVP]
    float *Ms = reinterpret_cast<float *>(Vs + NBUF * VBUF); // [2][AKT] key mask of the tile
    float *QRs = Ms + 2 * AKT;                               // [4][32][ATT_QRS] rel-key logits
    float *Sws = QRs + 4 * 32 * ATT_QRS;                     // [4][32][ATT_QRS] in-window raw scores
    unsigned *Xs = reinterpret_cast<unsigned *>(Sws + 4 * 32 * ATT_QRS);   // [2][4] split-f16: each wave's largest exponent of the staged K / V tile; [8 + buffer]: tile without masked keys
    float *RVs = smem;                                       // [nrel][dk] relative value embeddings: over the K buffers, after the loop

    const float *qb = p.q + (long long)b * p.bs + (long long)h * dk * T;
    const float *kb = p.k + (long long)b * p.bs + (long long)h * dk * T;
    const float *vb = p.v + (long long)b * p.bs + (long long)h * dk * T;
    const float *maskb = p.mask ? p.mask + (long long)b * T : nullptr;
    const float *relk = nrel ? p.rel_k + (long long)(p.nh_rel == 1 ? 0 : h) * nrel * dk : nullptr;
    const float *relv = nrel ? p.rel_v + (long long)(p.nh_rel == 1 ? 0 : h) * nrel * dk : nullptr;

    // ---- query fragments: B operand of S^T = K^T Q, element j of k-step ks = Q[d = 16 ks + 8 half + j][query l31] ----
    const int qi = i0 + l31;
    const int qic = min(qi, T - 1);
    // eight fp32 values -> NPL bf16-plane fragments (RNE for one plane, the exact three-way split otherwise)
    auto planes8 = [&](const float (&v)[8], u32x4 (&f)[NPL]) __attribute__((always_inline)) {
        unsigned d[4][NPL];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if constexpr (F16) split_pair_h(v[2 * t], v[2 * t + 1], d[t]);
            else split_pair<NPL>(v[2 * t], v[2 * t + 1], d[t]);
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            u32x4 o; o.x = d[0][pl]; o.y = d[1][pl]; o.z = d[2][pl]; o.w = d[3][pl];
            f[pl] = o;
        }
    };
    // D += sum over the cross products of the planes of a and b, smallest terms first (conv_split.hip)
    auto mma = [&](f32x16 &c, const u32x4 (&a)[NPL], const u32x4 (&bq)[NPL]) __attribute__((always_inline)) {
        auto mm = [&](int ta, int tb) __attribute__((always_inline)) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ta]), __builtin_bit_cast(bf16x8, bq[tb]), c, 0, 0, 0);
        };
        auto mh = [&](int ta, int tb) __attribute__((always_inline)) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ta]), __builtin_bit_cast(f16x8, bq[tb]), c, 0, 0, 0);
        };
        if constexpr (F16) { mh(NPL - 1, 0); mh(0, NPL - 1); mh(0, 0); return; }
        if constexpr (TERMS == 6) { mm(1, 1); mm(2, 0); mm(0, 2); mm(1, 0); mm(0, 1); }
        mm(0, 0);
    };
    // rel-key logits QR[i][r] = (q_i / sqrt(dk)) . rel_k[r] in fp32, from the very query values the fragments are made of: the table goes
    // into LDS once per workgroup, transposed and zero-padded ([d][16], over the K buffer, which nothing has touched yet), and every lane
    // walks the 8-channel groups of its half with one broadcast ds_read_b128 per four window positions.  (Round 6: the rolled loop this
    // replaces -- a global load of q and nine of rel_k per channel, one channel in flight -- was 88 000 of the 325 000 cycles of a launch at
    // T = 1024, tools/attn_phase_stamps.py.)
    float *RKs = smem;
    float qr[ATT_MAXREL];
#pragma unroll
    for (int r = 0; r < ATT_MAXREL; ++r) qr[r] = 0.f;
    if (nrel) {
#pragma unroll      // (consecutive lanes read consecutive channels of one window position: coalesced, all loads of a thread in flight)
        for (int i = 0; i < DKR * ATT_MAXREL / (256); ++i) {
            const int e = tid + i * (256);
            const int r = e / DKR, d = e % DKR;
            const float w = relk[min(r, nrel - 1) * dk + min(d, dk - 1)];      // (unconditional load on a clamped index: the loads of a thread overlap)
            RKs[d * ATT_MAXREL + r] = (d < dk && r < nrel) ? w : 0.f;
        }
    }
    __syncthreads();
    auto qr_add = [&](const float (&qv)[8], int ks) __attribute__((always_inline)) {
        if (!nrel) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 *row = reinterpret_cast<const float4 *>(RKs + (16 * ks + 8 * half + j) * ATT_MAXREL);
#pragma unroll
            for (int c4 = 0; c4 < ATT_MAXREL / 4; ++c4) {          // (all 16 columns of the zero-padded table: straight-line code)
                const float4 w = row[c4];
                qr[4 * c4 + 0] += qv[j] * w.x; qr[4 * c4 + 1] += qv[j] * w.y; qr[4 * c4 + 2] += qv[j] * w.z; qr[4 * c4 + 3] += qv[j] * w.w;
            }
        }
    };
    u32x4 qf[NKS][NPL];
    float q_inv = 1.f;                               // split-f16: 1 / (the power-of-two scale of this lane's query)
    if constexpr (F16) {
        // every query under its own scale (largest |q| / sqrt(dk) over the head's channels just below 2^15): a scale per COLUMN of S^T
        float qv[NKS][8];
        unsigned key = 0;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 16 * ks + 8 * half + j;
                const float v = qb[(long long)min(d, dk - 1) * T + qic];
                qv[ks][j] = (d < dk && qi < T) ? v * p.scale : 0.f;
                key = f16_maxkey(key, qv[ks][j]);
            }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qr_add(qv[ks], ks);
        key = max(key, (unsigned)__shfl_xor((int)key, 32));
        const int eq = max(f16_key_exponent(key), F16_EB_MIN);
        const float sq = f16_scale(eq);
        q_inv = f16_inv_scale(eq);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) qv[ks][j] *= sq;
            planes8(qv[ks], qf[ks]);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            float qv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 16 * ks + 8 * half + j;
                const float v = qb[(long long)min(d, dk - 1) * T + qic];
                qv[j] = (d < dk && qi < T) ? v * p.scale : 0.f;
            }
            qr_add(qv, ks);
            planes8(qv, qf[ks]);
        }
    }
    BSTAMP(1);
    float *QRw = QRs + wave * 32 * ATT_QRS;
    float *Sww = Sws + wave * 32 * ATT_QRS;
    for (int e = lane; e < 32 * ATT_QRS; e += 64) Sww[e] = -INFINITY;
    if (nrel) {
#pragma unroll
        for (int r = 0; r < ATT_MAXREL; ++r) {
            const float tot = qr[r] + __shfl_xor(qr[r], 32);      // the two lane halves hold complementary d's
            if (half == 0) QRw[l31 * ATT_QRS + r] = tot;
        }
    }

    BSTAMP(2);
    // ---- K / V tile staging ----
    constexpr int KU4 = KPL / 4, VU4 = VPL / 4;                          // 16-byte units of the K / V images (PK)
    constexpr int KIPT = (KU4 + 255) / 256, VIPT = (VU4 + 255) / 256;
    constexpr int IMG = KPL + VPL + AKT;                                  // dwords of a tile image: K, V, key mask
    float kst[PK ? 1 : KCPT][8][KW];
    float4 vst[PK ? 1 : VCPT];
    u32x4 kimg[PK ? KIPT : 1], vimg[PK ? VIPT : 1];
    float mst = 1.f;
    const unsigned *const imgb = PK ? p.kvimg + ((long long)(b * p.nh + h) * ((T + AKT - 1) / AKT)) * IMG : nullptr;
    auto load_k = [&](int jt) __attribute__((always_inline)) {
        if constexpr (PK) {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(imgb + (long long)jt * IMG);
#pragma unroll
            for (int i = 0; i < KIPT; ++i)
                if (KU4 % 256 == 0 || tid + 256 * i < KU4) kimg[i] = src[tid + 256 * i];
            if (tid < AKT) mst = u2f(imgb[(long long)jt * IMG + KPL + VPL + tid]);
            return;
        }
        const int j0 = jt * AKT;
#pragma unroll
        for (int i = 0; i < KCPT; ++i) {
            const int c = tid + 256 * i;
            const int kq = c % KQW, d8 = c / KQW;
            const int jc = min(j0 + KW * kq, T - KW);                // T % 4 == 0: a cell is wholly inside or wholly outside
#pragma unroll
            for (int jd = 0; jd < 8; ++jd) {
                const int d = min(8 * d8 + jd, dk - 1);
                if constexpr (KW == 4) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(kb + (long long)d * T + jc);
                    kst[i][jd][0] = t4.x; kst[i][jd][1] = t4.y; kst[i][jd][2] = t4.z; kst[i][jd][3] = t4.w;
                } else {
                    const float2 t2 = *reinterpret_cast<const float2 *>(kb + (long long)d * T + jc);
                    kst[i][jd][0] = t2.x; kst[i][jd][1] = t2.y;
                }
            }
        }
        if (tid < AKT) mst = maskb ? maskb[min(j0 + tid, T - 1)] : 1.f;
    };
    auto load_v = [&](int jt) __attribute__((always_inline)) {
        if constexpr (PK) {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(imgb + (long long)jt * IMG + KPL);
#pragma unroll
            for (int i = 0; i < VIPT; ++i)
                if (VU4 % 256 == 0 || tid + 256 * i < VU4) vimg[i] = src[tid + 256 * i];
            return;
        }
        const int j0 = jt * AKT;
#pragma unroll
        for (int i = 0; i < VCPT; ++i) {
            const int c = tid + 256 * i;
            const int kq = c % KQ, d = min(c / KQ, dk - 1);
            const int jc = min(j0 + 4 * kq, T - 4);
            vst[i] = *reinterpret_cast<const float4 *>(vb + (long long)d * T + jc);
        }
    };
    auto store_k = [&](int jt, int buf, float sk) __attribute__((always_inline)) {
        const int j0 = jt * AKT;
        unsigned *Kb = Ks + buf * KBUF;
        if constexpr (PK) {
#pragma unroll
            for (int i = 0; i < KIPT; ++i)
                if (KU4 % 256 == 0 || tid + 256 * i < KU4) reinterpret_cast<u32x4 *>(Kb)[tid + 256 * i] = kimg[i];
            if (tid < AKT) Ms[buf * AKT + tid] = mst;
            if (wave == 0) { const bool pl = __all(lane >= AKT || (mst != 0.f && j0 + lane < T)); if (lane == 0) Xs[8 + buf] = pl; }
            return;
        }
#pragma unroll
        for (int i = 0; i < KCPT; ++i) {
            const int c = tid + 256 * i;
            const int kq = c % KQW, d8 = c / KQW;
            if (KCELLS % 256 == 0 || c < KCELLS) {
                const bool okj = (j0 + KW * kq < T);
#pragma unroll
                for (int e = 0; e < KW; ++e) {
                    float v[8];
#pragma unroll
                    for (int jd = 0; jd < 8; ++jd) v[jd] = (okj && 8 * d8 + jd < dk) ? (F16 ? kst[i][jd][e] * sk : kst[i][jd][e]) : 0.f;
                    u32x4 f[NPL];
                    planes8(v, f);
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x4 *>(Kb + pl * KPL + (d8 * AKT + KW * kq + e) * 4) = f[pl];
                }
            }
        }
        if (tid < AKT) Ms[buf * AKT + tid] = (j0 + tid < T) ? mst : 1.f;
        // a tile wholly inside the sequence with no masked key: the score loop of its consumers skips the mask / range selects
        if (wave == 0) { const bool pl = __all(lane >= AKT || (mst != 0.f && j0 + lane < T)); if (lane == 0) Xs[8 + buf] = pl; }
    };
    auto store_v = [&](int jt, int buf, float sv) __attribute__((always_inline)) {
        const int j0 = jt * AKT;
        unsigned *Vb = Vs + buf * VBUF;
        if constexpr (PK) {
#pragma unroll
            for (int i = 0; i < VIPT; ++i)
                if (VU4 % 256 == 0 || tid + 256 * i < VU4) reinterpret_cast<u32x4 *>(Vb)[tid + 256 * i] = vimg[i];
            return;
        }
#pragma unroll
        for (int i = 0; i < VCPT; ++i) {
            const int c = tid + 256 * i;
            const int kq = c % KQ, d = c / KQ;
            const bool ok = (j0 + 4 * kq < T) && (d < dk);
            const float4 t4 = vst[i];
            unsigned lo[NPL], hi[NPL];
            if constexpr (F16) {
                split_pair_h(ok ? t4.x * sv : 0.f, ok ? t4.y * sv : 0.f, lo);
                split_pair_h(ok ? t4.z * sv : 0.f, ok ? t4.w * sv : 0.f, hi);
            } else {
                split_pair<NPL>(ok ? t4.x : 0.f, ok ? t4.y : 0.f, lo);
                split_pair<NPL>(ok ? t4.z : 0.f, ok ? t4.w : 0.f, hi);
            }
            // keys 4kq .. 4kq+3 of 16-group kq >> 2: quads (0, 1, 2, 3) of a group sit in slots (0, 2, 1, 3)
            const int slot = ((kq & 1) << 1) | ((kq >> 1) & 1);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                *reinterpret_cast<uint2 *>(Vb + pl * VPL + d * AVP + (kq >> 2) * 8 + slot * 2) = make_uint2(lo[pl], hi[pl]);
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_half = 0.f;
    const float mi = (maskb && qi < T) ? maskb[qi] : 1.f;
    const bool rows_plain = !__any(mi == 0.f);               // no masked query in this wave

    const int ntiles_all = (T + AKT - 1) / AKT;
    const int jt_lo = (int)((long long)ksi * ntiles_all / KSPL);
    const int ntiles = (int)((long long)(ksi + 1) * ntiles_all / KSPL);        // (end of this workgroup's key tiles)
    // split-f16: the staged K tile and V tile each go under ONE power-of-two scale -- K's from the tile's own largest magnitude (a scale per
    // tile is a common factor of a score tile), V's from the running maximum over the tiles so far (the output accumulators sum over tiles:
    // when a tile raises it they are rescaled by the exact ratio, together with the softmax rescale).  Every wave reduces the exponents of
    // what it staged (DPP), the four results cross through LDS under the barrier the single tile buffer needs anyway.
    int ek_cur = F16_EB_MIN, ev_cur = F16_EB_MIN, ev_acc = F16_EB_MIN, ek_next = F16_EB_MIN, ev_run = F16_EB_MIN;
    auto publish_exps = [&]() __attribute__((always_inline)) {
        unsigned kk = 0, kv = 0;
#pragma unroll
        for (int i = 0; i < (PK ? 1 : KCPT); ++i)
#pragma unroll
            for (int jd = 0; jd < 8; ++jd)
#pragma unroll
                for (int e = 0; e < KW; ++e) kk = f16_maxkey(kk, kst[i][jd][e]);
#pragma unroll
        for (int i = 0; i < (PK ? 1 : VCPT); ++i) {
            kv = f16_maxkey(kv, vst[i].x); kv = f16_maxkey(kv, vst[i].y); kv = f16_maxkey(kv, vst[i].z); kv = f16_maxkey(kv, vst[i].w);
        }
        const int wk = wave_max_u8(f16_key_exponent(kk)), wv = wave_max_u8(f16_key_exponent(kv));
        if (lane == 0) { Xs[wave] = (unsigned)wk; Xs[4 + wave] = (unsigned)wv; }
    };
    auto collect_exps = [&]() __attribute__((always_inline)) {
        const u32x4 a = *reinterpret_cast<const u32x4 *>(Xs), c = *reinterpret_cast<const u32x4 *>(Xs + 4);
        ek_next = __builtin_amdgcn_readfirstlane(max(max(max(a.x, a.y), max(a.z, a.w)), (unsigned)F16_EB_MIN));
        ev_run = __builtin_amdgcn_readfirstlane(max(max(max(c.x, c.y), max(c.z, c.w)), (unsigned)ev_run));
    };
    load_k(jt_lo);
    load_v(jt_lo);
    if constexpr (F16) publish_exps();
    __syncthreads();                                         // (every wave has read the rel-key table out of the K buffer)
    if constexpr (F16) {
        collect_exps();
        ev_acc = ev_run;
    }
    store_k(jt_lo, 0, f16_scale(ek_next));
    store_v(jt_lo, 0, f16_scale(ev_run));
    __syncthreads();
    BSTAMP(3);
    for (int jt = jt_lo; jt < ntiles; ++jt) {
        const int j0 = jt * AKT;
        const int buf = (NBUF == 2) ? ((jt - jt_lo) & 1) : 0;
#ifdef VS_ATTN_STAMPS
        if (jt - jt_lo < 100) BSTAMP(16 + jt - jt_lo);
#define TSTAMP(k) do { if (jt - jt_lo < 100) BSTAMP(128 + 8 * (jt - jt_lo) + (k)); } while (0)
#else
#define TSTAMP(k) do { } while (0)
#endif
        TSTAMP(0);
        const unsigned *Kb = Ks + buf * KBUF, *Vb = Vs + buf * VBUF;
        const float *Mb = Ms + buf * AKT;
        ek_cur = ek_next;
        ev_cur = ev_run;
        // two buffers: the next tile's K is in flight under the S^T MFMAs and written once they have issued, its V is in flight under
        // the softmax and the P V MFMAs (the fp32 staging registers of a tile -- 64 KB of K + 64 KB of V at 256 channels -- are never
        // all live).  One buffer (split arithmetic): both are in flight under the whole tile and written between two barriers.
        if (jt + 1 < ntiles) {
            load_k(jt + 1);
            if constexpr (NBUF == 1) load_v(jt + 1);
        }

        // ---- S^T tiles: rows = keys 32 kt + acc_row(r), columns (lanes) = queries ----
        f32x16 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                u32x4 a[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    a[pl] = *reinterpret_cast<const u32x4 *>(Kb + pl * KPL + ((2 * ks + half) * AKT + 32 * kt + l31) * 4);
                mma(s[kt], a, qf[ks]);
            }
        }
        if constexpr (NBUF == 2) {
            if (jt + 1 < ntiles) {
                store_k(jt + 1, buf ^ 1, 1.f);
                load_v(jt + 1);
            }
        }
        TSTAMP(1);
        const bool near_diag = nrel && (j0 + AKT - 1 >= i0 - p.ws) && (j0 <= i0 + 31 + p.ws);
        float tmax = -INFINITY;
        const float sfix = F16 ? f16_inv_scale(ek_cur) * q_inv : 1.f;      // (exact: powers of two)
        // a tile wholly inside the sequence with no masked key, rows with no masked query (every tile of a full-length item): no selects
        const bool plain = rows_plain && __builtin_amdgcn_readfirstlane((int)Xs[8 + buf]) != 0;
        float mulf = 1.f;                                    // what the exponential's fma still has to multiply the scores by
        if (plain && !near_diag) {
            // away from the diagonal nothing is added to a score: the maximum of the raw accumulators (one v_max3 per pair) times the positive
            // scale, and the scale itself rides in the fma that forms the exponential's argument
            float mx = s[0][0];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
            tmax = F16 ? mx * sfix : mx;
            mulf = sfix;
        } else if (plain) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float sv = F16 ? s[kt][r] * sfix : s[kt][r];
                    const int rel = j0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half - qi;
                    if (rel >= -p.ws && rel <= p.ws) {
                        sv += QRw[l31 * ATT_QRS + rel + p.ws];
                        Sww[l31 * ATT_QRS + rel + p.ws] = sv;
                    }
                    s[kt][r] = sv;
                    tmax = fmaxf(tmax, sv);
                }
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jj = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int j = j0 + jj;
                    float sv = F16 ? s[kt][r] * sfix : s[kt][r];
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
                    s[kt][r] = sv;
                    tmax = fmaxf(tmax, sv);
                }
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        float alpha;
        if constexpr (TERMS != 1) alpha = (m_run == -INFINITY) ? 0.f : exp_nonpos<true>(m_run - m_new);
        else alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
        float psum = 0.f;
        u32x4 pf[2 * NKT][NPL];                              // P^T fragments of the 16-key k-steps
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            float pv[16];
            if constexpr (TERMS != 1) {
                if (plain) {                                 // (finite scores: no -inf to guard)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pv[r] = exp_nonpos<false>(__builtin_fmaf(s[kt][r], mulf, -m_new));
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) pv[r] = exp_nonpos<true>(s[kt][r] - m_new);
                }
            } else {
                // exp(-inf) = 0 for excluded keys; with one bf16 plane the result is rounded to 8 bits anyway: the fast exp
#pragma unroll
                for (int r = 0; r < 16; ++r) pv[r] = __expf(s[kt][r] - m_new);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) psum += pv[r];
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float v8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = F16 ? pv[8 * sh + e] * 16384.f : pv[8 * sh + e];      // (split-f16: p <= 1 under the scale 2^14)
                planes8(v8, pf[2 * kt + sh]);
            }
        }
        l_half = l_half * alpha + psum;
        m_run = m_new;
        // the running maximum of a row settles after a few tiles: skip the rescale of the output accumulators (a read-multiply-write
        // of DT * 16 registers per tile) whenever no query of the wave moved its maximum
        float oresc = alpha;
        if constexpr (F16) {
            if (ev_cur != ev_acc) {                          // this V tile sits under a smaller scale than the accumulators: bring them down to it
                const int dl = ev_acc - ev_cur;              // < 0
                oresc *= (dl < -126) ? 0.f : u2f((unsigned)(127 + dl) << 23);
                ev_acc = ev_cur;
            }
        }
        if (__any(oresc != 1.f)) {
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= oresc;
        }

        TSTAMP(2);
        // ---- O^T += V P^T: k-step s4 sums over the keys 16 s4 + 8 (j >> 2) + 4 half + (j & 3), the order of the V rows ----
#pragma unroll
        for (int s4 = 0; s4 < 2 * NKT; ++s4) {
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                u32x4 a[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    a[pl] = *reinterpret_cast<const u32x4 *>(Vb + pl * VPL + (t * 32 + l31) * AVP + s4 * 8 + half * 4);
                mma(o[t], a, pf[s4]);
            }
        }
        if constexpr (NBUF == 2) {
            if (jt + 1 < ntiles) store_v(jt + 1, buf ^ 1, 1.f);
            __syncthreads();
        } else {
            TSTAMP(3);
            if constexpr (F16) {
                if (jt + 1 < ntiles) publish_exps();
            }
            TSTAMP(4);
            __syncthreads();                                 // every wave is done with the (single) K / V buffer
            TSTAMP(5);
            if (jt + 1 < ntiles) {
                if constexpr (F16) collect_exps();
                store_k(jt + 1, 0, f16_scale(ek_next));
                store_v(jt + 1, 0, f16_scale(ev_run));
            }
            TSTAMP(6);
            __syncthreads();
            TSTAMP(7);
        }
    }

    BSTAMP(4);
    const float ounscale = F16 ? f16_inv_scale(ev_acc) : 1.f;      // split-f16: the accumulators hold V * 2^(141 - ev_acc) times P * 2^14
    if (KSPL > 1) {
        // key split: un-normalised rows (relative to this range's maximum), the maximum, the sum and the in-window raw scores of this key
        // range; launch_attn_combine() merges the ranges, adds the relative-value term and normalises
        const float l_tot = l_half + __shfl_xor(l_half, 32);
        const int rows = dk + 2 + nrel;
        float *pb = p.part + ((long long)(b * p.nh + h) * KSPL + ksi) * rows * T;
        if (qi < T) {
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (d < dk) pb[(long long)d * T + qi] = F16 ? o[t][r] * (1.f / 16384.f) * ounscale : o[t][r];
                }
            if (half == 0) {
                pb[(long long)dk * T + qi] = m_run;
                pb[(long long)(dk + 1) * T + qi] = l_tot;
            }
            for (int rr = half; rr < nrel; rr += 2) pb[(long long)(dk + 2 + rr) * T + qi] = Sww[l31 * ATT_QRS + rr];
        }
        return;
    }

    // ---- finish: normalise, add the relative-value term (fp32), store ----
    // (the loop's last barrier has retired every read of the K / V buffers: 2 * KBUF dwords >= 16 rows x 256 channels)
    for (int e = tid; e < nrel * DKR; e += 256) RVs[e] = (e % DKR < dk) ? relv[(e / DKR) * dk + e % DKR] : 0.f;      // rows of DKR: float4 reads below
    __syncthreads();
    const float l_tot = l_half + __shfl_xor(l_half, 32);
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = F16 ? o[t][r] * (1.f / 16384.f) * ounscale * inv : o[t][r] * inv;
    // sum_r p[i, i + r - ws] * rel_v[r]: the window index is the (rolled) outer loop so that every access to the output accumulators
    // has a compile-time index (a runtime-indexed accumulator array lives in scratch memory)
#pragma unroll 1
    for (int rr = 0; rr < nrel; ++rr) {
        const float w = expf(Sww[l31 * ATT_QRS + rr] - m_run) * inv;
        const float *rv = RVs + rr * DKR;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 v4 = *reinterpret_cast<const float4 *>(rv + t * 32 + 8 * r4 + 4 * half);
                o[t][4 * r4 + 0] += w * v4.x; o[t][4 * r4 + 1] += w * v4.y; o[t][4 * r4 + 2] += w * v4.z; o[t][4 * r4 + 3] += w * v4.w;
            }
    }
    BSTAMP(5);
    float *ob = p.out + (long long)b * p.out_bs + (long long)h * dk * T;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (d < dk && qi < T) ob[(long long)d * T + qi] = o[t][r];
        }
    }
    BSTAMP(6);
}
#undef BSTAMP
#undef TSTAMP

// The LDS images of the K / V tiles of one (batch, head), written once per launch (AttnParams::kvimg): exactly the bytes store_k / store_v
// of relattn_bf16_kernel<DT, AKT, 1> put into LDS -- same cells, same key order, same zero fill beyond T and dk, RNE to bf16 -- followed
// by the tile's key mask.  blockIdx = (key tile, head, batch).
template <int DT, int AKT>
__global__ void __launch_bounds__(256) attn_pack_kv_kernel(const AttnParams p) {
    constexpr int DKR = DT * 32, KQ = AKT / 4, AVP = AKT / 2 + 4;
    constexpr int KCELLS = (DKR / 8) * KQ, VCELLS = DKR * KQ;
    constexpr int KPL = (DKR / 8) * AKT * 4, VPL = DKR * AVP, IMG = KPL + VPL + AKT;
    const int tid = threadIdx.x, jt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int dk = p.dk, T = p.T, j0 = jt * AKT;
    const float *kb = p.k + (long long)b * p.bs + (long long)h * dk * T;
    const float *vb = p.v + (long long)b * p.bs + (long long)h * dk * T;
    unsigned *img = p.kvimg + ((long long)(b * p.nh + h) * gridDim.x + jt) * IMG;
    for (int c = tid; c < KCELLS; c += 256) {
        const int kq = c % KQ, d8 = c / KQ;
        const int jc = min(j0 + 4 * kq, T - 4);
        const bool okj = (j0 + 4 * kq < T);
        float kv[8][4];
#pragma unroll
        for (int jd = 0; jd < 8; ++jd) {
            const float4 t4 = *reinterpret_cast<const float4 *>(kb + (long long)min(8 * d8 + jd, dk - 1) * T + jc);
            const bool ok = okj && (8 * d8 + jd < dk);
            kv[jd][0] = ok ? t4.x : 0.f; kv[jd][1] = ok ? t4.y : 0.f; kv[jd][2] = ok ? t4.z : 0.f; kv[jd][3] = ok ? t4.w : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u32x4 f;
            f.x = pack_hi(rne_bf16(kv[0][e]), rne_bf16(kv[1][e])); f.y = pack_hi(rne_bf16(kv[2][e]), rne_bf16(kv[3][e]));
            f.z = pack_hi(rne_bf16(kv[4][e]), rne_bf16(kv[5][e])); f.w = pack_hi(rne_bf16(kv[6][e]), rne_bf16(kv[7][e]));
            *reinterpret_cast<u32x4 *>(img + (d8 * AKT + 4 * kq + e) * 4) = f;
        }
    }
    // (the four pad dwords at the end of every V row are never read)
    for (int c = tid; c < VCELLS; c += 256) {
        const int kq = c % KQ, d = c / KQ;
        const int jc = min(j0 + 4 * kq, T - 4);
        const bool ok = (j0 + 4 * kq < T) && (d < dk);
        const float4 t4 = *reinterpret_cast<const float4 *>(vb + (long long)min(d, dk - 1) * T + jc);
        const int slot = ((kq & 1) << 1) | ((kq >> 1) & 1);
        *reinterpret_cast<uint2 *>(img + KPL + d * AVP + (kq >> 2) * 8 + slot * 2) =
            make_uint2(pack_hi(rne_bf16(ok ? t4.x : 0.f), rne_bf16(ok ? t4.y : 0.f)), pack_hi(rne_bf16(ok ? t4.z : 0.f), rne_bf16(ok ? t4.w : 0.f)));
    }
    if (tid < AKT) {
        const float m = (p.mask && j0 + tid < T) ? p.mask[(long long)b * T + j0 + tid] : 1.f;
        img[KPL + VPL + tid] = f2u(m);
    }
}

static int akt_of(int DT) { return DT <= 4 ? 64 : 32; }
static size_t kv_image_dwords(int DT) {
    const int AKT = akt_of(DT), DKR = DT * 32;
    return (size_t)(DKR / 8) * AKT * 4 + (size_t)DKR * (AKT / 2 + 4) + AKT;
}
static int dt_instance(int dk) {
    const int DT = (int)ceil_div(dk, 32);
    return DT <= 2 ? 2 : (DT <= 4 ? DT : (DT <= 6 ? 6 : 8));
}

// pre-packing pays once a tile is used by enough query blocks: T >= 1024 (eight blocks of 128 queries)
size_t attn_kv_work_bytes(long long B, int nh, int dk, long long T) {
    if (dk > 256 || T < 1024 || (T % 4) != 0 || opt(OPT_NO_ATTN_KVPACK)) return 0;
    const int DT = dt_instance(dk);
    return (size_t)B * nh * ceil_div(T, akt_of(DT)) * kv_image_dwords(DT) * 4;
}

// merge of the key ranges: out[d] = (sum_k O_k[d] e^{m_k - m}) / L + sum_r e^{s_r - m} / L * rel_v[r][d],
// m = max_k m_k, L = sum_k l_k e^{m_k - m}, s_r = the in-window score (owned by exactly one range; -inf elsewhere).
// Block = 64 queries x 4 channel groups, blockIdx.x = query block * CG + channel-group block: each thread re-derives the weights of its
// query (a few dozen loads) and merges dk / (4 CG) channels -- one thread per query walked all channels serially: 180 us at B = 1.
constexpr int CMB_CG = 4;       // channel-group blocks per query block (x 4 groups per block = 16 groups)
__global__ void __launch_bounds__(256) relattn_combine_kernel(const AttnParams p) {
    const int qi = (blockIdx.x / CMB_CG) * 64 + (threadIdx.x & 63);
    const int grp = (blockIdx.x % CMB_CG) * 4 + (threadIdx.x >> 6);          // 0 .. 4 CMB_CG - 1
    const int h = blockIdx.y, b = blockIdx.z;
    if (qi >= p.T) return;
    const int dk = p.dk, T = p.T, KS = p.ksplit;
    const int nrel = (p.ws >= 0 && p.rel_k) ? 2 * p.ws + 1 : 0;
    const int rows = dk + 2 + nrel;
    const float *pb = p.part + (long long)(b * p.nh + h) * KS * rows * T + qi;
    float mk[16], m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mk[k] = (k < KS) ? pb[((long long)k * rows + dk) * T] : -INFINITY;
        m = fmaxf(m, mk[k]);
    }
    float wk[16], L = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        wk[k] = (k < KS && mk[k] != -INFINITY) ? expf(mk[k] - m) : 0.f;
        if (k < KS) L += pb[((long long)k * rows + dk + 1) * T] * wk[k];
    }
    const float inv = 1.0f / L;
    float wr[ATT_MAXREL];
#pragma unroll
    for (int rr = 0; rr < ATT_MAXREL; ++rr) {
        float sb = -INFINITY;
        if (rr < nrel) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < KS) sb = fmaxf(sb, pb[((long long)k * rows + dk + 2 + rr) * T]);
        }
        wr[rr] = (rr < nrel) ? expf(sb - m) * inv : 0.f;
    }
    const float *relv = nrel ? p.rel_v + (long long)(p.nh_rel == 1 ? 0 : h) * nrel * dk : nullptr;
    float *ob = p.out + (long long)b * p.out_bs + (long long)h * dk * T + qi;
    const int per = (dk + 4 * CMB_CG - 1) / (4 * CMB_CG);
    for (int d = grp * per; d < min(dk, (grp + 1) * per); ++d) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < KS) acc += pb[((long long)k * rows + d) * T] * wk[k];
        acc *= inv;
#pragma unroll
        for (int rr = 0; rr < ATT_MAXREL; ++rr)
            if (rr < nrel) acc += wr[rr] * relv[rr * dk + d];
        ob[(long long)d * T] = acc;
    }
}

int launch_attn_combine(const AttnParams &p, hipStream_t s) {
    dim3 grid((unsigned)(ceil_div(p.T, 64) * CMB_CG), (unsigned)p.nh, (unsigned)p.B);
    hipLaunchKernelGGL(relattn_combine_kernel, grid, dim3(256), 0, s, p);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

bool attn_bf16_supported(const AttnParams &p, int terms) {
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    return p.dk <= (terms == 1 ? 256 : 128) && p.T >= 4 && (p.T % 4) == 0 && al16(p.k) && al16(p.v) && (p.bs % 4) == 0 && ((long long)p.dk * p.T) % 4 == 0;
}

template <int DT, int AKT, int TERMS, bool PK = false>
static int launch_bf16(const AttnParams &p, hipStream_t s) {
    constexpr int DKR = DT * 32, AVP = AKT / 2 + 4, NPL = (TERMS == 6) ? 3 : (TERMS == 3 ? 2 : 1), NBUF = (TERMS == 1) ? 2 : 1;
    const size_t lds = 4 * ((size_t)NBUF * NPL * (DKR / 8) * AKT * 4 + (size_t)NBUF * NPL * DKR * AVP + 2 * AKT + 2 * 4 * 32 * ATT_QRS + 12);
    auto kern = relattn_bf16_kernel<DT, AKT, TERMS, PK>;
    if constexpr (PK) {
        dim3 pgrid((unsigned)ceil_div(p.T, AKT), (unsigned)p.nh, (unsigned)p.B);
        hipLaunchKernelGGL((attn_pack_kv_kernel<DT, AKT>), pgrid, dim3(256), 0, s, p);
        VS_CHECK_HIP(hipGetLastError());
        if (AKT == 32 && attn_dma_supported(p)) return launch_attn_dma(p, s);      // (round 4: the LDS-DMA ring kernel, attention_dma.hip)
    }
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (lds > 160 * 1024) { set_error("vs_relattn_fwd (bf16): head dim %d needs %zu B of LDS", p.dk, lds); return VS_EUNSUPPORTED; }
    const int ks = (p.part && p.ksplit > 1) ? p.ksplit : 1;
    dim3 grid((unsigned)(ceil_div(p.T, 128) * ks), (unsigned)p.nh, (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    if (PK) set_last_kernel("relattn_bf16_kernel<%d, %d, %d, true>", DT, AKT, TERMS);
    else set_last_kernel("relattn_bf16_kernel<%d, %d, %d>", DT, AKT, TERMS);
    if (ks > 1) return launch_attn_combine(p, s);
    return VS_OK;
}

int launch_attn_bf16(const AttnParams &p, int terms, hipStream_t s) {
    const int DT = (int)ceil_div(p.dk, 32);
    if (terms == 3) {
        if (DT <= 2) return launch_bf16<2, 32, 3>(p, s);
        if (DT == 3) return launch_bf16<3, 32, 3>(p, s);
        return launch_bf16<4, 32, 3>(p, s);
    }
    if (terms == 6) {
        if (DT <= 2) return launch_bf16<2, 32, 6>(p, s);
        if (DT == 3) return launch_bf16<3, 32, 6>(p, s);
        return launch_bf16<4, 32, 6>(p, s);
    }
    if (p.kvimg) {
        if (DT <= 2) return launch_bf16<2, 64, 1, true>(p, s);
        if (DT == 3) return launch_bf16<3, 64, 1, true>(p, s);
        if (DT == 4) return launch_bf16<4, 64, 1, true>(p, s);
        if (DT <= 6) return launch_bf16<6, 32, 1, true>(p, s);
        return launch_bf16<8, 32, 1, true>(p, s);
    }
    if (DT <= 2) return launch_bf16<2, 64, 1>(p, s);
    if (DT == 3) return launch_bf16<3, 64, 1>(p, s);
    if (DT == 4) return launch_bf16<4, 64, 1>(p, s);
    if (DT <= 6) return launch_bf16<6, 32, 1>(p, s);
    return launch_bf16<8, 32, 1>(p, s);
}

}  // namespace vs
