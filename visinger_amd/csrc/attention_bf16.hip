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
// fp32 matrix rate (VS_MATH_SPLIT6, the default arithmetic of the path); three times the LDS per tile, so 32-key tiles, ONE K / V
// buffer (two barriers per tile) and two workgroups per CU that overlap each other; heads of up to 128 channels.
// PK: the K / V tiles arrive as ready LDS images (AttnParams::kvimg, attn_pack_kv_kernel): a tile is 9 (DT = 8) 16-byte loads and LDS
// writes per thread, no conversion -- in place, the fp32 -> bf16 conversion of a tile (64 values per thread at 256 channels: ~300 VALU
// instructions next to 32 MFMAs per wave, one wave per SIMD) and its 64 KB of fp32 loads were repeated by every one of the T / 128 query
// blocks.
template <int DT, int AKT, int TERMS, bool PK = false>
__global__ void __launch_bounds__(256, (DT <= 4) ? 2 : 1) relattn_bf16_kernel(const AttnParams p) {
    static_assert(TERMS == 1 || TERMS == 6, "plain bf16 or split-bf16 x6");
    static_assert(!PK || TERMS == 1, "pre-packed tiles: plain-bf16 arithmetic");
    constexpr int NPL = (TERMS == 6) ? 3 : 1;
    constexpr int NBUF = (TERMS == 6) ? 1 : 2;
    constexpr int DKR = DT * 32;                     // padded head dim
    constexpr int NKS = DKR / 16;                    // k-steps of S^T = K^T Q
    constexpr int NKT = AKT / 32;                    // S^T accumulator tiles per key tile
    constexpr int KQ = AKT / 4;                      // key quads per tile
    constexpr int KW = (TERMS == 6) ? 2 : 4;         // keys per K staging cell (work per thread x planes: finer cells for the split)
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

    unsigned *Ks = reinterpret_cast<unsigned *>(smem);      // [NBUF][plane][DKR/8][AKT keys][4 dwords]
    unsigned *Vs = Ks + NBUF * KBUF;                         // [NBUF][plane][DKR][AVP]
    float *Ms = reinterpret_cast<float *>(Vs + NBUF * VBUF); // [2][AKT] key mask of the tile
    float *QRs = Ms + 2 * AKT;                               // [4][32][ATT_QRS] rel-key logits
    float *Sws = QRs + 4 * 32 * ATT_QRS;                     // [4][32][ATT_QRS] in-window raw scores
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
        for (int t = 0; t < 4; ++t) split_pair<NPL>(v[2 * t], v[2 * t + 1], d[t]);
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
        if constexpr (TERMS == 6) { mm(1, 1); mm(2, 0); mm(0, 2); mm(1, 0); mm(0, 1); }
        mm(0, 0);
    };
    u32x4 qf[NKS][NPL];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        float qv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 16 * ks + 8 * half + j;
            const float v = qb[(long long)min(d, dk - 1) * T + qic];
            qv[j] = (d < dk && qi < T) ? v * p.scale : 0.f;
        }
        planes8(qv, qf[ks]);
    }
    // rel-key logits QR[i][r] = (q_i / sqrt(dk)) . rel_k[r] from the fp32 query (a rolled loop: prologue code, kept off the
    // register budget of the main loop); each lane half covers every other group of 8 channels
    float qr[ATT_MAXREL];
#pragma unroll
    for (int r = 0; r < ATT_MAXREL; ++r) qr[r] = 0.f;
    if (nrel) {
#pragma unroll 1
        for (int d8 = half; d8 < (dk + 7) / 8; d8 += 2) {
#pragma unroll 1
            for (int j = 0; j < 8; ++j) {
                const int d = 8 * d8 + j;
                if (d < dk && qi < T) {
                    const float qs = qb[(long long)d * T + qic] * p.scale;
#pragma unroll
                    for (int r = 0; r < ATT_MAXREL; ++r)
                        if (r < nrel) qr[r] += qs * relk[r * dk + d];
                }
            }
        }
    }
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
    auto store_k = [&](int jt, int buf) __attribute__((always_inline)) {
        const int j0 = jt * AKT;
        unsigned *Kb = Ks + buf * KBUF;
        if constexpr (PK) {
#pragma unroll
            for (int i = 0; i < KIPT; ++i)
                if (KU4 % 256 == 0 || tid + 256 * i < KU4) reinterpret_cast<u32x4 *>(Kb)[tid + 256 * i] = kimg[i];
            if (tid < AKT) Ms[buf * AKT + tid] = mst;
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
                    for (int jd = 0; jd < 8; ++jd) v[jd] = (okj && 8 * d8 + jd < dk) ? kst[i][jd][e] : 0.f;
                    u32x4 f[NPL];
                    planes8(v, f);
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x4 *>(Kb + pl * KPL + (d8 * AKT + KW * kq + e) * 4) = f[pl];
                }
            }
        }
        if (tid < AKT) Ms[buf * AKT + tid] = (j0 + tid < T) ? mst : 1.f;
    };
    auto store_v = [&](int jt, int buf) __attribute__((always_inline)) {
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
            split_pair<NPL>(ok ? t4.x : 0.f, ok ? t4.y : 0.f, lo);
            split_pair<NPL>(ok ? t4.z : 0.f, ok ? t4.w : 0.f, hi);
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

    const int ntiles_all = (T + AKT - 1) / AKT;
    const int jt_lo = (int)((long long)ksi * ntiles_all / KSPL);
    const int ntiles = (int)((long long)(ksi + 1) * ntiles_all / KSPL);        // (end of this workgroup's key tiles)
    load_k(jt_lo);
    load_v(jt_lo);
    store_k(jt_lo, 0);
    store_v(jt_lo, 0);
    __syncthreads();
    for (int jt = jt_lo; jt < ntiles; ++jt) {
        const int j0 = jt * AKT;
        const int buf = (NBUF == 2) ? ((jt - jt_lo) & 1) : 0;
        const unsigned *Kb = Ks + buf * KBUF, *Vb = Vs + buf * VBUF;
        const float *Mb = Ms + buf * AKT;
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
                store_k(jt + 1, buf ^ 1);
                load_v(jt + 1);
            }
        }
        const bool near_diag = nrel && (j0 + AKT - 1 >= i0 - p.ws) && (j0 <= i0 + 31 + p.ws);
        float tmax = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int j = j0 + jj;
                float sv = s[kt][r];
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
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        float alpha;
        if constexpr (TERMS == 6) alpha = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
        else alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
        float psum = 0.f;
        u32x4 pf[2 * NKT][NPL];                              // P^T fragments of the 16-key k-steps
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // exp(-inf) = 0 for excluded keys; with one bf16 plane the result is rounded to 8 bits anyway: the fast exp
                if constexpr (TERMS == 6) pv[r] = expf(s[kt][r] - m_new);
                else pv[r] = __expf(s[kt][r] - m_new);
                psum += pv[r];
            }
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float v8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = pv[8 * sh + e];
                planes8(v8, pf[2 * kt + sh]);
            }
        }
        l_half = l_half * alpha + psum;
        m_run = m_new;
        // the running maximum of a row settles after a few tiles: skip the rescale of the output accumulators (a read-multiply-write
        // of DT * 16 registers per tile) whenever no query of the wave moved its maximum
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }

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
            if (jt + 1 < ntiles) store_v(jt + 1, buf ^ 1);
            __syncthreads();
        } else {
            __syncthreads();                                 // every wave is done with the (single) K / V buffer
            if (jt + 1 < ntiles) {
                store_k(jt + 1, 0);
                store_v(jt + 1, 0);
            }
            __syncthreads();
        }
    }

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
                    if (d < dk) pb[(long long)d * T + qi] = o[t][r];
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
    for (int e = tid; e < nrel * dk; e += 256) RVs[e] = relv[e];
    __syncthreads();
    const float l_tot = l_half + __shfl_xor(l_half, 32);
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= inv;
    // sum_r p[i, i + r - ws] * rel_v[r]: the window index is the (rolled) outer loop so that every access to the output accumulators
    // has a compile-time index (a runtime-indexed accumulator array lives in scratch memory)
#pragma unroll 1
    for (int rr = 0; rr < nrel; ++rr) {
        const float w = expf(Sww[l31 * ATT_QRS + rr] - m_run) * inv;
        const float *rv = RVs + rr * dk;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] += w * rv[min(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, dk - 1)];
    }
    float *ob = p.out + (long long)b * p.out_bs + (long long)h * dk * T;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (d < dk && qi < T) ob[(long long)d * T + qi] = o[t][r];
        }
    }
}

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
    return p.dk <= (terms == 6 ? 128 : 256) && p.T >= 4 && (p.T % 4) == 0 && al16(p.k) && al16(p.v) && (p.bs % 4) == 0 && ((long long)p.dk * p.T) % 4 == 0;
}

template <int DT, int AKT, int TERMS, bool PK = false>
static int launch_bf16(const AttnParams &p, hipStream_t s) {
    constexpr int DKR = DT * 32, AVP = AKT / 2 + 4, NPL = (TERMS == 6) ? 3 : 1, NBUF = (TERMS == 6) ? 1 : 2;
    const size_t lds = 4 * ((size_t)NBUF * NPL * (DKR / 8) * AKT * 4 + (size_t)NBUF * NPL * DKR * AVP + 2 * AKT + 2 * 4 * 32 * ATT_QRS);
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
