// attention_train.hip -- the attention core of the TRAINING path (modules/rel_transformer.py:148-179 with the relative
// terms of :181-243 and the dropout of :173) as three streaming kernels on the exact-fp32 matrix instruction: forward (output + row
// log-sum-exp), backward for dQ (+ the gradients of both relative-embedding tables), backward for dK / dV.  No [T, T] tensor exists in
// either direction: scores, probabilities and their gradients are recomputed tile by tile from q, k, v, the saved log-sum-exp and
// D = rowsum(dO * O); the dropout mask is a counter-based hash of (seed, batch * head, query, key) evaluated wherever it is needed.
//
// The problem is small (B = 16, T = 512, 2 heads of 96 channels: 11 GFLOP per layer forward + backward) and was bound by the ~40 PyTorch
// launches and the [B, h, T, T] round trips of the autograd version, so the kernels are written for simplicity, not for the roofline:
//   * one WAVE per workgroup owns a tile of 32 queries (forward, dQ) or 32 keys (dK / dV) of one (batch, head) and walks over the tiles of
//     the other side; all operand tiles are fp32 [channel][32 positions] in LDS (pitch 33);
//   * orientation: the owner's positions sit on the LANES, the other side's on the accumulator REGISTERS.  The first GEMM of a tile pair
//     contracts over channels (A = other side's tile, B = owner's tile: conflict-free row reads); its result -- probabilities or score
//     gradients, lane = owner position, register r = other-side position acc_pos(r, half) -- is, register for register, the B operand of
//     the second GEMM, which contracts over the other side's positions in exactly that order (A = M[c][acc_pos(s, half)]): no transposes,
//     no LDS round trip, no shuffles between the GEMMs.  Row statistics of a query are in-register reductions + one cross-half exchange
//     in the query-owner kernels; the key-owner kernel reads lse / D / the relative rows of its 32 queries from LDS;
//   * relative terms by index arithmetic on the band |key - query| <= window: QR[q][r] = scale * q . rel_k[r] and DOR[q][r] = dO . rel_v[r]
//     (2w+1 dots per query, computed by the query-owner kernel and handed to the key-owner kernel through a [B, h, T, 2w+1] table);
//     d rel_k / d rel_v are per-workgroup partial sums [tile][2w+1][dk] reduced by the caller (deterministic).
#include "attn_common.h"

#include <cstdlib>

namespace vs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TP = 33;             // LDS pitch of a [channel][32 positions] tile
constexpr int RP = ATT_MAXREL + 1; // LDS pitch of the per-query relative rows
constexpr float NEG_BIG = -3.0e38f;

struct AttnTrainParams {
    const float *q, *k, *v;
    long long bs;                   // batch stride of q / k / v
    const float *rel_k, *rel_v;     // [nh_rel, R, dk] or null (no window)
    const float *mask;              // [B, T] or null
    float *out;                     // forward: written; backward: read
    const float *dout;
    long long out_bs;               // batch stride of out / dout
    float *lse;                     // [2][B, nh, T]: row maximum m, log of the row sum of exp(score - m)  (kept apart: a fully masked row
                                    // has m = -1e4, where fp32 m + log(l) would round the log away)
    float *gq, *gk, *gv;
    long long g_bs;                 // batch stride of dq / dk / dv
    float *dvec, *qr, *dor;         // [B, nh, T], [B, nh, T, R], [B, nh, T, R]
    float *drelk_part, *drelv_part; // [B * nh * ceil(T / 32)][R][dk]
    int B, nh, dk, T, ws, nh_rel, R;
    float scale, inv_keep;
    unsigned thr, seed_lo, seed_hi;
};

__device__ __forceinline__ int acc_pos(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// 1 / (1 - p) if the (query, key) probability of (batch, head) bh is kept, else 0
__device__ __forceinline__ float drop_factor(const AttnTrainParams &p, unsigned bh, int q, int k) {
    if (p.thr == 0u) return 1.f;
    unsigned x = (unsigned)q * (unsigned)p.T + (unsigned)k;
    x ^= p.seed_lo; x *= 0x9E3779B1u; x ^= x >> 16;
    x += bh * 0x85EBCA6Bu + p.seed_hi; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x >= p.thr ? p.inv_keep : 0.f;
}

// A [channel][32 positions] tile on its way from global memory to LDS: 64 lanes take two rows per step (128-byte row segments), all
// 16 DT loads of a tile are in flight at once and stay in registers while the previous tile is being used (the walk over the other side's
// tiles is a chain of dependent global -> LDS -> MFMA steps of ONE wave: issued late, every tile would cost a full memory round trip).
template <int DT>
struct TileRegs {
    float v[16 * DT];
};

// the (batch, head) slice [dk][T] of a tensor as a buffer resource: rows >= dk are out of range and read as zero
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slice_rsrc(const float *src, int dk, int T) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, dk * T * 4, 0x00020000);
}

// per-lane offset + one SCALAR offset per row pair: no per-row address registers (48 hoisted 64-bit pointers per tile otherwise)
template <int DT>
__device__ __forceinline__ void tile_load(TileRegs<DT> &r, __amdgpu_buffer_rsrc_t src, int t0, int T, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    const int voff = (t0 + l31 < T) ? (half * T + t0 + l31) * 4 : 0x7ffffff0;      // positions >= T: out of range -> 0
#pragma unroll
    for (int it = 0; it < 16 * DT; ++it)
        r.v[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src, voff, 2 * it * T * 4, 0));
}

template <int DT>
__device__ __forceinline__ void tile_store(float *dst, const TileRegs<DT> &r, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int it = 0; it < 16 * DT; ++it) dst[(2 * it + half) * TP + l31] = r.v[it];
}

template <int DT>
__device__ __forceinline__ void stage_tile(float *dst, const float *src, int t0, int T, int dk, int lane) {
    TileRegs<DT> r;
    tile_load<DT>(r, slice_rsrc(src, dk, T), t0, T, lane);
    tile_store<DT>(dst, r, lane);
}

// acc[i][j] = sum_c Rt[c][i] * Lt[c][j]   (i on the accumulator rows, j on the lanes)
template <int DT>
__device__ __forceinline__ f32x16 gemm_cc(const float *Rt, const float *Lt, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
    for (int s = 0; s < 16 * DT; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Rt[(2 * s + half) * TP + l31], Lt[(2 * s + half) * TP + l31], acc, 0, 0, 0);
    return acc;
}

// out[ct][c][j] += sum_i Mt[32 ct + c][i] * X[i][j]   with X in accumulator layout (register s of half h <-> i = acc_pos(s, h))
template <int DT>
__device__ __forceinline__ void gemm_rx(f32x16 (&out)[DT], const float *Mt, const f32x16 &X, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
        for (int s = 0; s < 16; ++s)
            out[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(Mt[(32 * ct + l31) * TP + acc_pos(s, half)], X[s], out[ct], 0, 0, 0);
}

// per-query dots with a relative table: res[r] = f * sum_c Xt[c][l31] * tab[r][c]   (both halves split the channels, then exchange)
template <int DT, int RM>
__device__ __forceinline__ void rel_dots(float (&res)[RM], const float *Xt, const float *tab, int R, float f, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        float a = 0.f;
        if (r < R) {
            for (int cc = 0; cc < 16 * DT; ++cc) a += Xt[(2 * cc + half) * TP + l31] * tab[r * (32 * DT) + 2 * cc + half];
            a += __shfl_xor(a, 32);
        }
        res[r] = a * f;
    }
}

template <int DT>
__device__ __forceinline__ void stage_rel(float *dst, const float *tab, int h, const AttnTrainParams &p, int lane) {
    // dst[r][c] (row pitch 32 DT, zero padded) from tab[(h or 0), r, c]
    const float *src = tab + (long long)(p.nh_rel == 1 ? 0 : h) * p.R * p.dk;
    for (int i = lane; i < ATT_MAXREL * 32 * DT; i += 64) {
        const int r = i / (32 * DT), c = i % (32 * DT);
        dst[i] = (r < p.R && c < p.dk) ? src[r * p.dk + c] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------------ forward
// RM: register rows kept per query for the relative band (9 covers the reference's window of 4; 16 = ATT_MAXREL)
template <int DT, int RM>
__global__ void __launch_bounds__(64) relattn_train_fwd_kernel(const AttnTrainParams p) {
    constexpr int DKR = 32 * DT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *const Qt = smem, *const Kt = Qt + DKR * TP, *const Vt = Kt + DKR * TP;
    float *const relk_s = Vt + DKR * TP, *const relv_s = relk_s + ATT_MAXREL * DKR, *const maskt = relv_s + ATT_MAXREL * DKR;
    const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    const unsigned bh = (unsigned)(b * p.nh + h);
    const long long hoff = (long long)h * p.dk * p.T;
    const float *const qb = p.q + (long long)b * p.bs + hoff, *const kb = p.k + (long long)b * p.bs + hoff, *const vb = p.v + (long long)b * p.bs + hoff;
    const bool rel = p.ws >= 0;
    stage_tile<DT>(Qt, qb, q0, p.T, p.dk, lane);
    if (rel) { stage_rel<DT>(relk_s, p.rel_k, h, p, lane); stage_rel<DT>(relv_s, p.rel_v, h, p, lane); }
    __syncthreads();
    const int query = q0 + l31;
    const bool qvalid = query < p.T;
    const float mq = (p.mask && qvalid) ? p.mask[(long long)b * p.T + query] : 1.f;
    float qr[RM], sband[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) { qr[r] = 0.f; sband[r] = NEG_BIG; }
    if (rel) rel_dots<DT, RM>(qr, Qt, relk_s, p.R, p.scale, lane);

    f32x16 acc_o[DT];
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[ct][r] = 0.f;
    float m = NEG_BIG, l = 0.f;
    const int nkt = (p.T + 31) / 32;
    TileRegs<DT> rk, rv;
    const __amdgpu_buffer_rsrc_t ksrc = slice_rsrc(kb, p.dk, p.T), vsrc = slice_rsrc(vb, p.dk, p.T);
    tile_load<DT>(rk, ksrc, 0, p.T, lane);
    tile_load<DT>(rv, vsrc, 0, p.T, lane);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32;
        __syncthreads();
        tile_store<DT>(Kt, rk, lane);
        tile_store<DT>(Vt, rv, lane);
        if (lane < 32) maskt[lane] = (p.mask && k0 + lane < p.T) ? p.mask[(long long)b * p.T + k0 + lane] : 1.f;
        __syncthreads();
        if (kt + 1 < nkt) {
            tile_load<DT>(rk, ksrc, k0 + 32, p.T, lane);
            tile_load<DT>(rv, vsrc, k0 + 32, p.T, lane);
        }
        f32x16 s = gemm_cc<DT>(Kt, Qt, lane);
        const bool near = rel && (k0 - q0 <= 31 + p.ws) && (q0 - k0 <= 31 + p.ws);
        float mt = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kp = acc_pos(r, half), key = k0 + kp;
            const bool kvalid = key < p.T;
            float v = s[r] * p.scale;
            const int d = key - query + p.ws;
            if (near) {
#pragma unroll
                for (int rr = 0; rr < RM; ++rr) v += (d == rr) ? qr[rr] : 0.f;
            }
            if (mq * maskt[kp] == 0.f) v = -1e4f;
            if (near) {
#pragma unroll
                for (int rr = 0; rr < RM; ++rr) sband[rr] = (d == rr && kvalid) ? v : sband[rr];
            }
            v = kvalid ? v : NEG_BIG;
            s[r] = v;
            mt = fmaxf(mt, v);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m, mt);
        const float alpha = expf(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + acc_pos(r, half);
            const float pr = (key < p.T) ? expf(s[r] - m_new) : 0.f;
            psum += pr;
            s[r] = pr * drop_factor(p, bh, query, key);
        }
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[ct][r] *= alpha;
        gemm_rx<DT>(acc_o, Vt, s, lane);
    }
    const float l_tot = l + __shfl_xor(l, 32);
    const float inv_l = 1.f / l_tot;
    float pb[RM];
#pragma unroll
    for (int rr = 0; rr < RM; ++rr) {
        // both halves saw different keys of the band: the one that holds the score has sband > NEG_BIG
        const float sb = fmaxf(sband[rr], __shfl_xor(sband[rr], 32));
        const int key = query + rr - p.ws;
        const bool ok = rel && rr < p.R && key >= 0 && key < p.T;
        pb[rr] = ok ? expf(sb - m) * inv_l * drop_factor(p, bh, query, key) : 0.f;
    }
    if (qvalid) {
        float *const ob = p.out + (long long)b * p.out_bs + hoff;
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * ct + acc_pos(r, half);
                if (c < p.dk) {
                    float o = acc_o[ct][r] * inv_l;
                    if (rel) {
#pragma unroll
                        for (int rr = 0; rr < RM; ++rr) o += pb[rr] * relv_s[rr * DKR + c];
                    }
                    ob[(long long)c * p.T + query] = o;
                }
            }
        if (half == 0) {
            p.lse[(long long)bh * p.T + query] = m;
            p.lse[(long long)p.B * p.nh * p.T + (long long)bh * p.T + query] = logf(l_tot);
        }
    }
}

// ------------------------------------------------------------------------------------------ backward, query owner: dQ, d rel_k, d rel_v
// RM: register rows kept per query for the relative band (9 covers the reference's window of 4; 16 = ATT_MAXREL)
template <int DT, int RM>
__global__ void __launch_bounds__(64) relattn_train_bwd_q_kernel(const AttnTrainParams p) {
    constexpr int DKR = 32 * DT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *const Qt = smem, *const Gt = Qt + DKR * TP, *const Kt = Gt + DKR * TP, *const Vt = Kt + DKR * TP;
    float *const relk_s = Vt + DKR * TP, *const relv_s = relk_s + ATT_MAXREL * DKR, *const maskt = relv_s + ATT_MAXREL * DKR;
    float *const bds = maskt + 32, *const bpb = bds + ATT_MAXREL * 32;
    const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
    const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    const unsigned bh = (unsigned)(b * p.nh + h);
    const long long hoff = (long long)h * p.dk * p.T;
    const float *const qb = p.q + (long long)b * p.bs + hoff, *const kb = p.k + (long long)b * p.bs + hoff, *const vb = p.v + (long long)b * p.bs + hoff;
    const float *const gb = p.dout + (long long)b * p.out_bs + hoff, *const ob = p.out + (long long)b * p.out_bs + hoff;
    const bool rel = p.ws >= 0;
    stage_tile<DT>(Qt, qb, q0, p.T, p.dk, lane);
    stage_tile<DT>(Gt, gb, q0, p.T, p.dk, lane);
    if (rel) { stage_rel<DT>(relk_s, p.rel_k, h, p, lane); stage_rel<DT>(relv_s, p.rel_v, h, p, lane); }
    __syncthreads();
    const int query = q0 + l31;
    const bool qvalid = query < p.T;
    const float mq = (p.mask && qvalid) ? p.mask[(long long)b * p.T + query] : 1.f;
    const float m_q = qvalid ? p.lse[(long long)bh * p.T + query] : 0.f;
    const float ll_q = qvalid ? p.lse[(long long)p.B * p.nh * p.T + (long long)bh * p.T + query] : 0.f;
    float Dq = 0.f;
    if (qvalid) {
        for (int cc = 0; cc < 16 * DT; ++cc) {
            const int c = 2 * cc + half;
            if (c < p.dk) Dq += Gt[c * TP + l31] * ob[(long long)c * p.T + query];
        }
    }
    Dq += __shfl_xor(Dq, 32);
    float qr[RM], dor[RM], dsb[RM], pbb[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) { qr[r] = 0.f; dor[r] = 0.f; dsb[r] = 0.f; pbb[r] = 0.f; }
    if (rel) {
        rel_dots<DT, RM>(qr, Qt, relk_s, p.R, p.scale, lane);
        rel_dots<DT, RM>(dor, Gt, relv_s, p.R, 1.f, lane);
    }
    if (half == 0 && qvalid) {
        p.dvec[(long long)bh * p.T + query] = Dq;
        if (rel) {
#pragma unroll
            for (int rr = 0; rr < RM; ++rr)
                if (rr < p.R) {
                    p.qr[((long long)bh * p.T + query) * p.R + rr] = qr[rr];
                    p.dor[((long long)bh * p.T + query) * p.R + rr] = dor[rr];
                }
        }
    }

    f32x16 acc_dq[DT];
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_dq[ct][r] = 0.f;
    const int nkt = (p.T + 31) / 32;
    TileRegs<DT> rk, rv;
    const __amdgpu_buffer_rsrc_t ksrc = slice_rsrc(kb, p.dk, p.T), vsrc = slice_rsrc(vb, p.dk, p.T);
    tile_load<DT>(rk, ksrc, 0, p.T, lane);
    tile_load<DT>(rv, vsrc, 0, p.T, lane);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32;
        __syncthreads();
        tile_store<DT>(Kt, rk, lane);
        tile_store<DT>(Vt, rv, lane);
        if (lane < 32) maskt[lane] = (p.mask && k0 + lane < p.T) ? p.mask[(long long)b * p.T + k0 + lane] : 1.f;
        __syncthreads();
        if (kt + 1 < nkt) {
            tile_load<DT>(rk, ksrc, k0 + 32, p.T, lane);
            tile_load<DT>(rv, vsrc, k0 + 32, p.T, lane);
        }
        f32x16 s = gemm_cc<DT>(Kt, Qt, lane);
        const f32x16 dp = gemm_cc<DT>(Vt, Gt, lane);
        const bool near = rel && (k0 - q0 <= 31 + p.ws) && (q0 - k0 <= 31 + p.ws);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kp = acc_pos(r, half), key = k0 + kp;
            const bool kvalid = key < p.T;
            float v = s[r] * p.scale, dpv = dp[r];
            const int d = key - query + p.ws;
            if (near) {
#pragma unroll
                for (int rr = 0; rr < RM; ++rr) {
                    v += (d == rr) ? qr[rr] : 0.f;
                    dpv += (d == rr) ? dor[rr] : 0.f;
                }
            }
            const bool masked = (mq * maskt[kp] == 0.f);
            if (masked) v = -1e4f;
            const float pr = (kvalid && qvalid) ? expf((v - m_q) - ll_q) : 0.f;
            const float f = drop_factor(p, bh, query, key);
            const float ds = masked ? 0.f : pr * (dpv * f - Dq);
            if (near) {
#pragma unroll
                for (int rr = 0; rr < RM; ++rr) {
                    dsb[rr] += (d == rr) ? ds : 0.f;
                    pbb[rr] += (d == rr) ? pr * f : 0.f;
                }
            }
            s[r] = ds;
        }
        gemm_rx<DT>(acc_dq, Kt, s, lane);
    }
#pragma unroll
    for (int rr = 0; rr < RM; ++rr) {
        dsb[rr] += __shfl_xor(dsb[rr], 32);
        pbb[rr] += __shfl_xor(pbb[rr], 32);
    }
    if (qvalid) {
        float *const dqb = p.gq + (long long)b * p.g_bs + hoff;
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * ct + acc_pos(r, half);
                if (c < p.dk) {
                    float o = acc_dq[ct][r];
                    if (rel) {
#pragma unroll
                        for (int rr = 0; rr < RM; ++rr) o += dsb[rr] * relk_s[rr * DKR + c];
                    }
                    dqb[(long long)c * p.T + query] = o * p.scale;
                }
            }
    }
    if (rel) {
        // d rel_k[r][c] += scale * sum_q ds[q, q + r - w] Q[c][q],  d rel_v[r][c] += sum_q pt[q, q + r - w] dO[c][q]: this tile's partial sums
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int rr = 0; rr < RM; ++rr) { bds[rr * 32 + l31] = dsb[rr]; bpb[rr * 32 + l31] = pbb[rr]; }
        }
        __syncthreads();
        const long long wg = ((long long)bh * gridDim.x + blockIdx.x) * p.R * p.dk;
        for (int i = lane; i < p.R * p.dk; i += 64) {
            const int rr = i / p.dk, c = i % p.dk;
            float sk = 0.f, sv = 0.f;
            for (int j = 0; j < 32; ++j) {
                sk += bds[rr * 32 + j] * Qt[c * TP + j];
                sv += bpb[rr * 32 + j] * Gt[c * TP + j];
            }
            p.drelk_part[wg + i] = sk * p.scale;
            p.drelv_part[wg + i] = sv;
        }
    }
}

// ----------------------------------------------------------------------------------------------- backward, key owner: dK, dV
// RM: register rows kept per query for the relative band (9 covers the reference's window of 4; 16 = ATT_MAXREL)
template <int DT, int RM>
__global__ void __launch_bounds__(64) relattn_train_bwd_kv_kernel(const AttnTrainParams p) {
    constexpr int DKR = 32 * DT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *const Kt = smem, *const Vt = Kt + DKR * TP, *const Qt = Vt + DKR * TP, *const Gt = Qt + DKR * TP;
    float *const lset = Gt + DKR * TP, *const llt = lset + 32, *const dt = llt + 32, *const mqt = dt + 32, *const qrt = mqt + 32, *const dort = qrt + 32 * RP;
    const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
    const int k0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
    const unsigned bh = (unsigned)(b * p.nh + h);
    const long long hoff = (long long)h * p.dk * p.T;
    const float *const qb = p.q + (long long)b * p.bs + hoff, *const kb = p.k + (long long)b * p.bs + hoff, *const vb = p.v + (long long)b * p.bs + hoff;
    const float *const gb = p.dout + (long long)b * p.out_bs + hoff;
    const bool rel = p.ws >= 0;
    stage_tile<DT>(Kt, kb, k0, p.T, p.dk, lane);
    stage_tile<DT>(Vt, vb, k0, p.T, p.dk, lane);
    const int key = k0 + l31;
    const bool kvalid = key < p.T;
    const float mk = (p.mask && kvalid) ? p.mask[(long long)b * p.T + key] : 1.f;
    f32x16 acc_dk[DT], acc_dv[DT];
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_dk[ct][r] = 0.f; acc_dv[ct][r] = 0.f; }
    const int nqt = (p.T + 31) / 32;
    TileRegs<DT> rq, rg;
    const __amdgpu_buffer_rsrc_t qsrc = slice_rsrc(qb, p.dk, p.T), gsrc = slice_rsrc(gb, p.dk, p.T);
    tile_load<DT>(rq, qsrc, 0, p.T, lane);
    tile_load<DT>(rg, gsrc, 0, p.T, lane);
    for (int qt = 0; qt < nqt; ++qt) {
        const int q0 = qt * 32;
        __syncthreads();
        tile_store<DT>(Qt, rq, lane);
        tile_store<DT>(Gt, rg, lane);
        if (lane < 32) {
            const bool ok = q0 + lane < p.T;
            lset[lane] = ok ? p.lse[(long long)bh * p.T + q0 + lane] : 0.f;
            llt[lane] = ok ? p.lse[(long long)p.B * p.nh * p.T + (long long)bh * p.T + q0 + lane] : 0.f;
            dt[lane] = ok ? p.dvec[(long long)bh * p.T + q0 + lane] : 0.f;
            mqt[lane] = (p.mask && ok) ? p.mask[(long long)b * p.T + q0 + lane] : 1.f;
        }
        const bool near = rel && (k0 - q0 <= 31 + p.ws) && (q0 - k0 <= 31 + p.ws);
        if (near) {
            for (int i = lane; i < 32 * p.R; i += 64) {
                const int qq = i / p.R, rr = i % p.R;
                const bool ok = q0 + qq < p.T;
                qrt[qq * RP + rr] = ok ? p.qr[((long long)bh * p.T + q0 + qq) * p.R + rr] : 0.f;
                dort[qq * RP + rr] = ok ? p.dor[((long long)bh * p.T + q0 + qq) * p.R + rr] : 0.f;
            }
        }
        __syncthreads();
        if (qt + 1 < nqt) {
            tile_load<DT>(rq, qsrc, q0 + 32, p.T, lane);
            tile_load<DT>(rg, gsrc, q0 + 32, p.T, lane);
        }
        f32x16 s = gemm_cc<DT>(Qt, Kt, lane);           // rows: queries, lanes: keys
        f32x16 dp = gemm_cc<DT>(Gt, Vt, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = acc_pos(r, half), query = q0 + qi;
            const bool qvalid = query < p.T;
            float v = s[r] * p.scale, dpv = dp[r];
            const int d = key - query + p.ws;
            if (near && d >= 0 && d < p.R) {
                v += qrt[qi * RP + d];
                dpv += dort[qi * RP + d];
            }
            const bool masked = (mqt[qi] * mk == 0.f);
            if (masked) v = -1e4f;
            const float pr = (kvalid && qvalid) ? expf((v - lset[qi]) - llt[qi]) : 0.f;
            const float f = drop_factor(p, bh, query, key);
            s[r] = masked ? 0.f : pr * (dpv * f - dt[qi]);
            dp[r] = pr * f;
        }
        gemm_rx<DT>(acc_dv, Gt, dp, lane);
        gemm_rx<DT>(acc_dk, Qt, s, lane);
    }
    if (kvalid) {
        float *const dkb = p.gk + (long long)b * p.g_bs + hoff, *const dvb = p.gv + (long long)b * p.g_bs + hoff;
#pragma unroll
        for (int ct = 0; ct < DT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * ct + acc_pos(r, half);
                if (c < p.dk) {
                    dkb[(long long)c * p.T + key] = acc_dk[ct][r] * p.scale;
                    dvb[(long long)c * p.T + key] = acc_dv[ct][r];
                }
            }
    }
}

template <int DT>
static int launch_fwd(const AttnTrainParams &p, hipStream_t s) {
    const size_t lds = (size_t)(3 * 32 * DT * TP + 2 * ATT_MAXREL * 32 * DT + 32) * sizeof(float);
    auto k9 = relattn_train_fwd_kernel<DT, 9>;
    auto k16 = relattn_train_fwd_kernel<DT, ATT_MAXREL>;
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)k9, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)k16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.T, 32), (unsigned)p.nh, (unsigned)p.B);
    if (p.R <= 9) hipLaunchKernelGGL(k9, grid, dim3(64), lds, s, p);
    else hipLaunchKernelGGL(k16, grid, dim3(64), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

template <int DT>
static int launch_bwd(const AttnTrainParams &p, hipStream_t s) {
    const size_t lds_q = (size_t)(4 * 32 * DT * TP + 2 * ATT_MAXREL * 32 * DT + 32 + 2 * ATT_MAXREL * 32) * sizeof(float);
    const size_t lds_k = (size_t)(4 * 32 * DT * TP + 4 * 32 + 2 * 32 * RP) * sizeof(float);
    auto kq9 = relattn_train_bwd_q_kernel<DT, 9>;
    auto kq16 = relattn_train_bwd_q_kernel<DT, ATT_MAXREL>;
    auto kk = relattn_train_bwd_kv_kernel<DT, ATT_MAXREL>;
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kq9, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kq16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.T, 32), (unsigned)p.nh, (unsigned)p.B);
    // the query-owner kernel writes D / QR / DOR, which the key-owner kernel reads: same stream, in order
    if (p.R <= 9) hipLaunchKernelGGL(kq9, grid, dim3(64), lds_q, s, p);
    else hipLaunchKernelGGL(kq16, grid, dim3(64), lds_q, s, p);
    VS_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(kk, grid, dim3(64), lds_k, s, p);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

static int fill_params(AttnTrainParams &p, const float *q, const float *k, const float *v, int64_t bs, const float *rel_k, const float *rel_v,
                       const float *mask, int64_t B, int n_heads, int k_channels, int64_t T, int window_size, int n_heads_rel, float p_drop,
                       uint64_t seed, const char *who) {
    VS_REQUIRE(q && k && v, "%s: NULL tensor", who);
    VS_REQUIRE(B > 0 && B <= 65535 && n_heads > 0 && n_heads <= 65535 && k_channels > 0 && T > 0 && T <= 65535, "%s: bad dims", who);
    VS_REQUIRE(k_channels <= 128, "%s: heads of %d channels are not supported (<= 128)", who, k_channels);
    VS_REQUIRE(window_size < 0 || (rel_k && rel_v), "%s: window given but relative embeddings are NULL", who);
    VS_REQUIRE(window_size < 0 || 2 * window_size + 1 <= ATT_MAXREL, "%s: window_size %d too large", who, window_size);
    VS_REQUIRE(n_heads_rel == 1 || n_heads_rel == n_heads, "%s: bad n_heads_rel", who);
    VS_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "%s: dropout probability %g", who, (double)p_drop);
    memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v;
    p.bs = bs ? bs : (long long)n_heads * k_channels * T;
    p.rel_k = window_size >= 0 ? rel_k : nullptr;
    p.rel_v = window_size >= 0 ? rel_v : nullptr;
    p.mask = mask;
    p.B = (int)B; p.nh = n_heads; p.dk = k_channels; p.T = (int)T; p.ws = window_size; p.nh_rel = n_heads_rel;
    p.R = window_size >= 0 ? 2 * window_size + 1 : 0;
    p.scale = 1.0f / sqrtf((float)k_channels);
    p.thr = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
    p.inv_keep = 1.f / (1.f - p_drop);
    p.seed_lo = (unsigned)(seed & 0xffffffffu);
    p.seed_hi = (unsigned)(seed >> 32);
    return VS_OK;
}

}  // namespace vs

using namespace vs;

extern "C" {

int vs_relattn_train_fwd(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k, const float *rel_v,
                         const float *mask, float *out, int64_t out_batch_stride, float *lse, int64_t B, int n_heads, int k_channels, int64_t T,
                         int window_size, int n_heads_rel, float p_drop, uint64_t seed, void *stream) {
    AttnTrainParams p;
    VS_TRY(fill_params(p, q, k, v, qkv_batch_stride, rel_k, rel_v, mask, B, n_heads, k_channels, T, window_size, n_heads_rel, p_drop, seed,
                       "vs_relattn_train_fwd"));
    VS_REQUIRE(out && lse, "vs_relattn_train_fwd: NULL output");
    p.out = out; p.lse = lse;
    p.out_bs = out_batch_stride ? out_batch_stride : (long long)n_heads * k_channels * T;
    hipStream_t s = as_stream(stream);
    switch ((int)ceil_div(k_channels, 32)) {
        case 1: return launch_fwd<1>(p, s);
        case 2: return launch_fwd<2>(p, s);
        case 3: return launch_fwd<3>(p, s);
        default: return launch_fwd<4>(p, s);
    }
}

int vs_relattn_train_bwd(const float *q, const float *k, const float *v, int64_t qkv_batch_stride, const float *rel_k, const float *rel_v,
                         const float *mask, const float *out, const float *dout, int64_t out_batch_stride, const float *lse, float *dq, float *dk,
                         float *dv, int64_t grad_batch_stride, float *work, float *drel_k_part, float *drel_v_part, int64_t B, int n_heads,
                         int k_channels, int64_t T, int window_size, int n_heads_rel, float p_drop, uint64_t seed, void *stream) {
    AttnTrainParams p;
    VS_TRY(fill_params(p, q, k, v, qkv_batch_stride, rel_k, rel_v, mask, B, n_heads, k_channels, T, window_size, n_heads_rel, p_drop, seed,
                       "vs_relattn_train_bwd"));
    VS_REQUIRE(out && dout && lse && dq && dk && dv && work, "vs_relattn_train_bwd: NULL tensor");
    VS_REQUIRE(window_size < 0 || (drel_k_part && drel_v_part), "vs_relattn_train_bwd: NULL buffers for the relative-embedding gradients");
    p.out = const_cast<float *>(out); p.dout = dout; p.lse = const_cast<float *>(lse);
    p.out_bs = out_batch_stride ? out_batch_stride : (long long)n_heads * k_channels * T;
    p.gq = dq; p.gk = dk; p.gv = dv;
    p.g_bs = grad_batch_stride ? grad_batch_stride : (long long)n_heads * k_channels * T;
    const long long rows = (long long)B * n_heads * T;
    p.dvec = work; p.qr = work + rows; p.dor = p.qr + rows * p.R;      // work: B * nh * T * (1 + 2 R) floats
    p.drelk_part = drel_k_part; p.drelv_part = drel_v_part;
    hipStream_t s = as_stream(stream);
    switch ((int)ceil_div(k_channels, 32)) {
        case 1: return launch_bwd<1>(p, s);
        case 2: return launch_bwd<2>(p, s);
        case 3: return launch_bwd<3>(p, s);
        default: return launch_bwd<4>(p, s);
    }
}

}  // extern "C"
