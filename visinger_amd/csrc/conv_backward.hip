// conv_backward.hip -- weight gradient of the 1-D convolutions of the training step on gfx950.
//
//     gw[co, ci, k] = sum_{b, t} gy[b, co, t] * x[b, ci, t + k*dil - pad]          (x = 0 outside [0, T_in))
//
// A GEMM with M = C_out, N = C_in * K and a reduction over B * T_out positions -- long and thin, the transpose of the
// forward engine's shape.  One workgroup owns a 32 x 32 (co, ci) tile for ALL taps and a slice of the (b, t) reduction:
// per unit of 256 positions it stages the gy tile [32 x 256] and the x tile [32 x (256 + span)] in LDS (odd row pitch: the
// fragment reads walk down a column of rows), and every tap reads the SAME x tile at a shifted column, exactly like the
// forward engine.  The four waves split the taps (k = wave, wave + 4, ...: up to 4 accumulator tiles each) or, for convs of
// <= 4 taps, the positions of the unit; the gy fragment is read once per wave and reduction step and feeds up to 4
// exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32).
// Every (reduction slice, wave plane) writes its own partial gw; the caller sums the planes (a few hundred KB per conv): no
// atomics -- 512 workgroups adding 11 k values each into the same 11 k addresses (C = 32, k = 11) serialise in L2 -- and
// a run-to-run deterministic gradient.
//
// The grad-INPUT of a conv is a forward conv with reversed / transposed weights and runs on conv_engine.hip
// (visinger_amd/autograd.py::conv_backward); a transposed conv's weight gradient is this kernel with the roles of x and gy
// swapped over the de-interleaved phases of gy.
#include "conv_common.h"

#include <algorithm>
#include <cstdlib>

namespace vs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_TCH = 256;      // reduction positions per unit
constexpr int WG_MAXTAPS = 4;    // taps per wave (K <= 16)
constexpr int WG_MAXSPAN = 64;   // (K-1)*dil

struct WgradParams {
    const float *gy, *x;
    float *gw;                   // partial planes [slices * planes][Cout][Cin][K]
    int B, Cout, Cin, Tout, Tin, K, dil, pad;
    int units_per_item, units;   // ceil(Tout / WG_TCH), B * units_per_item
    long long plane_stride;      // floats between two partial planes: Cout * Cin * K (+ Cout with `bias`)
    int bias;                    // the bias gradient's partial sums ride behind each plane: plane[Cout * Cin * K + co] = sum over the slice's (b, t) of gy[b, co, t]
};

__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(const WgradParams p) {
    constexpr int PA = WG_TCH + 1;                    // gy tile row pitch
    constexpr int PX = WG_TCH + WG_MAXSPAN + 1;       // x tile row pitch (321)
    __shared__ float Ga[32 * PA];
    __shared__ float Xs[32 * PX];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lhalf = lane >> 5, l31 = lane & 31;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int span = (p.K - 1) * p.dil;
    const int W = WG_TCH + span;                      // staged x columns

    // Work split between the four waves: with more than 4 taps each wave takes the taps k = wave, wave + 4, ... over the whole
    // unit; with <= 4 taps (most convs of the path are 1x1) each wave takes ALL taps over a quarter of the unit's positions.
    const bool by_taps = p.K > WG_MAXTAPS;
    int ntap, tap0, tapstep, m_lo, m_hi;
    if (by_taps) {
        ntap = (p.K - wave + 3) / 4; tap0 = wave; tapstep = 4; m_lo = 0; m_hi = WG_TCH / 2;
    } else {
        ntap = p.K; tap0 = 0; tapstep = 1; m_lo = wave * (WG_TCH / 8); m_hi = m_lo + WG_TCH / 8;
    }
    f32x16 acc[WG_MAXTAPS];
#pragma unroll
    for (int i = 0; i < WG_MAXTAPS; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float bsum = 0.f;

    for (int u = blockIdx.z; u < p.units; u += gridDim.z) {
        const int b = u / p.units_per_item;
        const int t0 = (u - b * p.units_per_item) * WG_TCH;
        const float *gyb = p.gy + (long long)b * p.Cout * p.Tout;
        const float *xb = p.x + (long long)b * p.Cin * p.Tin;
        __syncthreads();                               // previous unit's fragments are consumed
        // staging: wave w loads rows w, w + 4, ...; a wave-instruction covers 64 consecutive positions of one row
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = wave + 4 * j;
            const int co = co0 + r, ci = ci0 + r;
            const float *grow = gyb + (long long)min(co, p.Cout - 1) * p.Tout;
            const float *xrow = xb + (long long)min(ci, p.Cin - 1) * p.Tin;
#pragma unroll
            for (int i = 0; i < WG_TCH / 64; ++i) {
                const int c = lane + 64 * i, t = t0 + c;
                const float v = grow[min(t, p.Tout - 1)];
                Ga[r * PA + c] = (co < p.Cout && t < p.Tout) ? v : 0.f;
            }
#pragma unroll
            for (int i = 0; i < (WG_TCH + WG_MAXSPAN + 63) / 64; ++i) {
                const int c = lane + 64 * i, n = t0 - p.pad + c;
                const float v = xrow[min(max(n, 0), p.Tin - 1)];
                if (c < W) Xs[r * PX + c] = (ci < p.Cin && n >= 0 && n < p.Tin) ? v : 0.f;
            }
        }
        __syncthreads();
        if (p.bias && blockIdx.x == 0) {               // (the workgroups of the first ci tile also sum their gy rows: thread -> row tid / 8, 32 of its positions)
            const float *gr = Ga + (tid >> 3) * PA + (tid & 7) * 32;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) bsum += gr[i];
        }
        const float *ga = Ga + l31 * PA + lhalf;
        const float *xs = Xs + l31 * PX + lhalf + tap0 * p.dil;
        const int tstep = tapstep * p.dil;
        for (int m = m_lo; m < m_hi; ++m) {
            const float a = ga[2 * m];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[2 * m], acc[0], 0, 0, 0);
            if (ntap > 1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[2 * m + tstep], acc[1], 0, 0, 0);
            if (ntap > 2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[2 * m + 2 * tstep], acc[2], 0, 0, 0);
            if (ntap > 3) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[2 * m + 3 * tstep], acc[3], 0, 0, 0);
        }
    }
    // acc[i][r] -> plane[co0 + row(r)][ci0 + l31][tap0 + i * tapstep]; plane = slice (taps split over the waves: the four
    // waves fill disjoint taps of one plane) or slice * 4 + wave (positions split: one plane per wave)
    float *plane = p.gw + (long long)(by_taps ? blockIdx.z : blockIdx.z * 4 + wave) * p.plane_stride;
    if (p.bias && blockIdx.x == 0) {
        // the slice's bias partial goes to its first plane; with one plane per wave the other three planes of the slice hold zeros
        bsum += __shfl_xor(bsum, 1); bsum += __shfl_xor(bsum, 2); bsum += __shfl_xor(bsum, 4);
        const int co = co0 + (tid >> 3);
        if ((tid & 7) == 0 && co < p.Cout) {
            float *bp = p.gw + (long long)(by_taps ? blockIdx.z : blockIdx.z * 4) * p.plane_stride + (long long)p.Cout * p.Cin * p.K + co;
            bp[0] = bsum;
            if (!by_taps) { bp[p.plane_stride] = 0.f; bp[2 * p.plane_stride] = 0.f; bp[3 * p.plane_stride] = 0.f; }
        }
    }
#pragma unroll
    for (int i = 0; i < WG_MAXTAPS; ++i) {
        const int k = tap0 + i * tapstep;
        if (i < ntap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * lhalf, ci = ci0 + l31;
                if (co < p.Cout && ci < p.Cin) plane[((long long)co * p.Cin + ci) * p.K + k] = acc[i][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same reduction on the bf16 matrix instruction in the split-bf16 x6 arithmetic of conv_split.hip (fp32 class), for the convs
// with 2..12 taps and >= 64 output channels that dominate the weight-gradient time of the training step (WaveNet k = 5, FFN k = 9,
// the MRF convs: tools/wgrad_breakdown.py measured the kernel above at 24-44 TFLOP/s on them).
//   D[co][ci] (one accumulator tile per tap) += sum over 16 consecutive positions:  A = gy[co][t .. t+15],  B = x[ci][t + k d - p ..]
// Both operands are contiguous along the contraction index t in memory, so a fragment is 8 consecutive positions of one row:
//   * gy: fp32 tile [128 co][64 t] in LDS; a wave (one 32-row co tile) reads its A fragment as two aligned ds_read_b128 and splits
//     it ONCE per 16-position step for all taps (44 VALU per 6 K MFMAs);
//   * x: split into bf16 planes while staged, [plane][32 ci][136 positions] row-major along t; a tap's B fragment starts at an
//     arbitrary position: five ds_read_b32 per plane from the 4-byte-aligned dword below it, and for odd offsets a v_alignbyte per
//     dword -- no transposition, no per-tap copy of the tile;
//   * a wave keeps one accumulator tile per tap (up to 12: 192 registers); workgroup = 128 co x 32 ci, 60 KB of LDS, two per CU;
//   * partial planes per reduction slice as above (deterministic, summed by the caller).
constexpr int WS_TU = 64;                     // positions per unit
constexpr int WS_GP = WS_TU + 4;              // gy tile row pitch in floats (272 B = 17 x 16: conflict-free ds_read_b128 down the rows)
constexpr int WS_XW = WS_TU + WG_MAXSPAN + 8; // staged x positions per row (136)
constexpr int WS_XP = WS_XW / 2 + 1;          // x plane row pitch in dwords (69: odd)
constexpr int WS_MAXK = 12;

// KT taps per wave; TS = 1: four co tiles per workgroup (128 rows), every wave all K <= KT taps; TS = 2 (K > 8): two co tiles (64 rows),
// the two waves of a tile take the taps [0, KT) and [KT, 2 KT) -- 12 accumulator tiles next to the fragments do not fit 256 registers.
template <int KT, int TS>
__global__ void __launch_bounds__(256, 2) conv_wgrad_split_kernel(const WgradParams p) {
    constexpr int CO_T = 128 / TS;
    __shared__ __attribute__((aligned(16))) float Gs[128 * WS_GP];
    __shared__ unsigned Xp[3 * 32 * WS_XP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lhalf = lane >> 5, l31 = lane & 31;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * CO_T;
    const int cot = wave / TS;                       // this wave's 32-row co tile within the workgroup
    const int tapb = (wave % TS) * KT;               // its first tap
    f32x16 acc[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = p.bias && blockIdx.x == 0;      // (the workgroups of the first ci tile also sum their gy rows: 256 / CO_T threads per row)

    // A unit's global loads -- the gy tile (CO_T rows x 16 float4; T_out % 4 == 0, rows 16-byte aligned: host-checked) and the x tile (32 rows x 68
    // position pairs) -- are ALL unconditional on clamped addresses (what lies outside is zeroed when the registers go to LDS) and are issued for unit
    // u + 1 right behind the barrier that publishes unit u's tiles, so that they fly under unit u's MFMAs.  (Round 4, found in the ISA: each of the
    // predicated float4 loads of the gy tile had its own `s_waitcnt vmcnt(0)` -- four to eight SERIAL round trips per unit in front of the x loads'
    // one, none of them overlapped with matrix work: 5-15 us of latency per unit against ~2 us of MFMAs.)
    constexpr int GIT = 8 / TS;
    constexpr int XPAIRS = 32 * (WS_XW / 2), XIT = (XPAIRS + 255) / 256;
    float4 gv[GIT];
    float xv[XIT][2];
    auto load_g = [&](int u, int i0, int i1) __attribute__((always_inline)) {
        const int b = u / p.units_per_item;
        const int t0 = (u - b * p.units_per_item) * WS_TU;
        const float *gyb = p.gy + (long long)b * p.Cout * p.Tout;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int e = tid + 256 * i;
            const int row = e >> 4, c4 = (e & 15) * 4;
            const int co = min(co0 + row, p.Cout - 1), t = min(t0 + c4, p.Tout - 4);
            gv[i] = *reinterpret_cast<const float4 *>(gyb + (long long)co * p.Tout + t);
        }
    };
    auto load_x = [&](int u) __attribute__((always_inline)) {
        const int b = u / p.units_per_item;
        const int t0 = (u - b * p.units_per_item) * WS_TU;
        const float *xb = p.x + (long long)b * p.Cin * p.Tin;
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
            const int e = min(tid + 256 * i, XPAIRS - 1);
            const int row = e / (WS_XW / 2), pr = e - row * (WS_XW / 2);
            const int ci = min(ci0 + row, p.Cin - 1), n = t0 - p.pad + 2 * pr;
            const float *xr = xb + (long long)ci * p.Tin;
            xv[i][0] = xr[min(max(n, 0), p.Tin - 1)];
            xv[i][1] = xr[min(max(n + 1, 0), p.Tin - 1)];
        }
    };
    auto load_unit = [&](int u) __attribute__((always_inline)) { load_g(u, 0, GIT); load_x(u); };
    // (the prefetch registers -- 32 + 18 at TS = 1 -- fit next to at most 4 (TS = 1) / 5 (TS = 2) accumulator tiles; the other instances issue the unit's
    //  loads together at the top of the unit: one exposed round trip instead of five to nine; seven tiles: the gy tile in two halves, then the x tile -- three)
    constexpr bool PRE = (TS == 1) ? (KT <= 4) : (KT <= 5);
    constexpr bool TWO = (TS == 1 && KT >= 7);
    if constexpr (PRE) { if ((int)blockIdx.z < p.units) load_unit(blockIdx.z); }
    for (int u = blockIdx.z; u < p.units; u += gridDim.z) {
        const int b = u / p.units_per_item;
        const int t0 = (u - b * p.units_per_item) * WS_TU;
        if constexpr (!PRE) { if constexpr (TWO) load_g(u, 0, GIT / 2); else load_unit(u); }
        __syncthreads();                               // previous unit's fragments are consumed
        auto store_g = [&](int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
            for (int i = i0; i < i1; ++i) {
                const int e = tid + 256 * i;
                const int row = e >> 4, c4 = (e & 15) * 4;
                const bool ok = (co0 + row < p.Cout) && (t0 + c4 < p.Tout);
                *reinterpret_cast<float4 *>(Gs + row * WS_GP + c4) = ok ? gv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        if constexpr (TWO) { store_g(0, GIT / 2); load_g(u, GIT / 2, GIT); store_g(GIT / 2, GIT); }
        else store_g(0, GIT);
        // x tile: split exactly into three bf16 planes (one dword per pair and plane)
        if constexpr (TWO) load_x(u);
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
            const int e = tid + 256 * i;
            const int row = e / (WS_XW / 2), pr = e - row * (WS_XW / 2);
            const int n = t0 - p.pad + 2 * pr;
            const bool okr = (ci0 + row < p.Cin);
            const float v0 = (okr && n >= 0 && n < p.Tin) ? xv[i][0] : 0.f;
            const float v1 = (okr && n + 1 >= 0 && n + 1 < p.Tin) ? xv[i][1] : 0.f;
            unsigned d[3];
            split_pair<3>(v0, v1, d);
            if (e < XPAIRS) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) Xp[(pl * 32 + row) * WS_XP + pr] = d[pl];
            }
        }
        __syncthreads();
        if constexpr (PRE) { if (u + (int)gridDim.z < p.units) load_unit(u + gridDim.z); }
        if (do_bias) {
            constexpr int TPR = 256 / CO_T, NP = WS_TU / TPR;       // threads per row, positions per thread
            const float *gr = Gs + (tid / TPR) * WS_GP + (tid % TPR) * NP;
#pragma unroll
            for (int i = 0; i < NP; i += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(gr + i);
                // (scalar adds: as `(v.x + v.y) + (v.z + v.w)` this became v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0], the form conv_common.h describes)
                bsum += add_f32_scalar(add_f32_scalar(v.x, v.y), add_f32_scalar(v.z, v.w));
            }
        }
        const float *ga = Gs + (cot * 32 + l31) * WS_GP + 8 * lhalf;
#pragma unroll
        for (int ks = 0; ks < WS_TU / 16; ++ks) {
            // A: 8 consecutive positions of this lane's gy row, split once for all taps
            const float4 g0 = *reinterpret_cast<const float4 *>(ga + 16 * ks), g1 = *reinterpret_cast<const float4 *>(ga + 16 * ks + 4);
            unsigned a0[3], a1[3], a2[3], a3[3];
            split_pair<3>(g0.x, g0.y, a0); split_pair<3>(g0.z, g0.w, a1); split_pair<3>(g1.x, g1.y, a2); split_pair<3>(g1.z, g1.w, a3);
            u32x4 af[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { af[pl].x = a0[pl]; af[pl].y = a1[pl]; af[pl].z = a2[pl]; af[pl].w = a3[pl]; }
            // B: per tap the 3 x 5 dwords that hold its fragment (one tap ahead of the MFMAs), then v_alignbyte for odd offsets
            unsigned raw[2][3][5];
            auto load_raw = [&](unsigned (&dst)[3][5], int k) __attribute__((always_inline)) {
                const int q = (16 * ks + 8 * lhalf + (tapb + k) * p.dil) >> 1;      // 4-byte-aligned dword below the fragment's first position
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const unsigned *xr = Xp + (pl * 32 + l31) * WS_XP + q;
#pragma unroll
                    for (int i = 0; i < 5; ++i) dst[pl][i] = xr[i];
                }
            };
            load_raw(raw[0], 0);
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                if (k + 1 < KT) load_raw(raw[(k + 1) & 1], k + 1);
                if (tapb + k < p.K) {                                // (wave-uniform: only the last tap of the second tap-half can be absent)
                    const bool odd = (((tapb + k) * p.dil) & 1) != 0;             // wave-uniform
                    u32x4 bf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const unsigned(&d)[5] = raw[k & 1][pl];
                        bf[pl].x = odd ? __builtin_amdgcn_alignbyte(d[1], d[0], 2) : d[0];
                        bf[pl].y = odd ? __builtin_amdgcn_alignbyte(d[2], d[1], 2) : d[1];
                        bf[pl].z = odd ? __builtin_amdgcn_alignbyte(d[3], d[2], 2) : d[2];
                        bf[pl].w = odd ? __builtin_amdgcn_alignbyte(d[4], d[3], 2) : d[3];
                    }
                    auto mm = [&](int ta, int tb) __attribute__((always_inline)) {
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ta]), __builtin_bit_cast(bf16x8, bf[tb]),
                                                                        acc[k], 0, 0, 0);
                    };
                    mm(1, 1); mm(2, 0); mm(0, 2); mm(1, 0); mm(0, 1); mm(0, 0);      // smallest terms first
                }
            }
        }
    }
    float *plane = p.gw + (long long)blockIdx.z * p.plane_stride;
    if (do_bias) {
        constexpr int TPR = 256 / CO_T;
        bsum += __shfl_xor(bsum, 1);
        if constexpr (TPR == 4) bsum += __shfl_xor(bsum, 2);
        const int co = co0 + tid / TPR;
        if (tid % TPR == 0 && co < p.Cout) plane[(long long)p.Cout * p.Cin * p.K + co] = bsum;
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        if (tapb + k < p.K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf, ci = ci0 + l31;
                if (co < p.Cout && ci < p.Cin) plane[((long long)co * p.Cin + ci) * p.K + tapb + k] = acc[k][r];
            }
        }
    }
}

// out[i] = sum over the planes of plane[i], i < n: the weight gradient (and, behind it, the bias gradient) from the partial planes of the kernels
// above.  A workgroup owns 64 consecutive elements; its four waves take the planes z = w, w + 4, ... (eight loads in flight each: a thread that walked
// all the planes alone spent 16 us per launch on 30 dependent round trips -- 245 launches per training step), then one fixed-order LDS tree: the same
// bits every run.
__global__ void __launch_bounds__(256) wgrad_finish_kernel(const float *__restrict__ planes, long long stride, int nplanes, float *__restrict__ out, long long n) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    const long long ic = min(i, n - 1);
    float s0 = 0.f, s1 = 0.f;
    int z = w;
    for (; z + 28 < nplanes; z += 32) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = planes[(long long)(z + 4 * q) * stride + ic];
        s0 += (v[0] + v[1]) + (v[2] + v[3]);
        s1 += (v[4] + v[5]) + (v[6] + v[7]);
    }
    for (; z < nplanes; z += 4) s0 += planes[(long long)z * stride + ic];
    red[w][lane] = s0 + s1;
    __syncthreads();
    if (w == 0 && i < n) out[i] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

}  // namespace vs

using namespace vs;

extern "C" {

static int wgrad_slices(int64_t B, int64_t c_out, int64_t c_in, int64_t T_out) {
    const int64_t units = B * ceil_div(T_out, WG_TCH);
    const int64_t tiles = ceil_div(c_in, 32) * ceil_div(c_out, 32);
    return (int)std::max<int64_t>(1, std::min<int64_t>(units, ceil_div(1024, tiles)));
}

// the split-bf16 kernel: 1..12 taps, >= 64 output channels, float4-loadable gy rows, enough positions to fill its 64-position units
static bool wgrad_split_ok(int64_t B, int64_t c_out, int64_t T_out, int k) {
    return k >= 1 && k <= WS_MAXK && c_out >= 64 && (T_out % 4) == 0 && B * T_out >= 512 && !opt(OPT_NO_WGRAD_SPLIT);
}
static int wgrad_split_slices(int64_t B, int64_t c_out, int64_t c_in, int64_t T_out) {
    const int64_t units = B * ceil_div(T_out, WS_TU);
    const int64_t tiles = ceil_div(c_in, 32) * ceil_div(c_out, 128);      // (64-row tiles for K > 8: up to twice the workgroups)
    return (int)std::max<int64_t>(1, std::min<int64_t>(units, ceil_div(512, tiles)));
}

int vs_conv_wgrad_planes(int64_t B, int64_t c_out, int64_t c_in, int64_t T_out, int k) {
    if (B <= 0 || c_out <= 0 || c_in <= 0 || T_out <= 0 || k < 1) return 0;
    if (wgrad_split_ok(B, c_out, T_out, k)) return wgrad_split_slices(B, c_out, c_in, T_out);
    return wgrad_slices(B, c_out, c_in, T_out) * (k > WG_MAXTAPS ? 1 : 4);
}

static int wgrad_launch(const float *gy, const float *x, float *gw_planes, int64_t B, int64_t c_out, int64_t c_in, int64_t T_out,
                        int64_t T_in, int k, int dil, int pad, int bias, void *stream);

int vs_conv_wgrad(const float *gy, const float *x, float *gw_planes, int64_t B, int64_t c_out, int64_t c_in, int64_t T_out,
                  int64_t T_in, int k, int dil, int pad, void *stream) {
    return wgrad_launch(gy, x, gw_planes, B, c_out, c_in, T_out, T_in, k, dil, pad, 0, stream);
}

int vs_conv_wgrad_bias(const float *gy, const float *x, float *work, float *out, int with_bias, int64_t B, int64_t c_out, int64_t c_in, int64_t T_out,
                       int64_t T_in, int k, int dil, int pad, void *stream) {
    VS_REQUIRE(work && out, "vs_conv_wgrad_bias: NULL work / out");
    VS_TRY(wgrad_launch(gy, x, work, B, c_out, c_in, T_out, T_in, k, dil, pad, with_bias ? 1 : 0, stream));
    const long long n = (long long)c_out * c_in * k + (with_bias ? c_out : 0);
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(256), 0, as_stream(stream), work, n,
                       vs_conv_wgrad_planes(B, c_out, c_in, T_out, k), out, n);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

static int wgrad_launch(const float *gy, const float *x, float *gw_planes, int64_t B, int64_t c_out, int64_t c_in, int64_t T_out,
                        int64_t T_in, int k, int dil, int pad, int bias, void *stream) {
    VS_REQUIRE(gy && x && gw_planes && B > 0 && c_out > 0 && c_in > 0 && T_out > 0 && T_in > 0, "vs_conv_wgrad: bad arguments");
    VS_REQUIRE(k >= 1 && k <= 4 * WG_MAXTAPS && dil >= 1 && pad >= 0 && (k - 1) * dil <= WG_MAXSPAN,
               "vs_conv_wgrad: k = %d, dil = %d outside the supported range (k <= 16, (k-1)*dil <= 64)", k, dil);
    WgradParams p;
    p.gy = gy; p.x = x; p.gw = gw_planes;
    p.B = (int)B; p.Cout = (int)c_out; p.Cin = (int)c_in; p.Tout = (int)T_out; p.Tin = (int)T_in;
    p.K = k; p.dil = dil; p.pad = pad;
    p.bias = bias;
    p.plane_stride = (long long)c_out * c_in * k + (bias ? c_out : 0);
    if (wgrad_split_ok(B, c_out, T_out, k)) {
        // vs_conv_wgrad_planes sized `gw_planes` for THIS kernel's reduction slices from the shape alone: an unaligned gy cannot fall back
        // to conv_wgrad_kernel (a different plane count: it would write past the buffer or leave planes unwritten)
        VS_REQUIRE((reinterpret_cast<uintptr_t>(gy) & 15u) == 0,
                   "vs_conv_wgrad: gy must be 16-byte aligned for this shape (k = %d, c_out = %lld: float4 row loads); copy it to an aligned buffer",
                   k, (long long)c_out);
        p.units_per_item = (int)ceil_div(T_out, WS_TU);
        p.units = p.B * p.units_per_item;
        dim3 grid((unsigned)ceil_div(c_in, 32), (unsigned)ceil_div(c_out, k < 8 ? 128 : 64), (unsigned)wgrad_split_slices(B, c_out, c_in, T_out));
        hipStream_t s = as_stream(stream);
        switch (k) {      // KT = the taps a wave really has: no idle accumulator tiles, registers left for the one-tap-ahead reads
            case 1: hipLaunchKernelGGL((conv_wgrad_split_kernel<1, 1>), grid, dim3(256), 0, s, p); break;      // 1x1 convs: a plain [Cout x B*T] . [B*T x Cin] GEMM
            case 2: hipLaunchKernelGGL((conv_wgrad_split_kernel<2, 1>), grid, dim3(256), 0, s, p); break;
            case 3: hipLaunchKernelGGL((conv_wgrad_split_kernel<3, 1>), grid, dim3(256), 0, s, p); break;
            case 4: hipLaunchKernelGGL((conv_wgrad_split_kernel<4, 1>), grid, dim3(256), 0, s, p); break;
            case 5: hipLaunchKernelGGL((conv_wgrad_split_kernel<5, 1>), grid, dim3(256), 0, s, p); break;
            case 6: hipLaunchKernelGGL((conv_wgrad_split_kernel<6, 1>), grid, dim3(256), 0, s, p); break;
            case 7: hipLaunchKernelGGL((conv_wgrad_split_kernel<7, 1>), grid, dim3(256), 0, s, p); break;
            case 8: hipLaunchKernelGGL((conv_wgrad_split_kernel<4, 2>), grid, dim3(256), 0, s, p); break;      // (8 tiles + the look-ahead spill)
            case 9: case 10: hipLaunchKernelGGL((conv_wgrad_split_kernel<5, 2>), grid, dim3(256), 0, s, p); break;
            default: hipLaunchKernelGGL((conv_wgrad_split_kernel<6, 2>), grid, dim3(256), 0, s, p); break;
        }
        VS_CHECK_HIP(hipGetLastError());
        return VS_OK;
    }
    p.units_per_item = (int)ceil_div(T_out, WG_TCH);
    p.units = p.B * p.units_per_item;
    dim3 grid((unsigned)ceil_div(c_in, 32), (unsigned)ceil_div(c_out, 32), (unsigned)wgrad_slices(B, c_out, c_in, T_out));
    hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 0, as_stream(stream), p);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

}  // extern "C"
