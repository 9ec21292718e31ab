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
#include "vs_internal.h"

#include <algorithm>

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
    float *plane = p.gw + (long long)(by_taps ? blockIdx.z : blockIdx.z * 4 + wave) * p.Cout * p.Cin * p.K;
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

}  // namespace vs

using namespace vs;

extern "C" {

static int wgrad_slices(int64_t B, int64_t c_out, int64_t c_in, int64_t T_out) {
    const int64_t units = B * ceil_div(T_out, WG_TCH);
    const int64_t tiles = ceil_div(c_in, 32) * ceil_div(c_out, 32);
    return (int)std::max<int64_t>(1, std::min<int64_t>(units, ceil_div(1024, tiles)));
}

int vs_conv_wgrad_planes(int64_t B, int64_t c_out, int64_t c_in, int64_t T_out, int k) {
    if (B <= 0 || c_out <= 0 || c_in <= 0 || T_out <= 0 || k < 1) return 0;
    return wgrad_slices(B, c_out, c_in, T_out) * (k > WG_MAXTAPS ? 1 : 4);
}

int vs_conv_wgrad(const float *gy, const float *x, float *gw_planes, int64_t B, int64_t c_out, int64_t c_in, int64_t T_out,
                  int64_t T_in, int k, int dil, int pad, void *stream) {
    VS_REQUIRE(gy && x && gw_planes && B > 0 && c_out > 0 && c_in > 0 && T_out > 0 && T_in > 0, "vs_conv_wgrad: bad arguments");
    VS_REQUIRE(k >= 1 && k <= 4 * WG_MAXTAPS && dil >= 1 && pad >= 0 && (k - 1) * dil <= WG_MAXSPAN,
               "vs_conv_wgrad: k = %d, dil = %d outside the supported range (k <= 16, (k-1)*dil <= 64)", k, dil);
    WgradParams p;
    p.gy = gy; p.x = x; p.gw = gw_planes;
    p.B = (int)B; p.Cout = (int)c_out; p.Cin = (int)c_in; p.Tout = (int)T_out; p.Tin = (int)T_in;
    p.K = k; p.dil = dil; p.pad = pad;
    p.units_per_item = (int)ceil_div(T_out, WG_TCH);
    p.units = p.B * p.units_per_item;
    dim3 grid((unsigned)ceil_div(c_in, 32), (unsigned)ceil_div(c_out, 32), (unsigned)wgrad_slices(B, c_out, c_in, T_out));
    hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 0, as_stream(stream), p);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

}  // extern "C"
