// grouped_conv.hip -- the grouped, strided 1-D convolutions of the HiFi-GAN scale discriminator (SURVEY.md 8a row a13,
// reference modules/discriminator.py:55-60: Conv1d(16, 64, 41, 4, groups=4), (64, 256, groups=16), (256, 1024, groups=64),
// (1024, 1024, groups=256), all with 4 input channels per group) and their gradients, on gfx950.
//
// A group is a [C_out/G x 4*41] GEMV-sized problem: nothing for the matrix cores (a 32x32 MFMA tile would be > 85 % padding),
// and the whole discriminator stack is ~2.6 GMAC per (real, generated) pair, so these are VALU kernels laid out for coalesced
// HBM/L2 access; the dense convs of the discriminators run on the MFMA engine (conv_engine.hip).
//
//   y[b, co, n]   = bias[co] + sum_{ci in group(co), k} w[co, ci, k] * x[b, g*cig + ci, n*s + k - p]
//   gx[b, c, t]   = sum_{co in group(c), k : (t + p - k) % s == 0} w[co, c - g*cig, k] * gy[b, co, (t + p - k) / s]
//   gw[co, ci, k] = sum_{b, n} gy[b, co, n] * x[b, g*cig + ci, n*s + k - p]
#include "vs_internal.h"

namespace vs {

struct GConvParams {
    const float *x, *w, *bias, *gy;
    float *y, *gx, *gw;
    int B, Cin, Cout, T, Tout, K, stride, pad, groups, cig, cog;
};

// one thread per output element, n fastest (x reads of a wave: stride-s gather inside a few cache lines)
__global__ void __launch_bounds__(256) gconv_fwd_kernel(const GConvParams p) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.B * p.Cout * p.Tout;
    if (e >= total) return;
    const int n = (int)(e % p.Tout);
    const int co = (int)((e / p.Tout) % p.Cout);
    const int b = (int)(e / ((long long)p.Tout * p.Cout));
    const int g = co / p.cog;
    const float *wr = p.w + (long long)co * p.cig * p.K;
    const float *xb = p.x + ((long long)b * p.Cin + (long long)g * p.cig) * p.T;
    float acc = p.bias ? p.bias[co] : 0.f;
    const int t0 = n * p.stride - p.pad;
    const int k_lo = max(0, -t0), k_hi = min(p.K, p.T - t0);
    for (int ci = 0; ci < p.cig; ++ci) {
        const float *xr = xb + (long long)ci * p.T + t0;
        const float *wk = wr + ci * p.K;
        for (int k = k_lo; k < k_hi; ++k) acc = fmaf(wk[k], xr[k], acc);
    }
    p.y[e] = acc;
}

// COG outputs per thread: a thread owns output position n of one (batch item, group) and all COG output channels of the group, so every
// x value it loads feeds COG FMAs (the one-output kernel above issues two loads per FMA: 3 TFLOP/s, 4.3 ms of the config-3 training
// step); the weights of a group are wave-uniform (blockIdx.y = group) and come through the scalar cache.
template <int COG>
__global__ void __launch_bounds__(256) gconv_fwd_group_kernel(const GConvParams p) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, b = blockIdx.z;
    if (n >= p.Tout) return;
    const float *__restrict__ wg = p.w + (long long)g * COG * p.cig * p.K;
    const float *xb = p.x + ((long long)b * p.Cin + (long long)g * p.cig) * p.T;
    float acc[COG];
#pragma unroll
    for (int c = 0; c < COG; ++c) acc[c] = p.bias ? p.bias[g * COG + c] : 0.f;
    const int t0 = n * p.stride - p.pad;
    for (int ci = 0; ci < p.cig; ++ci) {
        const float *xr = xb + (long long)ci * p.T;
        for (int k = 0; k < p.K; ++k) {
            const int t = t0 + k;
            const float xv = (t >= 0 && t < p.T) ? xr[t] : 0.f;
#pragma unroll
            for (int c = 0; c < COG; ++c) acc[c] = fmaf(wg[((long long)c * p.cig + ci) * p.K + k], xv, acc[c]);
        }
    }
    float *yb = p.y + ((long long)b * p.Cout + (long long)g * COG) * p.Tout + n;
#pragma unroll
    for (int c = 0; c < COG; ++c) yb[(long long)c * p.Tout] = acc[c];
}

// one thread per input element, t fastest
__global__ void __launch_bounds__(256) gconv_bwd_data_kernel(const GConvParams p) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.B * p.Cin * p.T;
    if (e >= total) return;
    const int t = (int)(e % p.T);
    const int c = (int)((e / p.T) % p.Cin);
    const int b = (int)(e / ((long long)p.T * p.Cin));
    const int g = c / p.cig, ci = c - g * p.cig;
    const float *gyb = p.gy + ((long long)b * p.Cout + (long long)g * p.cog) * p.Tout;
    const int tp = t + p.pad;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;      // (four output channels at a time on four accumulators: eight loads in flight where one fmaf chain had two)
    for (int k = tp % p.stride; k < p.K; k += p.stride) {      // taps with (t + p - k) divisible by the stride
        const int n = (tp - k) / p.stride;
        if (tp - k < 0 || n >= p.Tout) continue;
        const float *wk = p.w + ((long long)g * p.cog * p.cig + ci) * p.K + k;
        const long long ws = (long long)p.cig * p.K;
        int co = 0;
        for (; co + 3 < p.cog; co += 4) {
            const float w0 = wk[(co + 0) * ws], w1 = wk[(co + 1) * ws], w2 = wk[(co + 2) * ws], w3 = wk[(co + 3) * ws];
            const float g0 = gyb[(long long)(co + 0) * p.Tout + n], g1 = gyb[(long long)(co + 1) * p.Tout + n];
            const float g2 = gyb[(long long)(co + 2) * p.Tout + n], g3 = gyb[(long long)(co + 3) * p.Tout + n];
            a0 = fmaf(w0, g0, a0); a1 = fmaf(w1, g1, a1); a2 = fmaf(w2, g2, a2); a3 = fmaf(w3, g3, a3);
        }
        for (; co < p.cog; ++co) a0 = fmaf(wk[co * ws], gyb[(long long)co * p.Tout + n], a0);
    }
    p.gx[e] = (a0 + a1) + (a2 + a3);
}

// one workgroup per (co, batch item): thread j < cig*K owns gw[co, ci, k]; gy[b, co, n] is a broadcast, the x reads of a
// wave are consecutive taps = consecutive addresses.  Partial sums over b are written as planes [B][Cout][cig*K]
// and summed by the caller (deterministic, no atomics).
__global__ void __launch_bounds__(256) gconv_bwd_weight_kernel(const GConvParams p) {
    const int co = blockIdx.x, b = blockIdx.y;
    const int j = threadIdx.x;
    const int nj = p.cig * p.K;
    const int g = co / p.cog;
    const float *gyr = p.gy + ((long long)b * p.Cout + co) * p.Tout;
    for (int jj = j; jj < nj; jj += 256) {
        const int ci = jj / p.K, k = jj - ci * p.K;
        const float *xr = p.x + ((long long)b * p.Cin + (long long)g * p.cig + ci) * p.T;
        // eight positions at a time on eight accumulators, every load unconditional on a clamped address (a single fmaf chain behind a bounds branch
        // had one load pair in flight per thread: 433 us per launch at T_out = 8 192; round 4).  Fixed order: the same bits every run.
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int n0 = 0; n0 < p.Tout; n0 += 8) {
            float gv[8], xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int n = min(n0 + q, p.Tout - 1);
                const int t = n * p.stride + k - p.pad;
                gv[q] = gyr[n];
                xv[q] = xr[min(max(t, 0), p.T - 1)];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int t = (n0 + q) * p.stride + k - p.pad;
                const bool ok = (n0 + q < p.Tout) && (t >= 0) && (t < p.T);
                a[q] = fmaf(gv[q], ok ? xv[q] : 0.f, a[q]);
            }
        }
        p.gw[((long long)b * p.Cout + co) * nj + jj] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
}

static int check(const GConvParams &p) {
    VS_REQUIRE(p.B > 0 && p.Cin > 0 && p.Cout > 0 && p.T > 0 && p.K > 0 && p.stride > 0 && p.pad >= 0 && p.groups > 0,
               "vs_gconv1d: bad dims");
    VS_REQUIRE(p.Cin % p.groups == 0 && p.Cout % p.groups == 0, "vs_gconv1d: channels not divisible by groups");
    VS_REQUIRE(p.Tout == (p.T + 2 * p.pad - p.K) / p.stride + 1 && p.Tout > 0, "vs_gconv1d: inconsistent output length");
    return VS_OK;
}

static GConvParams make(int64_t B, int64_t c_in, int64_t c_out, int64_t T, int k, int stride, int pad, int groups) {
    GConvParams p;
    memset(&p, 0, sizeof(p));
    p.B = (int)B; p.Cin = (int)c_in; p.Cout = (int)c_out; p.T = (int)T; p.K = k; p.stride = stride; p.pad = pad; p.groups = groups;
    p.Tout = (stride > 0 && T + 2 * pad >= k) ? (int)((T + 2 * pad - k) / stride + 1) : 0;
    p.cig = groups > 0 ? (int)(c_in / groups) : 0;
    p.cog = groups > 0 ? (int)(c_out / groups) : 0;
    return p;
}

}  // namespace vs

using namespace vs;

extern "C" {

int vs_gconv1d_fwd(const float *x, const float *w, const float *bias, float *y, int64_t B, int64_t c_in, int64_t c_out, int64_t T,
                   int k, int stride, int pad, int groups, void *stream) {
    VS_REQUIRE(x && w && y, "vs_gconv1d_fwd: NULL tensor");
    GConvParams p = make(B, c_in, c_out, T, k, stride, pad, groups);
    VS_TRY(check(p));
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    const long long total = (long long)p.B * p.Cout * p.Tout;
    const dim3 ggrid((unsigned)ceil_div(p.Tout, 256), (unsigned)p.groups, (unsigned)p.B);
    if (p.cog == 16 && p.groups <= 65535 && p.B <= 65535) hipLaunchKernelGGL(gconv_fwd_group_kernel<16>, ggrid, dim3(256), 0, as_stream(stream), p);
    else if (p.cog == 4 && p.groups <= 65535 && p.B <= 65535) hipLaunchKernelGGL(gconv_fwd_group_kernel<4>, ggrid, dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(gconv_fwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), p);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_gconv1d_bwd_data(const float *gy, const float *w, float *gx, int64_t B, int64_t c_in, int64_t c_out, int64_t T, int k,
                        int stride, int pad, int groups, void *stream) {
    VS_REQUIRE(gy && w && gx, "vs_gconv1d_bwd_data: NULL tensor");
    GConvParams p = make(B, c_in, c_out, T, k, stride, pad, groups);
    VS_TRY(check(p));
    p.gy = gy; p.w = w; p.gx = gx;
    const long long total = (long long)p.B * p.Cin * p.T;
    hipLaunchKernelGGL(gconv_bwd_data_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), p);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_gconv1d_bwd_weight(const float *gy, const float *x, float *gw_planes, int64_t B, int64_t c_in, int64_t c_out, int64_t T,
                          int k, int stride, int pad, int groups, void *stream) {
    VS_REQUIRE(gy && x && gw_planes, "vs_gconv1d_bwd_weight: NULL tensor");
    GConvParams p = make(B, c_in, c_out, T, k, stride, pad, groups);
    VS_TRY(check(p));
    VS_REQUIRE(B <= 65535, "vs_gconv1d_bwd_weight: B too large for grid.y");
    p.gy = gy; p.x = x; p.gw = gw_planes;
    hipLaunchKernelGGL(gconv_bwd_weight_kernel, dim3((unsigned)p.Cout, (unsigned)p.B), dim3(256), 0, as_stream(stream), p);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

}  // extern "C"
