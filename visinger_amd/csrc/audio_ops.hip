// audio_ops.hip -- the elementwise stage of the on-device spectrograms (utils/audio/mel_processing.py:15-38: torchaudio
// Spectrogram / MelSpectrogram with power = 2).  The transforms themselves are matrix work and run on the conv engine:
//   framed, windowed DFT  = one strided conv of the padded waveform with the (cos | -sin) * hann basis  (visinger_amd/audio.py),
//   mel projection        = a 1x1 conv with the HTK filterbank;
// between them sits  P[b, f, t] = re^2 + im^2  on the [B, 2F, T] transform output (rows f: real parts, rows F + f: imaginary parts)
// and, for the mel loss of the training step (tasks/base.py:232-238 on the generated segment), its backward.  HBM-bound: 12 B / bin.
#include "vs_internal.h"

namespace vs {

__global__ void __launch_bounds__(256) spec_power_fwd_kernel(const float *__restrict__ y, float *__restrict__ p, int F, int T) {
    const int b = blockIdx.z, f = blockIdx.y;
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t >= T) return;
    const float *re = y + ((long long)b * 2 * F + f) * T + t, *im = re + (long long)F * T;
    float *po = p + ((long long)b * F + f) * T + t;
    if (t + 4 <= T && (T & 3) == 0) {
        const float4 a = *reinterpret_cast<const float4 *>(re), c = *reinterpret_cast<const float4 *>(im);
        float4 o;
        o.x = a.x * a.x + c.x * c.x; o.y = a.y * a.y + c.y * c.y; o.z = a.z * a.z + c.z * c.z; o.w = a.w * a.w + c.w * c.w;
        *reinterpret_cast<float4 *>(po) = o;
    } else {
        for (int i = 0; i < 4 && t + i < T; ++i) po[i] = re[i] * re[i] + im[i] * im[i];
    }
}

// dy[b, f, t] = 2 re dp,  dy[b, F + f, t] = 2 im dp
__global__ void __launch_bounds__(256) spec_power_bwd_kernel(const float *__restrict__ y, const float *__restrict__ dp, float *__restrict__ dy,
                                                             int F, int T) {
    const int b = blockIdx.z, f = blockIdx.y;
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t >= T) return;
    const long long ro = ((long long)b * 2 * F + f) * T + t, io = ro + (long long)F * T;
    const float *g = dp + ((long long)b * F + f) * T + t;
    for (int i = 0; i < 4 && t + i < T; ++i) {
        const float d = 2.f * g[i];
        dy[ro + i] = y[ro + i] * d;
        dy[io + i] = y[io + i] * d;
    }
}

}  // namespace vs

using namespace vs;

extern "C" {

int vs_spec_power_fwd(const float *y, float *p, int64_t B, int64_t F, int64_t T, void *stream) {
    VS_REQUIRE(y && p && B > 0 && B <= 65535 && F > 0 && F <= 65535 && T > 0, "vs_spec_power_fwd: bad arguments");
    dim3 grid((unsigned)ceil_div(T, 1024), (unsigned)F, (unsigned)B);
    hipLaunchKernelGGL(spec_power_fwd_kernel, grid, dim3(256), 0, as_stream(stream), y, p, (int)F, (int)T);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_spec_power_bwd(const float *y, const float *dp, float *dy, int64_t B, int64_t F, int64_t T, void *stream) {
    VS_REQUIRE(y && dp && dy && B > 0 && B <= 65535 && F > 0 && F <= 65535 && T > 0, "vs_spec_power_bwd: bad arguments");
    dim3 grid((unsigned)ceil_div(T, 1024), (unsigned)F, (unsigned)B);
    hipLaunchKernelGGL(spec_power_bwd_kernel, grid, dim3(256), 0, as_stream(stream), y, dp, dy, (int)F, (int)T);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

}  // extern "C"
