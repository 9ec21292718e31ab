// index_ops.hip -- the integer frame bookkeeping of the hot path, bit-exact by construction:
// values are moved, never recomputed.
//   vs_expand_states    models/commons/align_ops.py:22-26   (1-based gather with a zero pad row)
//   vs_make_positions   modules/rel_transformer.py:78-88    (cumsum(x != pad) * (x != pad) + pad, int64)
//   vs_slice_segments   modules/commons/utils.py:86-92      (per-item window of the time axis)
//   vs_mel2token_to_dur utils/audio/align.py:105-129        (frames per token: integer histogram of the alignment)
#include "vs_internal.h"

namespace vs {

// out[b, t, c] (channels_last) or out[b, c, t] (channels_first) = idx ? h[b, idx-1, c] : 0, idx = mel2token[b, t].
// h is [B, Tp, C] (channels_last) or [B, C, Tp] (channels_first).
__global__ void expand_states_kernel(const float *__restrict__ h, const long long *__restrict__ idx, float *__restrict__ out,
                                     int B, int Tp, int T, int C, int h_cf, int out_cf) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * T * C;
    if (e >= total) return;
    int b, t, c;
    if (out_cf) { t = (int)(e % T); c = (int)((e / T) % C); b = (int)(e / ((long long)T * C)); }
    else { c = (int)(e % C); t = (int)((e / C) % T); b = (int)(e / ((long long)T * C)); }
    const long long i = idx[(long long)b * T + t];
    float v = 0.f;
    if (i > 0 && i <= Tp) v = h_cf ? h[((long long)b * C + c) * Tp + (i - 1)] : h[((long long)b * Tp + (i - 1)) * C + c];
    out[e] = v;
}

// one wave per row: ballot + popcount prefix over 64-frame pieces
__global__ void make_positions_kernel(const float *__restrict__ x, long long *__restrict__ pos, int B, int T, float pad,
                                      long long pad_idx) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    int running = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const bool nz = (t < T) && (x[(long long)row * T + t] != pad);
        const unsigned long long m = __ballot(nz);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (t < T) pos[(long long)row * T + t] = (nz ? (long long)(running + before + 1) : 0ll) + pad_idx;
        running += __popcll(m);
    }
}

__global__ void slice_segments_kernel(const float *__restrict__ x, const long long *__restrict__ ids, float *__restrict__ out,
                                      int B, int C, int T, int seg) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * C * seg;
    if (e >= total) return;
    const int s = (int)(e % seg), c = (int)((e / seg) % C), b = (int)(e / ((long long)seg * C));
    const long long t = ids[b] + s;
    out[e] = (t >= 0 && t < T) ? x[((long long)b * C + c) * T + t] : 0.f;
}

// dur[b, i-1] += 1 for every frame whose (1-based) token index is i in [1, T_txt]; index 0 is padding.  Integer
// atomics: order-independent, hence bit-exact.
__global__ void mel2token_hist_kernel(const long long *__restrict__ m2t, unsigned long long *__restrict__ dur, int B, int T,
                                      int T_txt) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)B * T) return;
    const int b = (int)(e / T);
    const long long i = m2t[e];
    if (i > 0 && i <= T_txt) atomicAdd(&dur[(long long)b * T_txt + (i - 1)], 1ull);
}

__global__ void clamp_max_kernel(long long *__restrict__ v, long long n, long long hi) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n && v[e] > hi) v[e] = hi;
}

}  // namespace vs

using namespace vs;

extern "C" {

int vs_expand_states(const float *h, const int64_t *mel2token, float *out, int64_t B, int64_t T_tokens, int64_t T_frames,
                     int64_t C, int h_channels_first, int out_channels_first, void *stream) {
    VS_REQUIRE(h && mel2token && out && B > 0 && T_tokens > 0 && T_frames > 0 && C > 0, "vs_expand_states: bad arguments");
    const long long total = (long long)B * T_frames * C;
    hipLaunchKernelGGL(expand_states_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), h,
                       (const long long *)mel2token, out, (int)B, (int)T_tokens, (int)T_frames, (int)C, h_channels_first,
                       out_channels_first);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_make_positions(const float *x, int64_t *positions, int64_t B, int64_t T, int64_t padding_idx, void *stream) {
    VS_REQUIRE(x && positions && B > 0 && T > 0, "vs_make_positions: bad arguments");
    hipLaunchKernelGGL(make_positions_kernel, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, as_stream(stream), x,
                       (long long *)positions, (int)B, (int)T, (float)padding_idx, (long long)padding_idx);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_slice_segments(const float *x, const int64_t *ids_str, float *out, int64_t B, int64_t C, int64_t T, int64_t segment_size,
                      void *stream) {
    VS_REQUIRE(x && ids_str && out && B > 0 && C > 0 && T > 0 && segment_size > 0, "vs_slice_segments: bad arguments");
    const long long total = (long long)B * C * segment_size;
    hipLaunchKernelGGL(slice_segments_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), x,
                       (const long long *)ids_str, out, (int)B, (int)C, (int)T, (int)segment_size);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_mel2token_to_dur(const int64_t *mel2token, int64_t *dur, int64_t B, int64_t T_frames, int64_t T_tokens, int64_t max_dur,
                        void *stream) {
    VS_REQUIRE(mel2token && dur && B > 0 && T_frames > 0 && T_tokens > 0, "vs_mel2token_to_dur: bad arguments");
    VS_CHECK_HIP(hipMemsetAsync(dur, 0, sizeof(int64_t) * (size_t)(B * T_tokens), as_stream(stream)));
    hipLaunchKernelGGL(mel2token_hist_kernel, dim3((unsigned)ceil_div(B * T_frames, 256)), dim3(256), 0, as_stream(stream),
                       (const long long *)mel2token, (unsigned long long *)dur, (int)B, (int)T_frames, (int)T_tokens);
    VS_CHECK_HIP(hipGetLastError());
    if (max_dur >= 0) {
        hipLaunchKernelGGL(clamp_max_kernel, dim3((unsigned)ceil_div(B * T_tokens, 256)), dim3(256), 0, as_stream(stream),
                           (long long *)dur, (long long)(B * T_tokens), (long long)max_dur);
        VS_CHECK_HIP(hipGetLastError());
    }
    return VS_OK;
}

}  // extern "C"
