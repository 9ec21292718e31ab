// train_ops.hip -- fused elementwise kernels of the TRAINING path: the neighbours of the convs that the inference
// path folds into conv epilogues, here as one forward and one backward launch each instead of 10-25 PyTorch elementwise kernels
// (the GAN step at B=16, T_mel=512 is bound by launch count and small-kernel time):
//   vs_gate_fwd / vs_gate_bwd            WaveNet gate  acts = tanh(a + g_a) * sigmoid(b + g_b)         (encoder.py:206-213)
//   vs_layernorm_c_bwd                   channel LayerNorm backward (forward: vs_layernorm_c_fwd)      (rel_transformer.py:33-42)
//   vs_phase_stack / vs_phase_items / vs_phase_unstack   layout kernels of a strided conv as a stride-1 conv over its input phases
//                                        (discriminators, modules/discriminator.py:20-24: one launch each where pad / view / permute /
//                                        contiguous / zeros took 3-5)
// All HBM-bound: every tensor is read once and written once; reductions over channels in registers + one LDS exchange, reductions
// over (batch, time) as one float atomic per channel and 64-frame block (LayerNorm: into one row per batch item, summed by the caller).
#include "vs_internal.h"

#include <algorithm>

namespace vs {

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

// x_in [B, 2H, T] (rows c: tanh half, rows H + c: sigmoid half); g: optional per-item bias [B, >= 2H] with row stride g_bs
__global__ void __launch_bounds__(256) gate_fwd_kernel(const float *__restrict__ x_in, const float *__restrict__ g, long long g_bs,
                                                       float *__restrict__ acts, int H, int T) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t >= T) return;
    const float ga = g ? g[(long long)b * g_bs + c] : 0.f, gb = g ? g[(long long)b * g_bs + H + c] : 0.f;
    const float *pa = x_in + ((long long)b * 2 * H + c) * T + t, *pb = pa + (long long)H * T;
    float *po = acts + ((long long)b * H + c) * T + t;
    if (t + 4 <= T && (T & 3) == 0) {
        const float4 a = *reinterpret_cast<const float4 *>(pa), s = *reinterpret_cast<const float4 *>(pb);
        float4 o;
        o.x = tanhf(a.x + ga) * sigm(s.x + gb); o.y = tanhf(a.y + ga) * sigm(s.y + gb);
        o.z = tanhf(a.z + ga) * sigm(s.z + gb); o.w = tanhf(a.w + ga) * sigm(s.w + gb);
        *reinterpret_cast<float4 *>(po) = o;
    } else {
        for (int i = 0; i < 4 && t + i < T; ++i) po[i] = tanhf(pa[i] + ga) * sigm(pb[i] + gb);
    }
}

// dx_in[:, c] = dacts * s * (1 - th^2), dx_in[:, H + c] = dacts * th * s * (1 - s); dg[b, row] += sum_t dx_in (atomic per block)
__global__ void __launch_bounds__(256) gate_bwd_kernel(const float *__restrict__ x_in, const float *__restrict__ g, long long g_bs,
                                                       const float *__restrict__ dacts, float *__restrict__ dx_in, float *__restrict__ dg,
                                                       long long dg_bs, int H, int T) {
    __shared__ float red[2][4];
    const int b = blockIdx.z, c = blockIdx.y;
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    const float ga = g ? g[(long long)b * g_bs + c] : 0.f, gb = g ? g[(long long)b * g_bs + H + c] : 0.f;
    float sa = 0.f, sb = 0.f;
    if (t < T) {
        const float *pa = x_in + ((long long)b * 2 * H + c) * T + t, *pb = pa + (long long)H * T;
        const float *pd = dacts + ((long long)b * H + c) * T + t;
        float *qa = dx_in + ((long long)b * 2 * H + c) * T + t, *qb = qa + (long long)H * T;
        for (int i = 0; i < 4 && t + i < T; ++i) {
            const float th = tanhf(pa[i] + ga), s = sigm(pb[i] + gb), d = pd[i];
            const float da = d * s * (1.f - th * th), db = d * th * s * (1.f - s);
            qa[i] = da;
            qb[i] = db;
            sa += da;
            sb += db;
        }
    }
    if (dg) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[0][w] = sa; red[1][w] = sb; }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(dg + (long long)b * dg_bs + c, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
            atomicAdd(dg + (long long)b * dg_bs + H + c, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        }
    }
}

// LayerNorm over channels, backward.  x = a (+ r) [B, C, T]; y = xhat * gamma + beta, xhat = (x - mean_c) * rstd.
//   dxhat = dy * gamma;  dx = rstd * (dxhat - mean_c(dxhat) - xhat * mean_c(dxhat * xhat));  dgamma[c] = sum_{b,t} dy * xhat;  dbeta[c] = sum dy
// block = G channel groups x 64 frames (as layernorm_c_kernel); each thread keeps its C/G channel values in registers.
template <int G, int PT>
__global__ void __launch_bounds__(64 * G) layernorm_c_bwd_kernel(const float *__restrict__ a, const float *__restrict__ r,
                                                                const float *__restrict__ gamma, const float *__restrict__ dy,
                                                                float *__restrict__ dx, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                int C, int T, float eps) {
    __shared__ float red[G][64];
    const int tl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int t = blockIdx.x * 64 + tl;
    const bool okt = t < T;
    const int tc = min(t, T - 1);
    const long long base = (long long)b * C * T + tc;
    float x[PT], d[PT];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int c = grp + G * i, cc = min(c, C - 1);
        float v = a[base + (long long)cc * T];
        if (r) v += r[base + (long long)cc * T];
        x[i] = (c < C) ? v : 0.f;
        d[i] = (c < C && okt) ? dy[base + (long long)cc * T] : 0.f;
        sum += x[i];
    }
    auto block_sum = [&](float v) {
        red[grp][tl] = v;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int gI = 0; gI < G; ++gI) tot += red[gI][tl];
        __syncthreads();
        return tot;
    };
    const float mean = block_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int c = grp + G * i;
        const float dd = (c < C) ? x[i] - mean : 0.f;
        sq += dd * dd;
    }
    const float rstd = rsqrtf(block_sum(sq) / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int c = grp + G * i;
        const float xh = (c < C) ? (x[i] - mean) * rstd : 0.f;
        const float dxh = (c < C) ? d[i] * gamma[min(c, C - 1)] : 0.f;
        x[i] = xh;                       // keep xhat
        s1 += dxh;
        s2 += dxh * xh;
    }
    const float m1 = block_sum(s1) / (float)C, m2 = block_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int c = grp + G * i;
        if (c < C) {
            const float dxh = d[i] * gamma[c];
            if (okt) dx[base + (long long)c * T] = rstd * (dxh - m1 - x[i] * m2);
            // dgamma / dbeta: reduce over the 64 frames of this block, one atomic per channel
            float pg = d[i] * x[i], pb = d[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { pg += __shfl_xor(pg, o); pb += __shfl_xor(pb, o); }
            // per batch ITEM: all (T / 64) x B workgroups adding into one [C] row serialised on the same addresses (119 us per call at
            // B = 16, T = 512: 128-way contention per channel); the caller sums the B rows
            if (tl == 0) { atomicAdd(dgamma + (long long)b * 2 * C + c, pg); atomicAdd(dbeta + (long long)b * 2 * C + c, pb); }
        }
    }
}


// ---- a stride-s conv as the stride-1 conv of its s input phases (autograd.StridedConv1dFn): N items, padded by `pad` zeros on the left and
// as many as needed on the right, de-interleaved into s * C channels of Hq positions and laid END TO END as one sequence of N * Hq columns
//   XF[r * C + c][n * Hq + j] = xpad[n][c][j * s + r] = x[n][c][j * s + r - pad]   (0 outside [0, T))
// x is read through its strides (a permuted / sliced view needs no contiguous copy first).
__global__ void __launch_bounds__(256) phase_stack_kernel(const float *__restrict__ x, long long sn, long long sc, long long st,
                                                          float *__restrict__ xf, int N, int C, int T, int s, int pad, int Hq, long long cols) {
    const long long col = (long long)blockIdx.x * 256 + threadIdx.x;     // cols >= N * Hq: the columns past the last item are zeros
    if (col >= cols) return;
    const int row = blockIdx.y;                 // r * C + c
    const int r = row / C, c = row - r * C;
    const int n = (int)(col / Hq), j = (int)(col - (long long)n * Hq);
    const int t = j * s + r - pad;
    xf[(long long)row * cols + col] = (n < N && t >= 0 && t < T) ? x[n * sn + c * sc + t * st] : 0.f;
}

// items [N, C, Tv] -> one sequence [C][N * Hq], the Hq - Tv trailing positions of every item zero (output gradients on their way into the
// grad-input conv / the weight-gradient kernel), and back: sequence [C][ld] -> items [N, C, Tv]
__global__ void __launch_bounds__(256) phase_items_kernel(const float *__restrict__ src, float *__restrict__ dst, int N, int C, int Tv, int Hq,
                                                          long long ld, int to_sequence) {
    const long long cols = (long long)N * Hq;
    const long long col = (long long)blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    const int c = blockIdx.y;
    const int n = (int)(col / Hq), j = (int)(col - (long long)n * Hq);
    if (to_sequence) dst[(long long)c * ld + col] = (j < Tv) ? src[((long long)n * C + c) * Tv + j] : 0.f;
    else if (j < Tv) dst[((long long)n * C + c) * Tv + j] = src[(long long)c * ld + col];
}

// the adjoint of phase_stack: gx[n][c][t] = gXF[r * C + c][n * Hq + j] with (j, r) = divmod(t + pad, s)   (row stride ld of gXF)
__global__ void __launch_bounds__(256) phase_unstack_kernel(const float *__restrict__ gxf, long long ld, float *__restrict__ gx, int N, int C,
                                                            int T, int s, int pad, int Hq) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int c = blockIdx.y, n = blockIdx.z;
    const int j = (t + pad) / s, r = (t + pad) - j * s;
    gx[((long long)n * C + c) * T + t] = (j < Hq) ? gxf[(long long)(r * C + c) * ld + (long long)n * Hq + j] : 0.f;
}

}  // namespace vs

// ---- weight norm (torch.nn.utils.weight_norm, dim 0) of EVERY weight-normed conv of a network in one launch each way (the reference folds g * v / ||v||
// per module through the parametrisation hook; a training step did it as 256 + 182 PyTorch launches).  The table (device memory, n rows of 8 x int64:
// v, g, w, norm, rows, cols, first row, 0) lists the tensors; one wave per row of a tensor, the row's tensor found by bisection over `first row`.
struct WnEntry {
    const float *v, *g;
    float *w, *norm;
    long long rows, cols, row0, pad_;
};
struct WnGrad {            // per backward call: the incoming dL/dw (NULL: this tensor takes no gradient) and the two outputs
    const float *gw;
    float *gv, *gg;
    long long pad_;
};
__device__ __forceinline__ int wn_find(const WnEntry *__restrict__ tab, int n, long long row) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].row0 <= row) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    return v;
}
__global__ void __launch_bounds__(256) weight_norm_multi_fwd_kernel(const WnEntry *__restrict__ tab, int n, long long total_rows) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= total_rows) return;
    const WnEntry e = tab[wn_find(tab, n, row)];
    const long long r = row - e.row0;
    const float *vr = e.v + r * e.cols;
    float ss = 0.f;
    for (long long c = lane; c < e.cols; c += 64) ss += vr[c] * vr[c];
    const float nrm = sqrtf(wave_sum(ss));
    const float sc = e.g[r] / nrm;
    float *wr = e.w + r * e.cols;
    for (long long c = lane; c < e.cols; c += 64) wr[c] = vr[c] * sc;
    if (lane == 0) e.norm[r] = nrm;
}
// dL/dg[r] = <gw_r, v_r> / ||v_r||;  dL/dv_r = (g[r] / ||v_r||) * (gw_r - v_r <gw_r, v_r> / ||v_r||^2)
__global__ void __launch_bounds__(256) weight_norm_multi_bwd_kernel(const WnEntry *__restrict__ tab, const WnGrad *__restrict__ gr, int n,
                                                                    long long total_rows) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= total_rows) return;
    const int i = wn_find(tab, n, row);
    const WnGrad d = gr[i];
    if (!d.gw) return;
    const WnEntry e = tab[i];
    const long long r = row - e.row0;
    const float *vr = e.v + r * e.cols, *gwr = d.gw + r * e.cols;
    float dot = 0.f;
    for (long long c = lane; c < e.cols; c += 64) dot += gwr[c] * vr[c];
    dot = wave_sum(dot);
    const float rn = 1.f / e.norm[r];
    const float a = e.g[r] * rn, bcoef = e.g[r] * rn * rn * rn * dot;
    float *gvr = d.gv + r * e.cols;
    for (long long c = lane; c < e.cols; c += 64) gvr[c] = a * gwr[c] - bcoef * vr[c];
    if (lane == 0) d.gg[r] = dot * rn;
}

// ---- the residual / skip update of a WaveNet layer (encoder.py:186-193): rs = res_skip_layer(acts) [B, 2H, T];
//        x_new = (x + rs[:, :H]) * mask,   out_new = out_acc + rs[:, H:]       (out_acc NULL: the first layer, out_new = rs[:, H:])
// one launch where autograd recorded two slices, two adds and a multiply (and ran ~9 kernels backward); backward:
//        d_rs[:, :H] = dx = dx_new * mask,   d_rs[:, H:] = dout_new            (d out_acc = dout_new itself)
__global__ void __launch_bounds__(256) wn_step_fwd_kernel(const float *__restrict__ x, const float *__restrict__ rs, const float *__restrict__ out_acc,
                                                          const float *__restrict__ mask, float *__restrict__ x_new, float *__restrict__ out_new, int H, int T) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t >= T) return;
    const long long o = ((long long)b * H + c) * T + t, r0 = ((long long)b * 2 * H + c) * T + t, r1 = r0 + (long long)H * T;
    const float *m = mask + (long long)b * T + t;
    if (t + 4 <= T && (T & 3) == 0) {
        const float4 xv = *reinterpret_cast<const float4 *>(x + o), ra = *reinterpret_cast<const float4 *>(rs + r0), rb = *reinterpret_cast<const float4 *>(rs + r1);
        const float4 mv = *reinterpret_cast<const float4 *>(m);
        float4 oa = out_acc ? *reinterpret_cast<const float4 *>(out_acc + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 xn, on;
        xn.x = (xv.x + ra.x) * mv.x; xn.y = (xv.y + ra.y) * mv.y; xn.z = (xv.z + ra.z) * mv.z; xn.w = (xv.w + ra.w) * mv.w;
        on.x = oa.x + rb.x; on.y = oa.y + rb.y; on.z = oa.z + rb.z; on.w = oa.w + rb.w;
        *reinterpret_cast<float4 *>(x_new + o) = xn;
        *reinterpret_cast<float4 *>(out_new + o) = on;
    } else {
        for (int i = 0; i < 4 && t + i < T; ++i) {
            x_new[o + i] = (x[o + i] + rs[r0 + i]) * m[i];
            out_new[o + i] = (out_acc ? out_acc[o + i] : 0.f) + rs[r1 + i];
        }
    }
}
// dx_new / dout_new may be NULL (that output took no gradient): zeros
__global__ void __launch_bounds__(256) wn_step_bwd_kernel(const float *__restrict__ dx_new, const float *__restrict__ dout_new, const float *__restrict__ mask,
                                                          float *__restrict__ d_rs, float *__restrict__ dx, int H, int T) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t >= T) return;
    const long long o = ((long long)b * H + c) * T + t, r0 = ((long long)b * 2 * H + c) * T + t, r1 = r0 + (long long)H * T;
    const float *m = mask + (long long)b * T + t;
    for (int i = 0; i < 4 && t + i < T; ++i) {
        const float v = dx_new ? dx_new[o + i] * m[i] : 0.f;
        dx[o + i] = v;
        d_rs[r0 + i] = v;
        d_rs[r1 + i] = dout_new ? dout_new[o + i] : 0.f;
    }
}

// ---- mean |a - b| over n elements (the feature-matching loss, tasks/visinger.py:162-169: one such term per discriminator layer, 54 a step; autograd ran
// sub / abs / mean forward and div / sgn / mul / neg backward for each).  One launch each way.  Forward: <= 256 workgroups leave fixed-order partial sums in
// work[0 .. 255]; the last one to finish (a ticket in work[256], reset for the next call) adds them up in index order: deterministic, no second launch.
__global__ void __launch_bounds__(256) l1_mean_fwd_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ work, float *__restrict__ out,
                                                          long long n, float inv_n) {
    __shared__ float red[256];
    __shared__ unsigned ticket;
    const int tid = threadIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const long long stride = (long long)gridDim.x * 1024;
    for (long long i = (long long)blockIdx.x * 1024 + tid; i < n; i += stride) {
        s0 += fabsf(a[i] - b[i]);
        if (i + 256 < n) s1 += fabsf(a[i + 256] - b[i + 256]);
        if (i + 512 < n) s2 += fabsf(a[i + 512] - b[i + 512]);
        if (i + 768 < n) s3 += fabsf(a[i + 768] - b[i + 768]);
    }
    red[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    unsigned *counter = reinterpret_cast<unsigned *>(work + 256);
    if (tid == 0) {
        work[blockIdx.x] = red[0];
        __threadfence();
        ticket = atomicAdd(counter, 1u);
    }
    __syncthreads();
    if (ticket == gridDim.x - 1) {          // every partial is in memory
        __threadfence();
        red[tid] = (tid < (int)gridDim.x) ? __builtin_nontemporal_load(work + tid) : 0.f;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) { out[0] = red[0] * inv_n; *counter = 0u; }
    }
}
// d loss / d a = sign(a - b) * gout / n (sign(0) = 0, as torch.abs' backward); the gradient of b is its negative
__global__ void __launch_bounds__(256) l1_mean_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ gout,
                                                          float *__restrict__ da, long long n, float inv_n) {
    const float g = gout[0] * inv_n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = a[i] - b[i];
        da[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    }
}

using namespace vs;

// bias gradient of a conv: gb[c] = sum over (b, t) of gy[b, c, t] -- one workgroup per channel, every thread a fixed strided subset in
// a fixed order, then a fixed LDS tree: the same bits every run (no atomics), one launch where `gy.sum((0, 2))` took two
__global__ void __launch_bounds__(256) bias_grad_kernel(const float *__restrict__ gy, float *__restrict__ gb, int B, int C, int T) {
    __shared__ float red[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    if ((T & 3) == 0 && (reinterpret_cast<uintptr_t>(gy) & 15u) == 0) {
        // the (b, t) plane of the channel as ONE index space, eight float4 loads of a thread in flight at a time (a loop over b with a loop over t inside
        // left one load per thread in flight and half the threads idle at T = 512: 16 serial round trips, 16 us per launch, 271 launches per
        // training step); fixed assignment, fixed order: the same bits every run
        const int T4 = T >> 2, n4 = B * T4;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i0 = tid; i0 < n4; i0 += 8 * 256) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = min(i0 + q * 256, n4 - 1);
                const int b = i / T4, t4 = i - b * T4;
                v[q] = reinterpret_cast<const float4 *>(gy + ((long long)b * C + c) * T)[t4];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i0 + q * 256 < n4) acc[q] += (v[q].x + v[q].y) + (v[q].z + v[q].w);
        }
        s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    } else {
        for (int b = 0; b < B; ++b) {
            const float *row = gy + ((long long)b * C + c) * T;
            for (int i = tid; i < T; i += 256) s += row[i];
        }
    }
    red[tid] = s;
    __syncthreads();
#pragma unroll
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) gb[c] = red[0];
}

extern "C" {

int vs_bias_grad(const float *gy, float *gb, int64_t B, int64_t C, int64_t T, void *stream) {
    VS_REQUIRE(gy && gb && B > 0 && C > 0 && T > 0 && C <= 65535 * 32, "vs_bias_grad: bad arguments");
    hipLaunchKernelGGL(bias_grad_kernel, dim3((unsigned)C), dim3(256), 0, as_stream(stream), gy, gb, (int)B, (int)C, (int)T);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_weight_norm_multi_fwd(const void *table, int64_t n, int64_t total_rows, void *stream) {
    VS_REQUIRE(table && n > 0 && total_rows > 0 && total_rows <= 4ll * 0x7fffffff, "vs_weight_norm_multi_fwd: bad arguments");
    hipLaunchKernelGGL(weight_norm_multi_fwd_kernel, dim3((unsigned)ceil_div(total_rows, 4)), dim3(256), 0, as_stream(stream),
                       static_cast<const WnEntry *>(table), (int)n, (long long)total_rows);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_weight_norm_multi_bwd(const void *table, const void *grads, int64_t n, int64_t total_rows, void *stream) {
    VS_REQUIRE(table && grads && n > 0 && total_rows > 0 && total_rows <= 4ll * 0x7fffffff, "vs_weight_norm_multi_bwd: bad arguments");
    hipLaunchKernelGGL(weight_norm_multi_bwd_kernel, dim3((unsigned)ceil_div(total_rows, 4)), dim3(256), 0, as_stream(stream),
                       static_cast<const WnEntry *>(table), static_cast<const WnGrad *>(grads), (int)n, (long long)total_rows);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_wn_step_fwd(const float *x, const float *rs, const float *out_acc, const float *mask, float *x_new, float *out_new, int64_t B, int64_t H,
                   int64_t T, void *stream) {
    VS_REQUIRE(x && rs && mask && x_new && out_new && B > 0 && B <= 65535 && H > 0 && H <= 65535 && T > 0, "vs_wn_step_fwd: bad arguments");
    dim3 grid((unsigned)ceil_div(T, 1024), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL(wn_step_fwd_kernel, grid, dim3(256), 0, as_stream(stream), x, rs, out_acc, mask, x_new, out_new, (int)H, (int)T);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_wn_step_bwd(const float *dx_new, const float *dout_new, const float *mask, float *d_rs, float *dx, int64_t B, int64_t H, int64_t T, void *stream) {
    VS_REQUIRE(mask && d_rs && dx && B > 0 && B <= 65535 && H > 0 && H <= 65535 && T > 0, "vs_wn_step_bwd: bad arguments");
    dim3 grid((unsigned)ceil_div(T, 1024), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL(wn_step_bwd_kernel, grid, dim3(256), 0, as_stream(stream), dx_new, dout_new, mask, d_rs, dx, (int)H, (int)T);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_l1_mean_fwd(const float *a, const float *b, float *work, float *out, int64_t n, void *stream) {
    VS_REQUIRE(a && b && work && out && n > 0, "vs_l1_mean_fwd: bad arguments");
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, 1024), 256);
    hipLaunchKernelGGL(l1_mean_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), a, b, work, out, (long long)n, 1.0f / (float)n);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_l1_mean_bwd(const float *a, const float *b, const float *gout, float *da, int64_t n, void *stream) {
    VS_REQUIRE(a && b && gout && da && n > 0, "vs_l1_mean_bwd: bad arguments");
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, 256), 2048);
    hipLaunchKernelGGL(l1_mean_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), a, b, gout, da, (long long)n, 1.0f / (float)n);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_gate_fwd(const float *x_in, const float *g, int64_t g_bs, float *acts, int64_t B, int64_t H, int64_t T, void *stream) {
    VS_REQUIRE(x_in && acts && B > 0 && B <= 65535 && H > 0 && H <= 65535 && T > 0, "vs_gate_fwd: bad arguments");
    dim3 grid((unsigned)ceil_div(T, 1024), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL(gate_fwd_kernel, grid, dim3(256), 0, as_stream(stream), x_in, g, (long long)g_bs, acts, (int)H, (int)T);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_gate_bwd(const float *x_in, const float *g, int64_t g_bs, const float *dacts, float *dx_in, float *dg, int64_t dg_bs, int64_t B,
                int64_t H, int64_t T, void *stream) {
    VS_REQUIRE(x_in && dacts && dx_in && B > 0 && B <= 65535 && H > 0 && H <= 65535 && T > 0, "vs_gate_bwd: bad arguments");
    dim3 grid((unsigned)ceil_div(T, 1024), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL(gate_bwd_kernel, grid, dim3(256), 0, as_stream(stream), x_in, g, (long long)g_bs, dacts, dx_in, dg, (long long)dg_bs,
                       (int)H, (int)T);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_layernorm_c_bwd(const float *a, const float *r, const float *gamma, const float *dy, float *dx, float *dgamma, float *dbeta,
                       int64_t B, int64_t C, int64_t T, float eps, void *stream) {
    VS_REQUIRE(a && gamma && dy && dx && dgamma && dbeta && B > 0 && B <= 65535 && C > 0 && T > 0, "vs_layernorm_c_bwd: bad arguments");
    hipStream_t s = as_stream(stream);
    dim3 grid((unsigned)ceil_div(T, 64), (unsigned)B);
    const int Ci = (int)C, Ti = (int)T;
    // (16 channel groups of <= 16 channels for the usual widths: 1 024 threads per 64-frame block -- with 4 groups of 48 channels a
    //  B = 16, T = 512 launch was 128 blocks of 256 threads walking 48 rows each: 67 us for 19 MB)
    if (C <= 4 * 16) hipLaunchKernelGGL((layernorm_c_bwd_kernel<4, 16>), grid, dim3(256), 0, s, a, r, gamma, dy, dx, dgamma, dbeta, Ci, Ti, eps);
    else if (C <= 16 * 16) hipLaunchKernelGGL((layernorm_c_bwd_kernel<16, 16>), grid, dim3(1024), 0, s, a, r, gamma, dy, dx, dgamma, dbeta, Ci, Ti, eps);
    else if (C <= 16 * 32) hipLaunchKernelGGL((layernorm_c_bwd_kernel<16, 32>), grid, dim3(1024), 0, s, a, r, gamma, dy, dx, dgamma, dbeta, Ci, Ti, eps);
    else if (C <= 16 * 64) hipLaunchKernelGGL((layernorm_c_bwd_kernel<16, 64>), grid, dim3(1024), 0, s, a, r, gamma, dy, dx, dgamma, dbeta, Ci, Ti, eps);
    else { set_error("vs_layernorm_c_bwd: C=%lld > 1024 unsupported", (long long)C); return VS_EUNSUPPORTED; }
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_phase_stack(const float *x, int64_t sn, int64_t sc, int64_t st, float *xf, int64_t N, int64_t C, int64_t T, int stride, int pad,
                   int64_t Hq, int64_t cols, void *stream) {
    VS_REQUIRE(x && xf && N > 0 && C > 0 && T > 0 && stride > 0 && pad >= 0 && Hq > 0 && (int64_t)stride * C <= 65535 && cols >= N * Hq &&
               cols < (1ll << 31), "vs_phase_stack: bad arguments");
    dim3 grid((unsigned)ceil_div(cols, 256), (unsigned)(stride * C));
    hipLaunchKernelGGL(phase_stack_kernel, grid, dim3(256), 0, as_stream(stream), x, (long long)sn, (long long)sc, (long long)st, xf, (int)N,
                       (int)C, (int)T, stride, pad, (int)Hq, (long long)cols);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_phase_items(const float *src, float *dst, int64_t N, int64_t C, int64_t Tv, int64_t Hq, int64_t ld, int to_sequence, void *stream) {
    VS_REQUIRE(src && dst && N > 0 && C > 0 && C <= 65535 && Tv > 0 && Hq >= Tv && ld >= N * Hq - (to_sequence ? 0 : Hq - Tv) &&
               N * Hq < (1ll << 31), "vs_phase_items: bad arguments");
    dim3 grid((unsigned)ceil_div(N * Hq, 256), (unsigned)C);
    hipLaunchKernelGGL(phase_items_kernel, grid, dim3(256), 0, as_stream(stream), src, dst, (int)N, (int)C, (int)Tv, (int)Hq, (long long)ld,
                       to_sequence);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int vs_phase_unstack(const float *gxf, int64_t ld, float *gx, int64_t N, int64_t C, int64_t T, int stride, int pad, int64_t Hq, void *stream) {
    VS_REQUIRE(gxf && gx && N > 0 && N <= 65535 && C > 0 && C <= 65535 && T > 0 && stride > 0 && pad >= 0 && Hq > 0 && ld >= N * Hq,
               "vs_phase_unstack: bad arguments");
    dim3 grid((unsigned)ceil_div(T, 256), (unsigned)C, (unsigned)N);
    hipLaunchKernelGGL(phase_unstack_kernel, grid, dim3(256), 0, as_stream(stream), gxf, (long long)ld, gx, (int)N, (int)C, (int)T, stride, pad,
                       (int)Hq);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

}  // extern "C"
