// conv_split.hip -- the implicit-GEMM conv engine on the bf16 matrix instruction v_mfma_f32_32x32x16_bf16 (gfx950: 16x the
// FLOP/clk of the fp32 MFMA), with fp32 operands SPLIT into bf16 planes so that the result stays in the fp32 class.
//
//   x = xh + xm + xl   EXACTLY, by successive truncation: xh = top 8 significant bits of x, xm = top 8 of (x - xh), xl = the
//   rest (a 24-bit significand is three 8-bit pieces; every piece is a bf16 number and the subtractions are exact).  The
//   same for the weights, once, at pack time.  Then
//       x*w = xh*wh + xh*wm + xm*wh + xh*wl + xl*wh + xm*wm          (TERMS = 6; each product exact in fp32 inside the MFMA)
//             + [xm*wl + xl*wm + xl*wl]                               (dropped: <= 2^-23 |x*w|, the size of ONE fp32 rounding)
//   so a conv costs 6 bf16 MFMAs per 16 input channels and 32x32 outputs where the fp32 MFMA needs 8 of twice the duration:
//   16/6 = 2.67x the fp32 matrix peak with fp32 accumulation and fp32-class error (measured against the fp64 oracle in
//   tests/test_conv_split_gpu.py: below the error of the fp32 Winograd F(2,3) instances it replaces).
//   TERMS = 3 keeps (hh, hm, mh): ~2^-16 relative; TERMS = 1 is plain bf16 (round-to-nearest-even operands): the
//   "bf16 activations/weights, fp32 accumulate" arithmetic of BASELINE.json's long-form configuration.
//
// Same structure as conv_mfma_kernel (conv_engine.hip): weights pre-packed in fragment order and fetched by coalesced 16-B
// loads through L2/L1, activations staged once per 16-channel chunk into LDS (input transform fused, here also the split),
// every tap reads a shifted window of the same LDS tile, the same fused epilogue (conv_epilogue.inc).  What differs:
//   * LDS tile layout [plane][k-group(2)][column][8 channels as bf16] -- a B fragment (8 consecutive k of one column) is ONE
//     conflict-free ds_read_b128, and one read feeds up to three MFMAs (the h plane meets wh, wm, wl);
//   * a wave stages 4 CONSECUTIVE channels (two packed dwords per plane and column, one ds_write_b64);
//   * one (chunk, tap) step is NT_W groups of TERMS MFMAs; the planes of column tile j+1 are read under the MFMAs of tile j.
#include "conv_common.h"

#include <algorithm>
#include <type_traits>

namespace vs {

// Timing-only perturbations of the main loop (results are garbage): build with -DVS_SPLIT_PERTURB and set VS_SPLIT_DBG to a sum of
// 1 = no weight-fragment loads after the first steps, 2 = no activation loads after the prologue, 4 = no split + LDS write of the next
// chunk, 8 = no B-fragment reads.  Compiled out by default: a run-time test per column tile in the hot loop cost 1.6 % of the headline.
#ifdef VS_SPLIT_PERTURB
#define PERTURB(bit) ((p.dbg & (bit)) != 0)
#else
#define PERTURB(bit) false
#endif

int split_planes(int terms) { return terms == 1 ? 1 : (terms == 3 ? 2 : 3); }

template <int MT_W, int NT_W, int WAVES_M, int WAVES_N, int TERMS>
__global__ void __launch_bounds__(256, 2) conv_split_kernel(const ConvParams p) {
    constexpr bool XB = false;
#define VS_EPILOGUE_INC "conv_epilogue.inc"
#ifdef VS_EPI_OLD_BIAS_ROW             // (debug builds: the round-3 epilogue that loads a row's bias inside the row loop -- the A/B of conv_split_body.inc's note)
#define VS_EPI_BIAS_IN_LOOP 1
#else
#define VS_EPI_BIAS_IN_LOOP 0
#endif
#include "conv_split_body.inc"
#undef VS_EPI_BIAS_IN_LOOP
#undef VS_EPILOGUE_INC
}

// Transposed convs: the same body with the polyphase stores of conv_epilogue_tr.inc in front of the generic epilogue.  A kernel template
// of its own so that the instances every other conv runs stay byte-identical.
template <int MT_W, int NT_W, int WAVES_M, int WAVES_N, int TERMS>
__global__ void __launch_bounds__(256, 2) conv_split_tr_kernel(const ConvParams p) {
    constexpr bool XB = false;
#define VS_EPILOGUE_INC "conv_epilogue_tr.inc"
#define VS_EPI_BIAS_IN_LOOP 1      // (the vector epilogue behind the polyphase stores is the rare path here: no bias register held for it)
#define VS_EPI_NO_FAST 1           // (... and the row-contiguous fast path never applies to polyphase rows: compiled out)
#include "conv_split_body.inc"
#undef VS_EPI_NO_FAST
#undef VS_EPI_BIAS_IN_LOOP
#undef VS_EPILOGUE_INC
}

// bf16-RESIDENT tensors (plain-bf16 arithmetic only, BASELINE config 5): IO bit 0 -- x holds bf16 elements (widened on the way into LDS:
// every product of this arithmetic rounds its operands to bf16 anyway), bit 1 -- y / res / acc do (rounded to nearest even once, after
// residual / accumulate / scale / activation in fp32).  A kernel template of its own over the same body: generalising the fp32 instances
// in place changed their register allocation (see conv_epilogue_bf16.inc).
template <int MT_W, int NT_W, int WAVES_M, int WAVES_N, int TERMS, int IO>
__global__ void __launch_bounds__(256, 2) conv_split_kernel_bf16io(const ConvParams p) {
    static_assert(TERMS == 1 && IO >= 1 && IO <= 3, "bf16-resident tensors go with the plain-bf16 arithmetic");
    constexpr bool XB = (IO & 1) != 0;
    constexpr bool EPI_YB = (IO & 2) != 0;
#define VS_EPILOGUE_INC "conv_epilogue_bf16.inc"
#include "conv_split_body.inc"
#undef VS_EPILOGUE_INC
}

template <int MT_W, int NT_W, int WAVES_M, int WAVES_N, int TERMS, int IO>
__global__ void __launch_bounds__(256, 2) conv_split_tr_kernel_bf16io(const ConvParams p) {
    static_assert(TERMS == 1 && IO == 3, "bf16-resident x and y");
    constexpr bool XB = true;
    constexpr bool EPI_YB = true;
#define VS_EPILOGUE_INC "conv_epilogue_tr_bf16.inc"
#define VS_EPI_NO_FAST 1
#include "conv_split_body.inc"
#undef VS_EPI_NO_FAST
#undef VS_EPILOGUE_INC
}

// Ws[m_tile][tap][chunk][plane][64 lanes][8 bf16] from the fp32 fragment-order weights Wp[m_tile][tap][chunk][quad(2)][64][4]
// (pack_conv_kernel: lane l of quad qd, element e <-> row l&31, channel chunk*16 + 2*(4*qd + e) + (l>>5)); here lane l,
// element j <-> row l&31, channel chunk*16 + 8*(l>>5) + j.
__global__ void pack_split_kernel(const vs_split_pack q, int npl) {
    const long long total = (long long)q.MT_alloc * q.KT * q.nchunks * 64;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int lane = (int)(e & 63);
    const long long cell = e >> 6;                       // (m_tile, tap, chunk)
    const int row = lane & 31, kg = lane >> 5;
    const float *src = q.wp + cell * 512;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cl = 8 * kg + j, cp = cl >> 1, par = cl & 1;
        v[j] = src[(cp >> 2) * 256 + (row + 32 * par) * 4 + (cp & 3)];
    }
    u32x4 *dst = reinterpret_cast<u32x4 *>(q.ws) + cell * npl * 64 + lane;
    if (npl == 1) {
        unsigned d[4][1];
#pragma unroll
        for (int t = 0; t < 4; ++t) split_pair<1>(v[2 * t], v[2 * t + 1], d[t]);
        u32x4 o; o.x = d[0][0]; o.y = d[1][0]; o.z = d[2][0]; o.w = d[3][0];
        dst[0] = o;
    } else {
        unsigned d[4][3];
#pragma unroll
        for (int t = 0; t < 4; ++t) split_pair<3>(v[2 * t], v[2 * t + 1], d[t]);
        for (int pl = 0; pl < npl; ++pl) {
            u32x4 o; o.x = d[0][pl]; o.y = d[1][pl]; o.z = d[2][pl]; o.w = d[3][pl];
            dst[pl * 64] = o;
        }
    }
}

// ---- split-f16 (terms = 3): the weight planes carry ONE power-of-two scale per conv, s_w = 2^(14 - floor(log2 max|w|)) (largest weight below
// 2^15), found on the device (the pack never synchronises with the host): |w| bits -> atomic max -> {s_w, 1 / s_w} -> two f16 planes
__global__ void wabsmax_kernel(const float *wp, long long n, unsigned *maxbits) {
    unsigned m = 0;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        m = f16_maxkey(m, wp[e]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(maxbits, m);
}
__device__ __forceinline__ void pack_split_f16_body(const vs_split_pack &q, const unsigned *maxbits, unsigned blk) {
    const long long total = (long long)q.MT_alloc * q.KT * q.nchunks * 64;
    const long long e = (long long)blk * blockDim.x + threadIdx.x;
    const int eb = max(f16_key_exponent(*maxbits), F16_EB_MIN);      // (*maxbits: the largest f16_maxkey; all-zero weights: any scale does)
    const float sw = f16_scale(eb);
    if (e == 0) { q.wscale[0] = sw; q.wscale[1] = f16_inv_scale(eb); }
    if (e >= total) return;
    const int lane = (int)(e & 63);
    const long long cell = e >> 6;                       // (m_tile, tap, chunk)
    const int row = lane & 31, kg = lane >> 5;
    const float *src = q.wp + cell * 512;
    unsigned d[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cl = 8 * kg + 2 * t + u, cp = cl >> 1, par = cl & 1;
            v[u] = src[(cp >> 2) * 256 + (row + 32 * par) * 4 + (cp & 3)] * sw;
        }
        split_pair_h(v[0], v[1], d[t]);
    }
    u32x4 *dst = reinterpret_cast<u32x4 *>(q.ws) + cell * 2 * 64 + lane;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        u32x4 o; o.x = d[0][pl]; o.y = d[1][pl]; o.z = d[2][pl]; o.w = d[3][pl];
        dst[pl * 64] = o;
    }
}

__global__ void pack_split_f16_kernel(const vs_split_pack q, const unsigned *maxbits) { pack_split_f16_body(q, maxbits, blockIdx.x); }
__global__ void pack_split_f16_pair_kernel(const vs_split_pack q0, const vs_split_pack q1) {
    if (blockIdx.y == 0) pack_split_f16_body(q0, q0.maxbits, blockIdx.x);
    else pack_split_f16_body(q1, q1.maxbits, blockIdx.x);
}
// the planes of any number of handles in one launch (vs_conv_set_weights_batch): jobs and their first blocks in device memory (blk0[n] = the grid)
__global__ void pack_split_f16_multi_kernel(const vs_split_pack *__restrict__ jobs, const unsigned *__restrict__ blk0, int n) {
    int lo = 0, hi = n - 1;
    const unsigned b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blk0[mid] <= b) lo = mid;
        else hi = mid - 1;
    }
    const vs_split_pack q = jobs[lo];
    pack_split_f16_body(q, q.maxbits, b - blk0[lo]);
}

int pack_split_multi(const vs_split_pack *jobs_dev, const unsigned *blk0_dev, int n, unsigned nblocks, hipStream_t s) {
    hipLaunchKernelGGL(pack_split_f16_multi_kernel, dim3(nblocks), dim3(256), 0, s, jobs_dev, blk0_dev, n);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

// the f16 planes of two handles (vs_conv_set_weights_pair) in one launch; both carry their ready weight maximum
int pack_split_pair(const vs_split_pack &q0, const vs_split_pack &q1, hipStream_t s) {
    if (!q0.wscale || !q1.wscale || !q0.maxbits || !q1.maxbits) { set_error("pack_split_pair: scale buffers / maxima missing"); return VS_EINVAL; }
    const long long t0 = (long long)q0.MT_alloc * q0.KT * q0.nchunks * 64, t1 = (long long)q1.MT_alloc * q1.KT * q1.nchunks * 64;
    hipLaunchKernelGGL(pack_split_f16_pair_kernel, dim3((unsigned)ceil_div(std::max(t0, t1), 256), 2), dim3(256), 0, s, q0, q1);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

int pack_split(const vs_split_pack &q, hipStream_t s) {
    const long long total = (long long)q.MT_alloc * q.KT * q.nchunks * 64;
    if (q.terms == 3) {
        if (!q.wscale) { set_error("pack_split: the split-f16 arithmetic needs a scale buffer"); return VS_EINVAL; }
        const unsigned *maxbits = q.maxbits;
        if (!maxbits) {       // (vs_conv_set_math on a bound handle: a pass over the fp32 fragments; the slots of the regular packs stay untouched)
            if (!q.scratch) { set_error("pack_split: no scratch word for the weight maximum"); return VS_EINVAL; }
            VS_CHECK_HIP(hipMemsetAsync(q.scratch, 0, sizeof(unsigned), s));
            const long long n = total * 8;                   // fp32 fragment elements (zero padding included)
            hipLaunchKernelGGL(wabsmax_kernel, dim3((unsigned)std::min<long long>(ceil_div(n, 256 * 8), 1024)), dim3(256), 0, s, q.wp, n, q.scratch);
            VS_CHECK_HIP(hipGetLastError());
            maxbits = q.scratch;
        }
        hipLaunchKernelGGL(pack_split_f16_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, q, maxbits);
        VS_CHECK_HIP(hipGetLastError());
        return VS_OK;
    }
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, q, split_planes(q.terms));
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

template <int MT_W, int NT_W, int WAVES_M, int WAVES_N, int TERMS, int IO = 0>
static int launch_split_cfg(ConvParams p, int span, hipStream_t s) {
    constexpr int BN = 32 * NT_W * WAVES_N;
    constexpr int BM_TILES = MT_W * WAVES_M;
    constexpr int NPL = (TERMS == 1) ? 1 : (TERMS == 3 ? 2 : 3);
    // (transposed convs on the 128- / 64-row tiles of the split-f16 arithmetic: the instance with the polyphase store path)
    constexpr bool HAS_TR = ((IO == 0 && TERMS == 3) || (IO == 3 && TERMS == 1)) && (MT_W == 1) && (WAVES_M > 1);
    if constexpr (HAS_TR) {
        if (p.kind == VS_CONV_TRANSPOSE1D && !opt(OPT_NO_TR_EPI)) {
            auto kt = [] {
                if constexpr (IO == 0) return conv_split_tr_kernel<MT_W, NT_W, WAVES_M, WAVES_N, TERMS>;
                else return conv_split_tr_kernel_bf16io<MT_W, NT_W, WAVES_M, WAVES_N, TERMS, IO>;
            }();
            p.W = BN + span;
            const size_t lds_t = std::max<size_t>((size_t)2 * NPL * 2 * p.W * 16 + 32, (size_t)4 * 8 * (32 * NT_W) * sizeof(float));
            static bool attr_set_t = false;
            if (!attr_set_t) {
                VS_CHECK_HIP(hipFuncSetAttribute((const void *)kt, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_set_t = true;
            }
            dim3 grid_t((unsigned)ceil_div(p.N, BN), (unsigned)ceil_div(p.MT, BM_TILES), (unsigned)p.B);
            hipLaunchKernelGGL(kt, grid_t, dim3(256), lds_t, s, p);
            VS_CHECK_HIP(hipGetLastError());
            if (IO) set_last_kernel("conv_split_tr_kernel_bf16io<%d, %d, %d, %d, %d, %d>", MT_W, NT_W, WAVES_M, WAVES_N, TERMS, IO);
            else set_last_kernel("conv_split_tr_kernel<%d, %d, %d, %d, %d>", MT_W, NT_W, WAVES_M, WAVES_N, TERMS);
            return VS_OK;
        }
    }
    auto kern = [] {
        if constexpr (IO == 0) return conv_split_kernel<MT_W, NT_W, WAVES_M, WAVES_N, TERMS>;
        else return conv_split_kernel_bf16io<MT_W, NT_W, WAVES_M, WAVES_N, TERMS, IO>;
    }();
    p.W = BN + span;
    const size_t lds = std::max<size_t>((size_t)2 * NPL * 2 * p.W * 16 + (TERMS == 3 ? 32 : 0), (size_t)4 * 8 * (32 * NT_W) * sizeof(float));
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.N, BN), (unsigned)ceil_div(p.MT, BM_TILES), (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    if (IO) set_last_kernel("conv_split_kernel_bf16io<%d, %d, %d, %d, %d, %d>", MT_W, NT_W, WAVES_M, WAVES_N, TERMS, IO);
    else set_last_kernel("conv_split_kernel<%d, %d, %d, %d, %d>", MT_W, NT_W, WAVES_M, WAVES_N, TERMS);
    return VS_OK;
}

template <int TERMS>
static int launch_split_terms(const ConvParams &p, int cfg, int span, hipStream_t s) {
    switch (cfg) {
        case 0: return launch_split_cfg<1, 8, 4, 1, TERMS>(p, span, s);    // 128 x 256
        case 1:
        case 3: return launch_split_cfg<1, 4, 2, 2, TERMS>(p, span, s);    //  64 x 256
        case 2: return launch_split_cfg<1, 2, 1, 4, TERMS>(p, span, s);    //  32 x 256
        case 6: return launch_split_cfg<1, 1, 1, 4, TERMS>(p, span, s);    //  32 x 128 (single utterances: latency)
        case 4: return launch_split_cfg<2, 2, 2, 2, TERMS>(p, span, s);    // paired, 128 virtual rows x 128
        default: return launch_split_cfg<2, 2, 1, 4, TERMS>(p, span, s);   // paired,  64 virtual rows x 256
    }
}

// bf16-resident tensors: the 128 x 256 tile of the plain-bf16 arithmetic in every combination, the 64- and 32-row tiles for bf16 in AND out
// (their mixed variants spill 20 bytes per lane, which the hand-counted vmcnt waits of the main loop cannot tolerate)
template <int IO>
static int launch_split_io(const ConvParams &p, int cfg, int span, hipStream_t s) {
    if (cfg == 0) return launch_split_cfg<1, 8, 4, 1, 1, IO>(p, span, s);
    if constexpr (IO == 3) {       // bf16 in AND out: the narrow stages' own convs (transposed 64 -> 32, unfused pairs); these variants do not spill
        if (cfg == 1 || cfg == 3) return launch_split_cfg<1, 4, 2, 2, 1, 3>(p, span, s);
        if (cfg == 2) return launch_split_cfg<1, 2, 1, 4, 1, 3>(p, span, s);
    }
    set_error("launch_split: bf16-resident tensors: this tile shape has no instance (fp32 <-> bf16 conversions on the 128-row tile only)");
    return VS_EUNSUPPORTED;
}

int launch_split(const ConvParams &p, int cfg, int terms, int span, hipStream_t s) {
    const int io = (p.x_bf16 ? 1 : 0) | (p.y_bf16 ? 2 : 0);
    if (io) {
        if (terms != 1) { set_error("launch_split: bf16-resident tensors need the plain-bf16 arithmetic (VS_MATH_BF16)"); return VS_EUNSUPPORTED; }
        return io == 1 ? launch_split_io<1>(p, cfg, span, s) : io == 2 ? launch_split_io<2>(p, cfg, span, s) : launch_split_io<3>(p, cfg, span, s);
    }
    if (terms == 6) return launch_split_terms<6>(p, cfg, span, s);
    if (terms == 3) return launch_split_terms<3>(p, cfg, span, s);
    if (terms == 1) return launch_split_terms<1>(p, cfg, span, s);
    set_error("launch_split: unsupported term count %d", terms);
    return VS_EUNSUPPORTED;
}

}  // namespace vs
