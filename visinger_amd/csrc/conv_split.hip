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

int split_planes(int terms) { return terms == 1 ? 1 : (terms == 3 ? 2 : 3); }

template <int MT_W, int NT_W, int WAVES_M, int WAVES_N, int TERMS>
__global__ void __launch_bounds__(256, 2) conv_split_kernel(const ConvParams p) {
    static_assert(WAVES_M * WAVES_N == 4, "four waves: a wave stages four consecutive channels of a 16-channel chunk");
    constexpr int NPL = (TERMS == 1) ? 1 : (TERMS == 3 ? 2 : 3);
    constexpr int BN = 32 * NT_W * WAVES_N;
    constexpr int MAXW = BN + MAX_SPAN;
    constexpr int CIT = (MAXW + 63) / 64;        // column iterations per staged row
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M;
    const int wn = wave / WAVES_M;
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * BN;
    const int mt0 = (blockIdx.y * WAVES_M + wm) * MT_W;
    const int W = p.W;
    const int PLSZ = 2 * W * 4;                  // dwords per plane: [k-group][column][4 dwords = 8 bf16]
    unsigned *const lbuf0 = reinterpret_cast<unsigned *>(smem);
    unsigned *const lbuf1 = lbuf0 + NPL * PLSZ;
    const float *const xb = p.x + (long long)b * p.x_bs;
    const float *const maskb = p.mask ? p.mask + (long long)b * p.Tin : nullptr;

    // ---- tap range of this wave (polyphase transposed conv), as in conv_mfma_kernel ----
    int tap_b = 0, tap_e = p.KT;
    if (p.kind == VS_CONV_TRANSPOSE1D && (p.c_out & 31) == 0) {
        int lo_t = p.KT, hi_t = 0;
#pragma unroll
        for (int i = 0; i < MT_W; ++i) {
            const int phase = ((mt0 + i) * 32) / p.c_out;
            if (phase < p.up) {
                const int num_lo = -(phase + p.uppad);
                const int dlo = (num_lo >= 0) ? (num_lo + p.up - 1) / p.up : -((-num_lo) / p.up);
                const int num_hi = p.upK - 1 - phase - p.uppad;
                const int dhi = (num_hi >= 0) ? num_hi / p.up : -((-num_hi + p.up - 1) / p.up);
                lo_t = min(lo_t, dlo - p.dmin);
                hi_t = max(hi_t, dhi - p.dmin + 1);
            }
        }
        tap_b = max(0, lo_t);
        tap_e = min(p.KT, hi_t);
        if (tap_e <= tap_b) { tap_b = 0; tap_e = 1; }     // padding-only wave: keeps its seat at the chunk barriers
    }

    // accumulators start from the bias (+ per-item conditioning bias) of their row
    const float *const bbias = p.bias_b ? p.bias_b + (long long)b * p.bias_b_bs : nullptr;
    f32x16 acc[MT_W][NT_W];
#pragma unroll
    for (int i = 0; i < MT_W; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rt = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int m = (mt0 + i) * 32 + rt;
            float bv = p.biasp[m];
            if (bbias) {
                int row;
                if constexpr (MT_W == 2) {
                    row = (i & 1) * p.Hh + min((mt0 >> 1) * 32 + rt, p.Hh - 1);
                } else {
                    const int mc = min(m, p.M - 1);
                    row = (p.kind == VS_CONV_TRANSPOSE1D) ? mc % p.c_out : mc;
                }
                bv += bbias[row];
            }
#pragma unroll
            for (int j = 0; j < NT_W; ++j) acc[i][j][r] = bv;
        }

    // ---- staging: wave w owns channels 4w .. 4w+3 of every chunk (k-group w/2, dwords (w&1)*2 .. +1 of the 16-B cell) ----
    float st[4][CIT];
    float mk[CIT];
    const int in_act = p.in_act;
    const __amdgpu_buffer_rsrc_t xsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)((long long)p.Cin * p.Tin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t msrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(maskb ? maskb : xb), 0, p.Tin * 4, 0x00020000);
    auto stage_load = [&](int chunk) __attribute__((always_inline)) {
        const int nbase = n0 + p.lo + lane;
        if (in_act >= VS_IN_MASK) {
#pragma unroll
            for (int i = 0; i < CIT; ++i)
                mk[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(msrc, nbase * 4 + i * 256, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = min(chunk * CK + 4 * wave + j, p.Cin - 1);
            const int voff = (ci * p.Tin + nbase) * 4;
#pragma unroll
            for (int i = 0; i < CIT; ++i)
                st[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, voff + i * 256, 0, 0));
        }
    };
    const bool time_edge = (n0 + p.lo < 0) || (n0 + p.lo + W > p.Tin);
    auto stage_store = [&](unsigned *buf, int chunk) __attribute__((always_inline)) {
        auto run = [&](auto edge_tag, auto act_tag) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            constexpr int ACT = decltype(act_tag)::value;
            unsigned *const dst0 = buf + ((wave >> 1) * W + lane) * 4 + (wave & 1) * 2;
#pragma unroll
            for (int i = 0; i < CIT; ++i) {
                const int col = lane + 64 * i;
                const int n = n0 + p.lo + col;
                const bool okn = (n >= 0) && (n < p.Tin);
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = st[j][i];
                    if constexpr (EDGE) v[j] = (okn && (chunk * CK + 4 * wave + j < p.Cin)) ? v[j] : 0.f;
                    if constexpr (ACT == VS_IN_LRELU || ACT == VS_IN_LRELU_MASK) v[j] = fmaxf(v[j], 0.1f * v[j]);
                    if constexpr (ACT >= VS_IN_MASK) v[j] *= mk[i];
                }
                unsigned d0[NPL], d1[NPL];
                split_pair<NPL>(v[0], v[1], d0);
                split_pair<NPL>(v[2], v[3], d1);
                if (64 * (i + 1) <= BN || col < W) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        *reinterpret_cast<uint2 *>(dst0 + pl * PLSZ + i * 256) = make_uint2(d0[pl], d1[pl]);
                }
            }
        };
        const bool edge = time_edge || (chunk * CK + CK > p.Cin);
        if (edge) {
            if (in_act == VS_IN_NONE) run(std::true_type{}, std::integral_constant<int, VS_IN_NONE>{});
            else if (in_act == VS_IN_LRELU) run(std::true_type{}, std::integral_constant<int, VS_IN_LRELU>{});
            else if (in_act == VS_IN_MASK) run(std::true_type{}, std::integral_constant<int, VS_IN_MASK>{});
            else run(std::true_type{}, std::integral_constant<int, VS_IN_LRELU_MASK>{});
        } else {
            if (in_act == VS_IN_NONE) run(std::false_type{}, std::integral_constant<int, VS_IN_NONE>{});
            else if (in_act == VS_IN_LRELU) run(std::false_type{}, std::integral_constant<int, VS_IN_LRELU>{});
            else if (in_act == VS_IN_MASK) run(std::false_type{}, std::integral_constant<int, VS_IN_MASK>{});
            else run(std::false_type{}, std::integral_constant<int, VS_IN_LRELU_MASK>{});
        }
    };

    const int lhalf = lane >> 5;
    const int l31 = lane & 31;

    // ---------------------------------------------------------------------------------------------- main loop
    // Steps s = (chunk, tap).  A fragments of step s+1 (NPL 16-byte loads per row tile, L2/L1-resident) are requested at the
    // start of step s into the other of two NAMED register sets; activations as in conv_mfma_kernel: chunk c+2 is requested
    // at the first tap of chunk c and lives in registers, chunk c+1 is split and written to the other LDS buffer there.
    //
    // The A loads are inline asm with HAND-COUNTED vmcnt waits.  Left to hipcc, the wait in front of a step's first MFMA is
    // vmcnt(0) (its scoreboard goes flat across the conditional staging block and the back edge), which also waits for the
    // fragments requested a few instructions earlier for the NEXT step and for the activation loads of chunk c+2: one exposed
    // L2 round trip per step and one HBM round trip per chunk -- at bf16 MFMA speed that was 40 % of the kernel.  vmcnt counts
    // loads in issue order, so "fragments of THIS step have landed" is vmcnt(n) with n = everything issued after them: the
    // activation loads of the previous step (if it staged), this step's A prefetch, this step's activation loads (if it
    // stages).  (hipcc's own waits for its buffer loads do not know about the asm loads and are therefore merely stricter.)
    // The count assumes no other vector-memory instruction in the loop: csrc/build.py fails the build if an instance of this
    // kernel uses scratch (a spill would be one).
    const int ntaps = tap_e - tap_b;
    const int nsteps = p.nchunks * ntaps;
    const u32x4 *wbase[MT_W];
#pragma unroll
    for (int i = 0; i < MT_W; ++i)
        wbase[i] = reinterpret_cast<const u32x4 *>(p.wp) + (long long)(mt0 + i) * p.KT * p.nchunks * (NPL * 64) + lane;
    u32x4 a0[MT_W][NPL], a1[MT_W][NPL];
    auto load_a = [&](u32x4 (&dst)[MT_W][NPL], int chunk, int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT_W; ++i) {
            const u32x4 *src = wbase[i] + ((long long)tap * p.nchunks + chunk) * (NPL * 64);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst[i][0]) : "v"(src) : "memory");
            if constexpr (NPL > 1) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=&v"(dst[i][1]) : "v"(src) : "memory");
            if constexpr (NPL > 2) asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=&v"(dst[i][2]) : "v"(src) : "memory");
        }
    };
    constexpr int NA = MT_W * NPL;                              // A loads per step
    const int NY = ((in_act >= VS_IN_MASK) ? 5 : 4) * CIT;       // activation (+ mask) loads of one stage_load
    // wait until at most n vector-memory loads are outstanding (n < 64), then pin the fragment registers behind the wait
    auto wait_a = [&](u32x4 (&a)[MT_W][NPL], int n) __attribute__((always_inline)) {
        if (n >= NA + 2 * NY && NA + 2 * 5 * CIT < 64) {
            if (NY == 4 * CIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + 8 * CIT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + 10 * CIT < 64 ? NA + 10 * CIT : 0) : "memory");
        } else if (n >= NA + NY) {
            if (NY == 4 * CIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + 4 * CIT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + 5 * CIT) : "memory");
        } else if (n >= NA) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < MT_W; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) asm volatile("" : "+v"(a[i][pl]));
    };
    int pc = 0, pt = tap_b;      // (chunk, tap) of the step after the current one
    auto advance = [&]() __attribute__((always_inline)) { if (++pt == tap_e) { pt = tap_b; ++pc; } };
    if (nsteps > 0) { load_a(a0, pc, pt); advance(); }

    stamp(p, 0);
    stage_load(0);
    stage_store(lbuf0, 0);          // (hipcc waits vmcnt(0) for the staged registers: covers a0 as well)
    if (p.nchunks > 1) stage_load(1);
    __syncthreads();
    stamp(p, 1);

    int chunk = 0, tap = tap_b, s = 0;
    int ny_prev = (p.nchunks > 1) ? NY : 0;      // activation loads issued after the A loads of the current step
    auto step = [&](u32x4 (&acur)[MT_W][NPL], u32x4 (&apre)[MT_W][NPL]) __attribute__((always_inline)) {
        const unsigned *cur = (chunk & 1) ? lbuf1 : lbuf0;
        const bool more = (chunk + 1 < p.nchunks);
        const bool first = (tap == tap_b);
        if (first && more) stage_store((chunk & 1) ? lbuf0 : lbuf1, chunk + 1);
        int young = ny_prev;
        if (s + 1 < nsteps) { load_a(apre, pc, pt); advance(); young += NA; }
        ny_prev = 0;
        if (first && chunk + 2 < p.nchunks) { stage_load(chunk + 2); young += NY; ny_prev = NY; }
        const unsigned *xs = cur + (lhalf * W + wn * (NT_W * 32) + l31 - p.lo + (p.off0 + tap * p.tstep)) * 4;
        u32x4 bf[NPL], bn[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) bf[pl] = *reinterpret_cast<const u32x4 *>(xs + pl * PLSZ);
        wait_a(acur, young);
#pragma unroll
        for (int j = 0; j < NT_W; ++j) {
            if (j + 1 < NT_W) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) bn[pl] = *reinterpret_cast<const u32x4 *>(xs + pl * PLSZ + (j + 1) * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT_W; ++i) {
                auto mm = [&](int ta, int tb) __attribute__((always_inline)) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, acur[i][ta]),
                                                                        __builtin_bit_cast(bf16x8, bf[tb]), acc[i][j], 0, 0, 0);
                };
                // smallest terms first
                if constexpr (TERMS == 6) { mm(1, 1); mm(2, 0); mm(0, 2); }
                if constexpr (TERMS >= 3) { mm(1, 0); mm(0, 1); }
                mm(0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) bf[pl] = bn[pl];
        }
        if (++tap == tap_e) {
            __syncthreads();
            tap = tap_b;
            ++chunk;
        }
        ++s;
    };
    while (s < nsteps) {
        step(a0, a1);
        if (s < nsteps) step(a1, a0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of the asm loads may be in flight past here

    stamp(p, 2);
#include "conv_epilogue.inc"
    if (p.stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        stamp(p, 3);
    }
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

int pack_split(const vs_split_pack &q, hipStream_t s) {
    const long long total = (long long)q.MT_alloc * q.KT * q.nchunks * 64;
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, q, split_planes(q.terms));
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

template <int MT_W, int NT_W, int WAVES_M, int WAVES_N, int TERMS>
static int launch_split_cfg(ConvParams p, int span, hipStream_t s) {
    constexpr int BN = 32 * NT_W * WAVES_N;
    constexpr int BM_TILES = MT_W * WAVES_M;
    constexpr int NPL = (TERMS == 1) ? 1 : (TERMS == 3 ? 2 : 3);
    auto kern = conv_split_kernel<MT_W, NT_W, WAVES_M, WAVES_N, TERMS>;
    p.W = BN + span;
    const size_t lds = std::max<size_t>((size_t)2 * NPL * 2 * p.W * 16, (size_t)4 * 8 * (32 * NT_W) * sizeof(float));
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.N, BN), (unsigned)ceil_div(p.MT, BM_TILES), (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    set_last_kernel("conv_split_kernel<%d, %d, %d, %d, %d>", MT_W, NT_W, WAVES_M, WAVES_N, TERMS);
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

int launch_split(const ConvParams &p, int cfg, int terms, int span, hipStream_t s) {
    if (terms == 6) return launch_split_terms<6>(p, cfg, span, s);
    if (terms == 1) return launch_split_terms<1>(p, cfg, span, s);
    set_error("launch_split: unsupported term count %d", terms);
    return VS_EUNSUPPORTED;
}

}  // namespace vs
