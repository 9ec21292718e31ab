// conv_ktap.hip -- conv_ktap_kernel<KT, ACT>: the 128 x 256 tile of the split-f16 x3 conv engine (conv_split_kernel<1, 8, 4, 1, 3>, conv_split.hip)
// with the tap count a template constant and the whole 16-channel chunk -- KT taps x 8 column tiles x 3 cross products -- as ONE straight-line
// block in which every instruction that is not an MFMA sits in the shadow of one.  (Reference work: the stride-1 Conv1d's of the HiFi-GAN
// generator's ResBlock1 / conv_pre, modules/visinger/decoder.py:72-101, 40-43, and the FFN's conv_1, modules/rel_transformer.py:336-345.)
//
// Why (rounds 4-5): a wave of the tile kernel alone on its SIMD keeps the matrix pipe 48 % busy -- per (chunk, tap) step 768 cycles
// of MFMA issue and ~840 cycles of everything else (the staging of the next chunk: load, input transform, tile maximum, scale, split into two f16
// planes, LDS writes; the weight-fragment requests; the wait tree; ~45 SALU and ~14 branches of step bookkeeping), one AFTER the other: the staging sits
// in a conditional block of the first tap, and between the three MFMAs of a column tile there are two LDS reads and nothing else.  An MFMA
// 32x32x16 occupies the pipe for 32 cycles and the wave's issue port for 8: up to five or six other instructions issue for free behind each
// (MI355X_MICROARCH.md, "vector-instruction ISSUE cost").  Here
//   * the taps are unrolled (KT = 3 / 7 / 11 ...): a chunk body has no branch, no run-time tap bookkeeping, LDS offsets are immediates;
//   * the staging of chunk c + 1 is cut into ~55 micro-operations of 3 - 5 instructions (one value's transform, half a pair's split, one row
//     of loads ...) that are dealt out over the 3 * 8 * KT MFMA gaps of chunk c by a compile-time schedule, in program order, pinned by
//     scheduling barriers;
//   * nothing in the loop is predicated: columns outside the tensor or the staged window and chunks past the last one are buffer loads the
//     descriptor's range check answers with zeros (no memory traffic), the running tile maximum ignores zeros, rescales are a multiply by one;
//   * every wait is hipcc's own count on a straight-line scoreboard (no hand-counted vmcnt, no wait tree).
// Arithmetic, operand planes, MFMA order and epilogue are those of conv_split_kernel<1, 8, 4, 1, 3>: outputs are BIT-IDENTICAL to it
// (tests/test_conv_ktap_gpu.py), so every parity statement about that instance carries over.
#include "conv_common.h"

#include <type_traits>
#include <utility>

namespace vs {

namespace {

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

#define KT_SB() __builtin_amdgcn_sched_barrier(0)

// schedule of a chunk body: the first of the NOPS staging micro-operations that has not been issued when MFMA gap g (of NG) starts; they are
// spread evenly over the gaps [NG / 8, NG - 2) -- the loads requested late in the previous chunk get the first eighth of this one on top
template <int NG, int NOPS>
constexpr int ktap_ops_begin(int g) {
    constexpr int G_FIRST = NG / 8, G_LAST = NG - 2;
    if (g <= G_FIRST) return 0;
    if (g >= G_LAST) return NOPS;
    return (int)(((long long)(g - G_FIRST) * NOPS + (G_LAST - G_FIRST) - 1) / (G_LAST - G_FIRST));
}

}  // namespace

constexpr int KTAP_WP = 256 + MAX_SPAN;                 // constant LDS pitch of a staged row (columns): plane / buffer offsets are immediates
constexpr int KTAP_PLB = 2 * KTAP_WP * 16;              // bytes per plane: [k-group 2][column][8 f16]
constexpr int KTAP_BUFB = 2 * KTAP_PLB;                 // bytes per buffer (hi + lo plane)
constexpr int KTAP_SMAX = 2 * KTAP_BUFB;                // [2][4] exponents of the waves' staged maxima
constexpr int KTAP_DUMP = KTAP_SMAX + 64;               // [4 waves][64 lanes] words: where the lanes other than 0 put their copy of the exponent
constexpr int KTAP_LDS = KTAP_DUMP + 4 * 64 * 4;

template <int KT, int ACT>
__global__ void __launch_bounds__(256, 2) conv_ktap_kernel(const ConvParams p) {
    constexpr int MT_W = 1, NT_W = 8, WAVES_M = 4;
    constexpr bool F16 = true;
    constexpr int BN = 256, CIT = 5;
    constexpr int NG = 3 * NT_W * KT;                   // MFMA gaps of a chunk body
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char *const lds = reinterpret_cast<char *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = 0;
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * BN;
    const int mt0 = blockIdx.y * WAVES_M + wave;
    const int lhalf = lane >> 5;
    const int l31 = lane & 31;
    const int W = BN + (KT - 1) * p.tstep;              // staged window (columns)
    const float *const xb = p.x + (long long)b * p.x_bs;
    const float *const maskb = p.mask ? p.mask + (long long)b * p.Tin : nullptr;
    const float *const bbias = p.bias_b ? p.bias_b + (long long)b * p.bias_b_bs : nullptr;
    const int nchunks = p.nchunks;

    auto bias_of = [&](int i, int r) __attribute__((always_inline)) -> float {
        const int rt = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int m = (mt0 + i) * 32 + rt;
        float bv = p.biasp[m];
        if (bbias) bv += bbias[min(m, p.M - 1)];
        return bv;
    };

    f32x16 acc[MT_W][NT_W];
#pragma unroll
    for (int j = 0; j < NT_W; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    // ---- staging state: wave w owns channels 4w .. 4w + 3 of every chunk.  Column it of lane l is n0 + lo + l + 64 * it; outside the tensor or the
    // window its byte offset is past the descriptor's range: the load returns 0 without touching memory
    float st[4][CIT];
    float mk[CIT];
    int vcol[CIT];
#pragma unroll
    for (int it = 0; it < CIT; ++it) {
        const int col = lane + 64 * it;
        const int n = n0 + p.lo + col;
        vcol[it] = (n >= 0 && n < p.Tin && col < W) ? n * 4 : (int)0x7ffffff0;
    }
    const int xbytes = (int)((long long)p.Cin * p.Tin * 4);
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xsrc0 = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, 0, 0x00020000);        // (no records: every load returns 0)
    if constexpr (ACT >= VS_IN_MASK) {
        const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc((void *)maskb, 0, p.Tin * 4, 0x00020000);
#pragma unroll
        for (int it = 0; it < CIT; ++it) mk[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(msrc, vcol[it], 0, 0));
    }
    const int rowb = p.Tin * 4;                           // bytes per channel row

    // ---- LDS addresses (bytes)
    const unsigned wr_lane = (unsigned)(((wave >> 1) * KTAP_WP + lane) * 16 + (wave & 1) * 8);       // + it * 1024 + plane * PLB + buffer * BUFB
    const unsigned rd_lane = (unsigned)((lhalf * KTAP_WP + l31 - p.lo + p.off0) * 16);               // + tap * tstep * 16 + j * 512 + plane * PLB + buffer * BUFB
    const int tstep16 = p.tstep * 16;
    const unsigned slot_lane = (lane == 0) ? (unsigned)(KTAP_SMAX + wave * 4) : (unsigned)(KTAP_DUMP + (wave * 64 + lane) * 4);     // + set * 16

    // ---- weight fragments: Ws[m_tile][tap][chunk][plane][64 lanes][8 f16]
    const char *const wrow = reinterpret_cast<const char *>(p.wp) + (long long)mt0 * KT * nchunks * 2048;
    const unsigned wl16 = (unsigned)lane * 16u;
    const long long a_dtap = (long long)nchunks * 2048;
    u32x4 afr[2][2];                                      // [ring slot][plane]
    auto load_a = [&](u32x4 (&dst)[2], long long off) __attribute__((always_inline)) {
        const char *src = wrow + off;
        dst[0] = *reinterpret_cast<const u32x4 *>(src + wl16);
        dst[1] = *reinterpret_cast<const u32x4 *>(src + wl16 + 1024);
    };

    // ---- the running scale of the tile (split-f16 arithmetic, conv_common.h)
    int eb_cur = F16_EB_MIN;
    int resc_exp = 0;                                     // != 0: the accumulators take 2^resc_exp in front of the next chunk
    float sx = 1.f;
    unsigned mkey = 0u;
    int ebw = 0;
    int4 sl = make_int4(0, 0, 0, 0);
    float sv[4];                                          // scaled values of the column in flight
    unsigned dpl[2][2];                                   // [pair][plane]

    // The staging of chunk `cs` as micro-operations k = 0 .. NOPS - 1, to be executed in this order; SET = cs & 1 (LDS buffer and exponent slots).
    // `cl` = the chunk whose loads replace the consumed registers (cs + 1), `cl_ok` whether it exists.
    constexpr int OPS_A = 4 * CIT;                        // value (it, j): input transform, running maximum key
    constexpr int OPS_B = 3;                              // wave maximum (six DPP steps), exponent to LDS
    constexpr int OPS_C = 2;                              // barrier + the four waves' exponents; tile scale
    constexpr int OPS_D = 6 * CIT;                        // per column: four half-pair splits, the LDS writes, the loads of the next chunk
    constexpr int NOPS = OPS_A + OPS_B + OPS_C + OPS_D;
    auto stage_op = [&](auto set_c, auto k_c, const __amdgpu_buffer_rsrc_t &src_l, int soff_l) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
        constexpr int k = decltype(k_c)::value;
        if constexpr (k < OPS_A) {
            constexpr int it = k / 4, j = k % 4;
            float v = st[j][it];
            if constexpr (ACT == VS_IN_LRELU || ACT == VS_IN_LRELU_MASK) v = fmaxf(v, 0.1f * v);
            if constexpr (ACT >= VS_IN_MASK) v *= mk[it];
            st[j][it] = v;
            mkey = f16_maxkey(k == 0 ? 0u : mkey, v);
        } else if constexpr (k < OPS_A + OPS_B) {
            constexpr int q = k - OPS_A;
            if constexpr (q == 0) {
                int v = f16_key_exponent(mkey);
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));      // row_shr:1
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));      // row_shr:2
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));      // row_shr:4
                ebw = v;
            } else if constexpr (q == 1) {
                int v = ebw;
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));      // row_shr:8
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));     // row_bcast:15
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));     // row_bcast:31
                ebw = v;
            } else {
                const int m = __builtin_amdgcn_readlane(ebw, 63);
                *reinterpret_cast<int *>(lds + slot_lane + SET * 16) = m;                  // (lane 0: the wave's slot; the others: their own dump word)
            }
        } else if constexpr (k < OPS_A + OPS_B + OPS_C) {
            constexpr int q = k - OPS_A - OPS_B;
            if constexpr (q == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                sl = *reinterpret_cast<const int4 *>(lds + KTAP_SMAX + SET * 16);
            } else {
                const int eb = __builtin_amdgcn_readfirstlane(max(max(max(sl.x, sl.y), max(sl.z, sl.w)), F16_EB_MIN));
                const int eb_new = max(eb_cur, eb);
                resc_exp = eb_cur - eb_new;
                eb_cur = eb_new;
                sx = f16_scale(eb_cur);
            }
        } else {
            constexpr int q = k - OPS_A - OPS_B - OPS_C;
            constexpr int it = q / 6, u = q % 6;
            if constexpr (u < 4) {
                constexpr int pr = u >> 1;                // channel pair (0, 1) / (2, 3)
                if constexpr ((u & 1) == 0) {             // first half: scale, high plane
                    sv[0] = st[2 * pr][it] * sx;
                    sv[1] = st[2 * pr + 1][it] * sx;
                    const f32x2 v = {sv[0], sv[1]};
                    dpl[pr][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
                } else {                                  // second half: low plane = RNE_f16(value - high)
                    const f32x2 v = {sv[0], sv[1]};
                    const f16x2 h = __builtin_bit_cast(f16x2, dpl[pr][0]);
                    const f32x2 r = v - __builtin_convertvector(h, f32x2);
                    dpl[pr][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
                }
            } else if constexpr (u == 4) {
                char *const dst = lds + wr_lane + SET * KTAP_BUFB + it * 1024;
                *reinterpret_cast<uint2 *>(dst) = make_uint2(dpl[0][0], dpl[1][0]);
                *reinterpret_cast<uint2 *>(dst + KTAP_PLB) = make_uint2(dpl[0][1], dpl[1][1]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    st[j][it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src_l, vcol[it], soff_l + j * rowb, 0));
            }
        }
    };
    // byte offset of the first of this wave's four channel rows of chunk c
    auto soff_of = [&](int c) __attribute__((always_inline)) { return (c * CK + 4 * wave) * rowb; };

    // ---- prologue: chunk 0 staged, chunk 1 in the registers, the fragments of (chunk 0, tap 0) requested
    load_a(afr[0], 0);
    stamp(p, 0);
#pragma unroll
    for (int it = 0; it < CIT; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) st[j][it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, vcol[it], soff_of(0) + j * rowb, 0));
    {
        const int s1 = soff_of(1);
        const bool ok1 = nchunks > 1;
        static_for<NOPS>([&](auto k_c) { stage_op(std::integral_constant<int, 0>{}, k_c, ok1 ? xsrc : xsrc0, s1); });
    }
    resc_exp = 0;                                         // (chunk 0 sets the first scale: nothing to rescale)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamp(p, 1);

    // ---- main loop: chunk c on LDS buffer PAR, staging chunk c + 1 into buffer PAR ^ 1, loading chunk c + 2
    // schedule: micro-operation k of the staging goes behind MFMA gap g0 + k * gstep (several per gap where NOPS > the gaps left)
    int c = 0;
    long long aoff = 0;                                   // byte offset of (chunk c, tap 0) in this wave's fragment rows
    auto chunk_body = [&](auto par_c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        if (resc_exp != 0) {                              // (wave-uniform, rare: the staged chunk raised the tile's largest magnitude)
            const float f = __builtin_ldexpf(1.f, resc_exp);
#pragma unroll
            for (int j = 0; j < NT_W; ++j) acc[0][j] *= f;
            resc_exp = 0;
        }
        const bool ok2 = (c + 2 < nchunks);
        const __amdgpu_buffer_rsrc_t src_l = ok2 ? xsrc : xsrc0;
        const int soff_l = soff_of(c + 2);
        unsigned xs = rd_lane + PAR * KTAP_BUFB;
        u32x4 bfr[2][2];                                  // [ring slot][plane]
        bfr[0][0] = *reinterpret_cast<const u32x4 *>(lds + xs);
        bfr[0][1] = *reinterpret_cast<const u32x4 *>(lds + xs + KTAP_PLB);
        static_for<KT>([&](auto tap_c) {
            constexpr int tap = decltype(tap_c)::value;
            constexpr int AS = (tap + PAR * KT) & 1;      // fragment ring slot of this step
            // fragments of the next step: next tap of this chunk, or tap 0 of the next chunk (the last chunk re-reads its own: never used)
            if constexpr (tap + 1 < KT) load_a(afr[AS ^ 1], aoff + (tap + 1) * a_dtap);
            else load_a(afr[AS ^ 1], aoff + ((c + 1 < nchunks) ? 2048 : 0));
            const unsigned xs_next = xs + tstep16;
            static_for<NT_W>([&](auto j_c) {
                constexpr int j = decltype(j_c)::value;
                constexpr int t8 = tap * NT_W + j;
                constexpr int BS = t8 & 1;
                constexpr int g = 3 * t8;
                // B fragments of the next column tile (or of the next tap's first), under this tile's MFMAs
                if constexpr (j + 1 < NT_W) {
                    bfr[BS ^ 1][0] = *reinterpret_cast<const u32x4 *>(lds + xs + (j + 1) * 512);
                    bfr[BS ^ 1][1] = *reinterpret_cast<const u32x4 *>(lds + xs + (j + 1) * 512 + KTAP_PLB);
                } else if constexpr (tap + 1 < KT) {
                    bfr[BS ^ 1][0] = *reinterpret_cast<const u32x4 *>(lds + xs_next);
                    bfr[BS ^ 1][1] = *reinterpret_cast<const u32x4 *>(lds + xs_next + KTAP_PLB);
                }
                auto side = [&](auto g_c) __attribute__((always_inline)) {
                    constexpr int gg = decltype(g_c)::value;
                    constexpr int k0 = ktap_ops_begin<NG, NOPS>(gg), k1 = ktap_ops_begin<NG, NOPS>(gg + 1);
                    static_for<k1 - k0>([&](auto d_c) { stage_op(std::integral_constant<int, PAR ^ 1>{}, std::integral_constant<int, k0 + decltype(d_c)::value>{}, src_l, soff_l); });
                };
                KT_SB();
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[AS][1]), __builtin_bit_cast(f16x8, bfr[BS][0]), acc[0][j], 0, 0, 0);
                KT_SB();
                side(std::integral_constant<int, g>{});
                KT_SB();
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[AS][0]), __builtin_bit_cast(f16x8, bfr[BS][1]), acc[0][j], 0, 0, 0);
                KT_SB();
                side(std::integral_constant<int, g + 1>{});
                KT_SB();
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afr[AS][0]), __builtin_bit_cast(f16x8, bfr[BS][0]), acc[0][j], 0, 0, 0);
                KT_SB();
                side(std::integral_constant<int, g + 2>{});
                KT_SB();
            });
            xs = xs_next;
        });
        // every wave has read buffer PAR for the last time and finished writing buffer PAR ^ 1
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        aoff += 2048;
        ++c;
    };
    while (true) {
        chunk_body(std::integral_constant<int, 0>{});
        if (c >= nchunks) break;
        chunk_body(std::integral_constant<int, 1>{});
        if (c >= nchunks) break;
    }
    stamp(p, 2);

    // ---- epilogue: that of conv_split_kernel<1, 8, 4, 1, 3> (conv_split_body.inc): the accumulators hold s_x * s_w * sum
    // (eb_cur: the staging of the chunk past the last one saw zeros only and left it alone)
    const float acc_inv = f16_inv_scale(eb_cur) * p.wscale[1];
    constexpr bool EPI_BIAS_IN_LOOP = false;
    float bias_lane[MT_W];
    {
        const int m = mt0 * 32 + l31;
        float bv = p.biasp[m];
        if (bbias) bv += bbias[min(m, p.M - 1)];
        bias_lane[0] = bv;
    }
#define VS_ACC(i, j, r) fmaf(acc[i][j][r], acc_inv, bias_of(i, r))
#define VS_ACC_FAST(i, j, r) acc[i][j][r]
#define VS_ACC_PREP(name, i) float name[16];
#define VS_ACC_B(i, j, r, name) fmaf(acc[i][j][r], acc_inv, bias_of(i, r))
#define VS_ROW_FINISH(v, i, row0, lrow_, rpi_)                                                         \
    {                                                                                                  \
        float bb_;                                                                                     \
        if constexpr ((rpi_) == 1) bb_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bias_lane[i]), (row0)));   \
        else bb_ = __shfl(bias_lane[i], (row0) + (lrow_));                                             \
        v.x = fmaf(v.x, acc_inv, bb_); v.y = fmaf(v.y, acc_inv, bb_); v.z = fmaf(v.z, acc_inv, bb_); v.w = fmaf(v.w, acc_inv, bb_);   \
    }
#include "conv_epilogue.inc"
#undef VS_ACC
#undef VS_ACC_PREP
#undef VS_ACC_B
#undef VS_ACC_FAST
#undef VS_ROW_FINISH
    (void)F16; (void)EPI_BIAS_IN_LOOP; (void)wn;
    if (p.stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        stamp(p, 3);
    }
}

// ---------------------------------------------------------------------------------------------------------------- host side
bool ktap_taps(int kt) { return kt == 3 || kt == 7 || kt == 11; }

template <int KT, int ACT>
static int launch_ktap_inst(const ConvParams &p, hipStream_t s) {
    auto kern = conv_ktap_kernel<KT, ACT>;
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.N, 256), (unsigned)ceil_div(p.MT, 4), (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), (size_t)KTAP_LDS, s, p);
    VS_CHECK_HIP(hipGetLastError());
    set_last_kernel("conv_ktap_kernel<%d, %d>", KT, ACT);
    return VS_OK;
}

template <int KT>
static int launch_ktap_kt(const ConvParams &p, hipStream_t s) {
    switch (p.in_act) {
        case VS_IN_NONE: return launch_ktap_inst<KT, VS_IN_NONE>(p, s);
        case VS_IN_LRELU: return launch_ktap_inst<KT, VS_IN_LRELU>(p, s);
        case VS_IN_MASK: return launch_ktap_inst<KT, VS_IN_MASK>(p, s);
        default: return launch_ktap_inst<KT, VS_IN_LRELU_MASK>(p, s);
    }
}

// p as for launch_split(cfg 0) on a plain stride-1 conv of the split-f16 arithmetic with C_in a multiple of 16 and ktap_taps(p.KT)
int launch_ktap(const ConvParams &p, hipStream_t s) {
    static_assert(KTAP_LDS >= 4 * 8 * 256 * (int)sizeof(float), "the epilogue's transposition buffers fit the staging buffers");
    if (p.kind != VS_CONV1D || p.Cin % CK != 0 || p.x_bf16 || p.y_bf16 || p.tstep < 1 || (p.KT - 1) * p.tstep > MAX_SPAN || p.lo != p.off0) {
        set_error("launch_ktap: not a plain stride-1 conv of whole 16-channel chunks");
        return VS_EUNSUPPORTED;
    }
    switch (p.KT) {
        case 3: return launch_ktap_kt<3>(p, s);
        case 7: return launch_ktap_kt<7>(p, s);
        case 11: return launch_ktap_kt<11>(p, s);
        default: set_error("launch_ktap: no instance for %d taps", p.KT); return VS_EUNSUPPORTED;
    }
}

}  // namespace vs
