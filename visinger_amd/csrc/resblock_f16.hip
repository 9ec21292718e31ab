// resblock_f16.hip -- a whole MRF residual block of the HiFi-GAN generator in ONE launch, on the split-f16 x3 arithmetic (VS_MATH_SPLIT3):
//
//     for each pair m:   x = conv2_m(lrelu(conv1_m(lrelu(x)) + b1_m)) + b2_m + x          (reference modules/visinger/decoder.py:91-104)
//     y = (x [+ acc]) * scale                                                               (the MRF sum and its 1 / num_kernels: decoder.py:52-56)
//
// With two f16 planes and three cross products a 32- / 64-channel conv is far below the HBM ridge as its own launch, and still below it
// as a fused pair (x in, y out per pair).  Here the residual stream never leaves the CU between the pairs:
//
//   * a workgroup owns BN = 256 columns of the sequence and ALL channels.  The same BN columns go through every conv; each conv eats
//     its receptive half-width off both ends of the range that is still exact, so after the block BN - 2 H columns are final outputs
//     (H = the sum of the pads: 12 at k = 3, 36 at k = 7) and neighbouring tiles overlap by 2 H columns (recomputed, not exchanged);
//   * the residual stream x lives in REGISTERS, fp32, in the accumulator layout of the 32x32 MFMA tile (row = (r&3) + 8(r>>2) + 4(lane>>5),
//     column = lane&31): `x = xt + x` is an add on the accumulators of conv2, no LDS or HBM round trip, full fp32 precision;
//   * the B operand of every conv is one LDS tile [plane(2)][channel group of 8][column + margin][8 f16], written straight from that
//     layout (a lane holds 4 consecutive channels of a column = half a 16-byte cell: one ds_write_b64 per plane) after leaky-relu,
//     zeroing outside the sequence (each conv's own zero padding) and the split x * s = xh + xl.  s is the power of two that puts
//     the tile's largest magnitude below 2^15: one exponent per wave through LDS, merged behind the barrier that also ends the
//     previous conv's reads of the tile;
//   * every conv then is the barrier-free loop of resblock_pair_split.hip's phase 2: weight fragments from L2 one step ahead, each tap a
//     shifted window of the tile, three f16 MFMAs per 16 channels and 32x32 outputs;
//   * epilogue: the exact columns through the LDS transposition, float4 stores, the accumulate input read next to them.
//
// HBM traffic per block: x in, y out (+ acc in) for SIX convs; the 1-launch-per-conv form moves x, residual and y per conv.
#include "conv_common.h"

#include <algorithm>
#include <type_traits>

namespace vs {

__device__ __forceinline__ unsigned max3_u32(unsigned a, unsigned b, unsigned c) {      // one v_max3_u32
    unsigned r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

constexpr int RB_MAXCONV = 6;
constexpr int RB_MAXPAD = 28;      // largest pad of a conv of the chain: (k - 1) * d / 2 = 25 at k = 11, d = 5

struct ResblockParams {
    const float *x;
    long long x_bs;
    const void *ws[RB_MAXCONV];          // f16 plane fragments Ws[m_tile][tap][chunk][plane(2)][64][8 f16] (pack_split_f16_kernel)
    const float *bias[RB_MAXCONV];       // packed bias over rows
    const float *wscale[RB_MAXCONV];     // {s_w, 1 / s_w} of the planes
    int dil[RB_MAXCONV];
    float *y;
    const float *acc;
    long long y_bs, acc_bs;
    float scale;
    int B, C, T, K, nconv, nchunks;
    int H;                               // columns at each end of a tile that are not final outputs (sum of the pads, rounded up to 4)
    int MP;                              // the largest pad of the chain (the kernel instance's margin is the next of 8 / 16 / 28)
    int fast_epi;
    int bf16;                            // VS_MATH_BF16 on bf16-resident tensors (x, y, acc point at bf16 elements; strides in elements)
    unsigned long long *stamps;          // debug: per-workgroup phase stamps (vs_debug_set_stamp_buffer; NULL in production), tools/resblock_stamps.py
};

// [0] start, [1] x loaded, then per conv c: [2 + 4c] input transformed + exponent barrier, [3 + 4c] tile written + barrier, [4 + 4c] MFMA loop done,
// [5 + 4c] scaled out; [2 + 4 nconv] end.  s_memrealtime (100 MHz) in the slots, the shader clock (s_memtime) at [32 + slot].
__device__ __forceinline__ void rb_stamp(const ResblockParams &p, int slot) {
    if (p.stamps && threadIdx.x == 0) {
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.z;
        p.stamps[(size_t)lin * 64 + slot] = __builtin_amdgcn_s_memrealtime();
        p.stamps[(size_t)lin * 64 + 32 + slot] = __builtin_amdgcn_s_memtime();
    }
}

// MP: margin columns of the tile on each side (>= the largest pad of the chain: 8 / 16 / 28 for k = 3 / 7 / 11).  A template constant: with
// a run-time pitch the sixteen (column tile, channel group) cell addresses of the tile writes were hoisted out of the conv loop as
// registers and spilled (112 B of scratch per lane: 0.5 GB of scratch stores per launch in the WRITE_SIZE counter, 27 reloads per conv);
// with a constant pitch they are immediate offsets.
// PLANES = 2: the split-f16 x3 arithmetic (VS_MATH_SPLIT3).  PLANES = 1: operands rounded to bf16, one product (VS_MATH_BF16, BASELINE.json's
// long-form configuration): no scale, half the tile; PB: x, acc and y are bf16-RESIDENT tensors (vs_dtype; only with PLANES = 1).
template <int NT_W, int WAVES_M, int WAVES_N, int MP, int PLANES, bool PB>
__device__ __forceinline__ void resblock_body(const ResblockParams &p) {
    constexpr int NW = WAVES_M * WAVES_N;          // four waves (two workgroups per CU) or eight (one: 128 channels on 256-column tiles)
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    static_assert(PLANES == 2 || (PLANES == 1), "two f16 planes or one bf16 plane");
    static_assert(!PB || PLANES == 1, "bf16-resident tensors go with the plain-bf16 arithmetic");
    constexpr int BN = 32 * NT_W * WAVES_N;
    constexpr int KG = 4 * WAVES_M;                // channel groups of 8
    constexpr int WT = BN + 2 * MP;                // column pitch of the tile
    constexpr int TPL = KG * WT * 4;               // dwords per plane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned *const Tb = reinterpret_cast<unsigned *>(smem);
    int *const smax = reinterpret_cast<int *>(Tb + PLANES * TPL);  // [2][NW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int lhalf = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int NOUT = BN - 2 * p.H;
    const int n0 = blockIdx.x * NOUT;              // first final output of this tile
    const int t0 = n0 - p.H;                       // sequence position of tile column 0
    const int KT = p.K;
    const int nsteps = p.nchunks * KT;
    auto acc_row = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * lhalf; };

    rb_stamp(p, 0);
    // margins of the tile: zeros, once (the waves only ever write their own BN columns)
    for (int e = tid; e < PLANES * KG * 2 * MP; e += 64 * NW) {
        const int rowi = e / (2 * MP), c = e % (2 * MP);
        *reinterpret_cast<u32x4 *>(Tb + (rowi * WT + (c < MP ? c : BN + c)) * 4) = u32x4{0u, 0u, 0u, 0u};
    }

    // ---- the residual stream: x in the accumulator layout, zero outside the sequence
    bool inside[NT_W];
    f32x16 xr[NT_W], acc[NT_W];
    {
        const char *const xbase = reinterpret_cast<const char *>(p.x) + (long long)b * p.x_bs * (PB ? 2 : 4);
        const __amdgpu_buffer_rsrc_t xsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)xbase, 0, (int)((long long)p.C * p.T * (PB ? 2 : 4)), 0x00020000);
#pragma unroll
        for (int j = 0; j < NT_W; ++j) {
            const int n = t0 + wn * (NT_W * 32) + j * 32 + l31;
            inside[j] = (n >= 0) && (n < p.T);
            const int nc = min(max(n, 0), p.T - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (PB)
                    xr[j][r] = u2f((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xsrc, ((wm * 32 + acc_row(r)) * p.T + nc) * 2, 0, 0) << 16);
                else
                    xr[j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, ((wm * 32 + acc_row(r)) * p.T + nc) * 4, 0, 0));
            }
        }
#pragma unroll
        for (int j = 0; j < NT_W; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) xr[j][r] = inside[j] ? xr[j][r] : 0.f;
    }

    // (wave-uniform: every column of this wave's tiles lies inside the sequence)
    bool all_in = true;
#pragma unroll
    for (int j = 0; j < NT_W; ++j) all_in = all_in && (__builtin_amdgcn_ballot_w64(inside[j]) == ~0ull);
    rb_stamp(p, 1);
    // one (chunk, tap) step: NT_W column tiles x 3 cross products; the planes of tile j+1 are read under the MFMAs of tile j
    // ZC: the first step of a conv -- the first product of every column tile takes a ZERO C operand (an inline constant of the instruction) instead of the
    // accumulators being cleared by 64 moves per conv beforehand
    auto mma_step = [&](const u32x4 (&acur)[PLANES], const unsigned *xs, auto zc_c) __attribute__((always_inline)) {
        constexpr bool ZC = decltype(zc_c)::value;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        u32x4 bf[PLANES], bn[PLANES];
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) bf[pl] = *reinterpret_cast<const u32x4 *>(xs + pl * TPL);
#pragma unroll
        for (int j = 0; j < NT_W; ++j) {
            if (j + 1 < NT_W) {
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) bn[pl] = *reinterpret_cast<const u32x4 *>(xs + pl * TPL + (j + 1) * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PLANES == 2) {
                auto mm = [&](int ta, int tb) __attribute__((always_inline)) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, acur[ta]), __builtin_bit_cast(f16x8, bf[tb]), acc[j], 0, 0, 0);
                };
                if constexpr (ZC) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, acur[1]), __builtin_bit_cast(f16x8, bf[0]), zero16, 0, 0, 0);
                else mm(1, 0);
                mm(0, 1); mm(0, 0);        // smallest terms first
            } else {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, acur[0]), __builtin_bit_cast(bf16x8, bf[0]), ZC ? zero16 : acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) bf[pl] = bn[pl];
        }
    };

    for (int c = 0; c < p.nconv; ++c) {
        const bool second = (c & 1) != 0;
        // (an opaque copy of the lane id per conv: the LDS addresses below are a few VALU to rebuild -- hoisted out of this loop as
        //  loop-invariant registers they were spilled and reloaded once per conv)
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
        const int lh = lane_c >> 5, l5 = lane_c & 31;
        // ---- this conv's input in the accumulator registers: lrelu(x) (first conv of a pair) or lrelu(conv1 + b1) (second), zero outside
        unsigned mkey = 0u;
        // (0.1 v for two values per instruction: v_pk_mul_f32 on the register pairs of the accumulator tile)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 slope2 = {0.1f, 0.1f};
        // (round 6, VALU diet -- under the chip's power limit instructions saved come back as clock: a tile wholly inside the sequence, i.e. all but the two
        //  at an item's ends, skips the two selects per pair; the maximum key of a pair is ONE v_max3_u32)
        auto transform = [&](auto in_c) __attribute__((always_inline)) {
            constexpr bool ALL_IN = decltype(in_c)::value;
#pragma unroll
            for (int j = 0; j < NT_W; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v2 = second ? f32x2{acc[j][r], acc[j][r + 1]} : f32x2{xr[j][r], xr[j][r + 1]};
                    const f32x2 t2 = v2 * slope2;
                    float v0 = fmaxf(v2.x, t2.x), v1 = fmaxf(v2.y, t2.y);
                    if constexpr (!ALL_IN) { v0 = inside[j] ? v0 : 0.f; v1 = inside[j] ? v1 : 0.f; }
                    acc[j][r] = v0;
                    acc[j][r + 1] = v1;
                    if constexpr (PLANES == 2) mkey = max3_u32(mkey, (f2u(v0) << 1) + 0x01000000u, (f2u(v1) << 1) + 0x01000000u);      // (f16_maxkey of both)
                }
        };
        if (all_in) transform(std::true_type{});
        else transform(std::false_type{});
        // ---- the tile's scale: largest exponent over the four waves (the barrier also ends every wave's reads of the previous tile)
        int eb = 127;
        float sx = 1.f;
        if constexpr (PLANES == 2) {
            const int ebw = wave_max_u8(f16_key_exponent(mkey));
            int *const slot = smax + (c & 1) * NW;
            if (lane == 0) slot[wave] = ebw;
            __syncthreads();
            eb = max(max(max(slot[0], slot[1]), max(slot[2], slot[3])), F16_EB_MIN);
            if constexpr (NW == 8) eb = max(eb, max(max(slot[4], slot[5]), max(slot[6], slot[7])));
            sx = f16_scale(eb);
        } else {
            __syncthreads();
        }
        rb_stamp(p, 2 + 4 * c);
        // ---- split and write the tile
#pragma unroll
        for (int j = 0; j < NT_W; ++j) {
            const int col = MP + wn * (NT_W * 32) + j * 32 + l5;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned d0[PLANES], d1[PLANES];
                if constexpr (PLANES == 2) {
                    const f32x2 sx2 = {sx, sx};
                    const f32x2 a0 = f32x2{acc[j][4 * g], acc[j][4 * g + 1]} * sx2, a1 = f32x2{acc[j][4 * g + 2], acc[j][4 * g + 3]} * sx2;      // (v_pk_mul_f32)
                    split_pair_h(a0.x, a0.y, d0);
                    split_pair_h(a1.x, a1.y, d1);
                } else {
                    split_pair<1>(acc[j][4 * g], acc[j][4 * g + 1], d0);
                    split_pair<1>(acc[j][4 * g + 2], acc[j][4 * g + 3], d1);
                }
                unsigned *dst = Tb + ((wm * 4 + g) * WT + col) * 4 + lh * 2;
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) *reinterpret_cast<uint2 *>(dst + pl * TPL) = make_uint2(d0[pl], d1[pl]);
            }
        }
        // ---- weight fragments of the first step, then the conv (the accumulators are cleared by the first step's zero C operand)
        const int d = p.dil[c];
        const int pad = d * (KT - 1) / 2;
        const u32x4 *const wbase = reinterpret_cast<const u32x4 *>(p.ws[c]) + (long long)wm * KT * p.nchunks * (PLANES * 64) + lane_c;
        u32x4 a0[PLANES], a1[PLANES];
        // (fragment and tile positions as RUNNING offsets -- one scalar add per step each where (tap * nchunks + chunk) * (PLANES * 64) was a 64-bit multiply
        //  and the tile address a v_mul_lo: the scalar stream of a step is what it costs next to its 4-12 MFMAs)
        const int a_dtap = p.nchunks * (PLANES * 64), a_dwrap = (PLANES * 64) - (KT - 1) * a_dtap;
        const int a_last = ((KT - 1) * p.nchunks + (p.nchunks - 1)) * (PLANES * 64);
        int aoff = 0;
        auto load_a = [&](u32x4 (&dst)[PLANES]) __attribute__((always_inline)) {
            const u32x4 *src = wbase + min(aoff, a_last);          // (clamped: the last step's extra request, never used, stays inside this conv's fragments)
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) dst[pl] = src[pl * 64];
        };
        int pt = 0, tap = 0, s = 0;
        auto advance = [&]() __attribute__((always_inline)) { if (++pt == KT) { pt = 0; aoff += a_dwrap; } else aoff += a_dtap; };
        load_a(a0); advance();
        __syncthreads();
        rb_stamp(p, 3 + 4 * c);
        const unsigned *const xlane = Tb + (lh * WT + MP + wn * (NT_W * 32) + l5 - pad) * 4;
        const int x_dtap = d * 4, x_dwrap = 2 * WT * 4 - (KT - 1) * d * 4;
        int xoff = 0;
        auto step = [&](u32x4 (&acur)[PLANES], u32x4 (&apre)[PLANES], auto zc_c) __attribute__((always_inline)) {
            // (UNCONDITIONAL: behind `if (s + 1 < nsteps)` hipcc's wait in front of this step's first MFMA was vmcnt(0) -- the scoreboard merge of the two
            //  paths -- which also waited for the fragments requested on the line above: one exposed L2 round trip per step pair in every conv of
            //  every fused block.  Round 4, found in the ISA.)
            load_a(apre); advance();
            mma_step(acur, xlane + xoff, zc_c);
            if (++tap == KT) { tap = 0; xoff += x_dwrap; } else xoff += x_dtap;
            ++s;
        };
        step(a0, a1, std::true_type{});                      // (nsteps >= 2: two chunks at 32 channels, three taps at least)
        step(a1, a0, std::false_type{});
        while (s < nsteps) {
            step(a0, a1, std::false_type{});
            if (s < nsteps) step(a1, a0, std::false_type{});
        }
        rb_stamp(p, 4 + 4 * c);
        // ---- scale out, bias in; the second conv of a pair adds the residual stream
        const float isx = (PLANES == 2) ? f16_inv_scale(eb) : 1.f, isw = (PLANES == 2) ? p.wscale[c][1] : 1.f;      // (one after the other: conv_split_body.inc)
        const float *const bias = p.bias[c] + wm * 32;
        // (two values per instruction: v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 on the register pairs (r, r + 1) of an accumulator tile)
        const f32x2 isx2 = {isx, isx}, isw2 = {isw, isw};
        // (`second` is wave-uniform: a branch around each form -- as selects inside one loop the update cost two v_cndmask per value)
        if (second) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 bv2 = {bias[(r & 3) + 8 * (r >> 2) + 4 * lh], bias[((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * lh]};
#pragma unroll
                for (int j = 0; j < NT_W; ++j) {
                    const f32x2 a = {acc[j][r], acc[j][r + 1]};
                    const f32x2 v = __builtin_elementwise_fma(a * isx2, isw2, bv2);
                    f32x2 x2 = {xr[j][r], xr[j][r + 1]};
                    x2 += v;
                    xr[j][r] = x2.x; xr[j][r + 1] = x2.y;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 bv2 = {bias[(r & 3) + 8 * (r >> 2) + 4 * lh], bias[((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * lh]};
#pragma unroll
                for (int j = 0; j < NT_W; ++j) {
                    const f32x2 a = {acc[j][r], acc[j][r + 1]};
                    const f32x2 v = __builtin_elementwise_fma(a * isx2, isw2, bv2);
                    acc[j][r] = v.x; acc[j][r + 1] = v.y;
                }
            }
        }
        rb_stamp(p, 5 + 4 * c);
    }
    __syncthreads();                         // the tile is consumed: its space becomes the epilogue's

    // ---- epilogue: y = (x [+ acc]) * scale on the exact columns [H, BN - H) of the tile
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int tile_row0 = wm * 32;
    const bool has_acc = p.acc != nullptr;
    float *const yb = p.y + (long long)b * p.y_bs;
    const float *const accp = has_acc ? p.acc + (long long)b * p.acc_bs : nullptr;
    unsigned short *const yh = reinterpret_cast<unsigned short *>(p.y) + (long long)b * p.y_bs;                 // PB: bf16 elements
    const unsigned short *const acch = reinterpret_cast<const unsigned short *>(p.acc) + (long long)b * p.acc_bs;
    if (p.fast_epi && (n0 + NOUT <= p.T)) {
        constexpr int CW = 32 * NT_W, LPR = CW / 4, RPI = 64 / LPR, NIT = 8 / RPI;
        float *const Lw = smem + wave * 8 * CW;
        const int lrow = lane_e / LPR, c4 = (lane_e % LPR) * 4;
        const int ctile = wn * CW + c4;                          // column within the tile
        const bool live = (ctile >= p.H) && (ctile < BN - p.H);  // H % 4 == 0: a float4 is all in or all out
        const long long goff0 = (long long)(tile_row0 + lrow) * p.T + t0 + ctile;
        float4 a4[4][NIT];
        if (live && has_acc) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    if constexpr (PB) a4[ps][it] = bf4_to_f4(*reinterpret_cast<const uint2 *>(acch + goff0 + (long long)(8 * ps + it * RPI) * p.T));
                    else a4[ps][it] = *reinterpret_cast<const float4 *>(accp + goff0 + (long long)(8 * ps + it * RPI) * p.T);
                }
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < NT_W; ++j) Lw[(q + 4 * (lane_e >> 5)) * CW + 32 * j + (lane_e & 31)] = xr[j][4 * ps + q];
            if (live) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    float4 v = *reinterpret_cast<const float4 *>(Lw + (it * RPI + lrow) * CW + c4);
                    if (has_acc) { v.x += a4[ps][it].x; v.y += a4[ps][it].y; v.z += a4[ps][it].z; v.w += a4[ps][it].w; }
                    if (p.scale != 1.f) { v.x *= p.scale; v.y *= p.scale; v.z *= p.scale; v.w *= p.scale; }
                    if constexpr (PB) *reinterpret_cast<uint2 *>(yh + goff0 + (long long)(8 * ps + it * RPI) * p.T) = f4_to_bf4(v);
                    else *reinterpret_cast<float4 *>(yb + goff0 + (long long)(8 * ps + it * RPI) * p.T) = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT_W; ++j) {
            const int ctile = wn * (NT_W * 32) + j * 32 + l31;
            const int n = t0 + ctile;
            const bool okc = (ctile >= p.H) && (ctile < BN - p.H) && (n < p.T);      // (n >= 0 on exact columns: t0 + H = n0 >= 0)
            const int nc = min(max(n, 0), p.T - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long off = (long long)(tile_row0 + acc_row(r)) * p.T + nc;
                float v = xr[j][r];
                if constexpr (PB) {
                    if (has_acc) v += u2f((unsigned)acch[off] << 16);
                    v *= p.scale;
                    if (okc) yh[off] = (unsigned short)(rne_bf16(v) >> 16);
                } else {
                    if (has_acc) v += accp[off];
                    v *= p.scale;
                    if (okc) yb[off] = v;
                }
            }
        }
    }
    if (p.stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        rb_stamp(p, 2 + 4 * p.nconv);
    }
}

template <int NT_W, int WAVES_M, int WAVES_N, int MP>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, 2) resblock_f16_kernel(const ResblockParams p) {
    resblock_body<NT_W, WAVES_M, WAVES_N, MP, 2, false>(p);
}

// VS_MATH_BF16 on bf16-resident tensors (BASELINE.json configs[4])
template <int NT_W, int WAVES_M, int WAVES_N, int MP>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, 2) resblock_bf16_kernel(const ResblockParams p) {
    resblock_body<NT_W, WAVES_M, WAVES_N, MP, 1, true>(p);
}

static int margin_of(int maxpad) { return maxpad <= 8 ? 8 : (maxpad <= 16 ? 16 : RB_MAXPAD); }

template <int NT_W, int WAVES_M, int WAVES_N>
static size_t resblock_lds(int MP, int planes = 2) {
    constexpr int BN = 32 * NT_W * WAVES_N, KG = 4 * WAVES_M;
    return std::max<size_t>((size_t)planes * KG * (BN + 2 * MP) * 16 + 64, (size_t)WAVES_M * WAVES_N * 8 * (32 * NT_W) * sizeof(float));
}

template <int NT_W, int WAVES_M, int WAVES_N, int MP, bool BF>
static int launch_resblock_mp(const ResblockParams &p, hipStream_t s) {
    constexpr int BN = 32 * NT_W * WAVES_N;
    auto kern = BF ? resblock_bf16_kernel<NT_W, WAVES_M, WAVES_N, MP> : resblock_f16_kernel<NT_W, WAVES_M, WAVES_N, MP>;
    const size_t lds = resblock_lds<NT_W, WAVES_M, WAVES_N>(MP, BF ? 1 : 2);
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.T, BN - 2 * p.H), 1, (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_M * WAVES_N), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    set_last_kernel(BF ? "resblock_bf16_kernel<%d, %d, %d, %d>" : "resblock_f16_kernel<%d, %d, %d, %d>", NT_W, WAVES_M, WAVES_N, MP);
    return VS_OK;
}

template <int NT_W, int WAVES_M, int WAVES_N, bool BF>
static int launch_resblock_margin(const ResblockParams &p, hipStream_t s) {
    switch (margin_of(p.MP)) {
        case 8: return launch_resblock_mp<NT_W, WAVES_M, WAVES_N, 8, BF>(p, s);
        case 16: return launch_resblock_mp<NT_W, WAVES_M, WAVES_N, 16, BF>(p, s);
        default: return launch_resblock_mp<NT_W, WAVES_M, WAVES_N, RB_MAXPAD, BF>(p, s);
    }
}

template <int NT_W, int WAVES_M, int WAVES_N>
static int launch_resblock_cfg(const ResblockParams &p, hipStream_t s) {
    return p.bf16 ? launch_resblock_margin<NT_W, WAVES_M, WAVES_N, true>(p, s) : launch_resblock_margin<NT_W, WAVES_M, WAVES_N, false>(p, s);
}

}  // namespace vs

using namespace vs;

extern unsigned long long *g_stamp_buf;   // conv_engine.hip (vs_debug_set_stamp_buffer)

extern "C" {

int vs_resblock_supported(vs_conv_t *const *convs, int nconv) {
    if (!convs || nconv < 2 || nconv > RB_MAXCONV || (nconv & 1)) return 0;
    const vs_conv *c0 = convs[0];
    if (!c0) return 0;
    const int C = c0->c_in, k = c0->k;
    if (!(C == 32 || C == 64 || C == 128) || !(k & 1) || k < 3 || k > 11) return 0;
    if (c0->math != VS_MATH_SPLIT3 && c0->math != VS_MATH_BF16) return 0;
    const int planes = c0->math == VS_MATH_BF16 ? 1 : 2;
    int H = 0, MP = 0;
    for (int i = 0; i < nconv; ++i) {
        const vs_conv *c = convs[i];
        if (!c || c->kind != VS_CONV1D || c->c_in != C || c->c_out != C || c->k != k || c->flags != 0 || c->math != c0->math) return 0;
        if (c->pad != c->dil * (k - 1) / 2 || c->pad > RB_MAXPAD) return 0;
        H += c->pad;
        MP = std::max(MP, c->pad);
    }
    H = (H + 3) & ~3;
    if (C == 128)       // 256-column tiles on eight waves, one workgroup per CU (160 KB of LDS hold the all-channel tile up to a margin of 28)
        return 256 - 2 * H >= 64 && resblock_lds<4, 4, 2>(margin_of(MP), planes) <= 160 * 1024;
    return 256 - 2 * H >= 64;
}

int vs_resblock_forward(vs_conv_t *const *convs, int nconv, const vs_conv_io_t *io, void *stream) {
    VS_REQUIRE(convs && io, "vs_resblock_forward: NULL argument");
    VS_REQUIRE(vs_resblock_supported(convs, nconv), "vs_resblock_forward: unsupported chain of convs (32 / 64 / 128 channels, one odd k <= 11, VS_MATH_SPLIT3, "
                                                   "'same' padding, an even number of convs <= 6)");
    VS_REQUIRE(io->x && io->out[0].y && io->B > 0 && io->B <= 65535 && io->T > 0, "vs_resblock_forward: bad io");
    const bool bf = convs[0]->math == VS_MATH_BF16;
    VS_REQUIRE(io->x_dtype == (bf ? VS_DTYPE_BF16 : VS_DTYPE_F32) && io->y_dtype == io->x_dtype,
               "vs_resblock_forward: fp32 tensors with VS_MATH_SPLIT3, bf16-resident tensors with VS_MATH_BF16");
    VS_REQUIRE(io->in_act == VS_IN_LRELU && !io->mask && !io->bias_b && !io->split_row && io->out[0].mode == VS_OUT_LINEAR &&
                   io->out[0].out_act == VS_OUT_NONE && !io->out[0].out_mask && !io->out[0].res,
               "vs_resblock_forward: only the unmasked leaky-relu residual form is fused (the residual input is x itself)");
    ResblockParams p;
    memset(&p, 0, sizeof(p));
    const vs_conv *c0 = convs[0];
    const int C = c0->c_in;
    p.bf16 = bf;
    VS_REQUIRE((long long)C * io->T * 4 < (1ll << 31), "vs_resblock_forward: item exceeds the 2 GiB buffer-descriptor range");
    const long long dflt = (long long)C * io->T;
    p.x = static_cast<const float *>(io->x); p.x_bs = io->x_bs ? io->x_bs : dflt;
    int H = 0;
    for (int i = 0; i < nconv; ++i) {
        const vs_conv *c = convs[i];
        VS_REQUIRE(c->weights_set, "vs_resblock_forward: weights not set");
        p.ws[i] = c->ws.p; p.bias[i] = c->biasp.as<float>(); p.wscale[i] = c->wsc.as<float>(); p.dil[i] = c->dil;
        H += c->pad;
        p.MP = std::max(p.MP, c->pad);
    }
    p.H = (H + 3) & ~3;
    const vs_conv_out_t &o = io->out[0];
    p.y = static_cast<float *>(o.y); p.acc = static_cast<const float *>(o.acc);
    p.y_bs = o.y_bs ? o.y_bs : dflt; p.acc_bs = o.acc_bs ? o.acc_bs : dflt;
    p.scale = (o.scale == 0.f) ? 1.f : o.scale;
    p.B = (int)io->B; p.C = C; p.T = (int)io->T; p.K = c0->k; p.nconv = nconv; p.nchunks = c0->nchunks;
    p.stamps = g_stamp_buf;
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    auto al8 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 7u) == 0; };
    p.fast_epi = (io->T % 4 == 0) && (p.y_bs % 4 == 0) && (!p.acc || p.acc_bs % 4 == 0) &&
                 (bf ? al8(p.y) && (!p.acc || al8(p.acc)) : al16(p.y) && (!p.acc || al16(p.acc)));
    hipStream_t s = as_stream(stream);
    if (C == 32) {
        // 512-column tiles halve the halo and the weight-fragment traffic per output; with a small halo (k = 3: 12 columns) the 256-column
        // tile's shorter critical path per workgroup wins (tools/resblock_bench.py: k = 3 1.79 vs 1.90 ms, k = 7 3.47 vs 2.87, k = 11 4.87 vs 4.55)
        if (opt(OPT_RB_TILE256) || p.H <= 12 || 512 - 2 * p.H > io->T + 256) return launch_resblock_cfg<2, 1, 4>(p, s);
        return launch_resblock_cfg<4, 1, 4>(p, s);
    }
    if (C == 128) {
        // a single pair: 128-column tiles on four waves, two workgroups per CU (4.4 against 4.6 ms at k = 3), while the tile fits half the LDS;
        // longer chains: 256 columns on eight waves, one workgroup per CU -- half the halo per output (whole k = 3 block 4.0 against 4.25 ms)
        if (nconv == 2 && 128 - 2 * p.H >= 64 && 2 * (resblock_lds<4, 4, 1>(margin_of(p.MP), bf ? 1 : 2) + 64) <= 160 * 1024)
            return launch_resblock_cfg<4, 4, 1>(p, s);
        return launch_resblock_cfg<4, 4, 2>(p, s);
    }
    // 64 channels: 256-column tiles on four waves, two workgroups per CU -- unless the chain's halo is wide (H >= 20 columns at each end: the k = 7 pair at dilation 5,
    // the k = 11 pairs at dilations 3 and 5, whole k >= 5 blocks): then 512-column tiles on eight waves, one workgroup per CU, recompute half as many halo columns per
    // output.  Under the chip's power limit the launch time follows the EXECUTED matrix work (tools/resblock_stamps.py: the same MFMA cycle counts at 1.25-1.55 GHz):
    // round 6, B = 32 x T = 131072, launch spans 3.23 -> 2.98 ms (k = 11, d = 3), 3.43 -> 3.03 (k = 11, d = 5), 2.30 -> 2.06 (k = 7, d = 5); with H <= 12 the narrow
    // tile's two independent workgroups per CU win or tie (profiles/r06_resblock_wide64_ab.txt).
#ifndef VS_RB_NO_WIDE64      // (tools/build_variant.py: the A/B library without this branch)
    if (p.H >= 20 && (long long)ceil_div(p.T, 512 - 2 * p.H) * p.B >= 512 && !opt(OPT_RB_TILE256) &&      // (and the launch still fills the chip twice over)
        resblock_lds<4, 2, 4>(margin_of(p.MP), bf ? 1 : 2) <= 160 * 1024)
        return launch_resblock_cfg<4, 2, 4>(p, s);
#endif
    return launch_resblock_cfg<4, 2, 2>(p, s);
}

}  // extern "C"
