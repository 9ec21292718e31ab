// resblock_pair_split.hip -- the fused residual pair of resblock_pair.hip on the split-bf16 x6 arithmetic of conv_split.hip:
//
//     y = conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 + x   [+ acc] [* scale]            (reference decoder.py:92-101)
//
// On the narrow stages of the generator (32 / 64 channels, 2-4 input chunks) a conv on the bf16 matrix pipe is HBM-bound
// as its own launch (k = 3 at 32 channels: 3.2 GB in 0.70 ms = 4.6 TB/s; tools/conv_stamps.py: the main loop is 36-56 % of a
// workgroup, the rest is the exposed latency of the first stage-in and of the residual read / store).  Fused, a pair moves
// 3 tensor passes instead of 6: x staged once, x again as the residual (what sits in LDS is leaky-relu(x) in bf16 planes), y
// out; the intermediate never leaves the CU.
//
//   phase 1  conv1 over a tile of BN intermediate columns, exactly the main loop of conv_split_kernel (x split into three
//            bf16 planes while it is staged, six cross products per 16 input channels);
//   between  bias is already in the accumulators; leaky-relu, zero outside the sequence (conv2's own zero padding), split
//            into planes and written to LDS as conv2's B operand [plane][channel group of 8][column][8 bf16] -- a lane of the
//            32x32 accumulator tile holds 4 consecutive channels of a column (rows (r&3) + 8(r>>2) + 4(lane>>5)), i.e. half of
//            a 16-byte cell: one ds_write_b64 per plane and 8 rows, the same pattern as the staging writes;
//   phase 2  conv2 straight from that tile: no staging, no barrier; NOUT = BN - halo columns are final outputs;
//   epilogue + x [+ acc] [* scale] through the LDS transpose, all residual loads in flight at once.
//
// Measured and of no use here (tools/pair_bench.py): weight fragments requested three steps ahead through a 4-slot ring with
// counted vmcnt waits, and the next step's first B planes read under the current step's last MFMAs -- both within 1 %: with two
// or three workgroups per CU those latencies are already covered; 32 x 128 tiles (four to five workgroups per CU) were 5-15 % SLOWER
// (more halo columns, twice the weight-fragment traffic per MFMA).  What the phase stamps show instead (tools/pair_stamps.py,
// C=32, k=11): phase 2 runs at the rate of the (power-throttled) matrix pipe, phase 1 takes 1.45x as long for the same MFMAs
// (chunk barriers), and prologue + transform + epilogue are 8 us of a 45 us workgroup.
//
// Tiles: 32 channels: 32 x 256 (four waves side by side), 64 channels: 64 x 128 (2 x 2 waves); both keep LDS at ~54-61 KB
// (two workgroups per CU) and 32 accumulator registers per wave.
#include "conv_common.h"

#include <algorithm>
#include <type_traits>

namespace vs {

constexpr int QHALO = 12;    // pitch slack of the intermediate tile (>= k - 1 for k <= 13)

struct PairSplitParams {
    const float *x;
    long long x_bs;
    const void *ws1, *ws2;            // bf16 plane fragments (pack_split_kernel), Ws[m_tile][tap][chunk][plane][64][8 bf16]
    const float *bias1, *bias2;
    float *y;
    const float *res, *acc;
    long long y_bs, res_bs, acc_bs;
    float scale;
    int B, C, T, K, d1, nchunks;
    int W1;          // staged x columns: BN + (K - 1) * d1
    int PH;          // intermediate columns that are not final outputs: (K - 1) rounded up to 4
    int fast_epi;
    unsigned long long *stamps;   // debug: per-workgroup phase time stamps (NULL in production; tools/pair_stamps.py)
};

__device__ __forceinline__ void pstamp(const PairSplitParams &p, int slot) {
    if (p.stamps && threadIdx.x == 0)
        p.stamps[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 64 + slot] = __builtin_amdgcn_s_memrealtime();
}

// TERMS = 6: split-bf16 x6 (three planes, fp32 class); TERMS = 1: operands rounded to bf16 (one plane) -- VS_MATH_BF16, where a
// narrow conv is even further below the HBM ridge as its own launch.
// PB: x, res, acc and y are bf16-RESIDENT tensors (plain-bf16 arithmetic only, vs_dtype): half the bytes of a kernel that is HBM-bound in
// that arithmetic (3.3-4.7 TB/s algorithmic with fp32 tensors).  Every difference is one `if constexpr (PB)` around a load or a store, so
// that the fp32-tensor instances compile to what they were.
template <int NT_W, int WAVES_M, int WAVES_N, int TERMS, bool PB = false>
__global__ void __launch_bounds__(256, 2) respair_split_kernel(const PairSplitParams p) {
    static_assert(WAVES_M * WAVES_N == 4, "four waves");
    static_assert(TERMS == 6 || TERMS == 1, "split-bf16 x6 or plain bf16");
    static_assert(!PB || TERMS == 1, "bf16-resident tensors go with the plain-bf16 arithmetic");
    constexpr int NPL = (TERMS == 6) ? 3 : 1;
    constexpr int BN = 32 * NT_W * WAVES_N;
    constexpr int WT = BN + QHALO;                 // column pitch of the intermediate tile
    constexpr int CIT = (BN + 64 + 63) / 64;       // (K - 1) * d1 <= 64
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int lhalf = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int NOUT = BN - p.PH;
    const int n0 = blockIdx.x * NOUT;              // first final output of this tile
    const int pad1 = p.d1 * (p.K - 1) / 2, pad2 = (p.K - 1) / 2;
    const int tg0 = n0 - pad2;                     // sequence position of intermediate column 0
    const int xg0 = tg0 - pad1;                    // sequence position of staged x column 0
    const int W = p.W1;
    const int PLSZ = 2 * W * 4;                    // dwords per staged plane: [k-group(2)][column][4 dwords]
    unsigned *const lbuf0 = reinterpret_cast<unsigned *>(smem);
    unsigned *const lbuf1 = lbuf0 + NPL * PLSZ;
    const float *xb = p.x + (long long)b * p.x_bs;
    if constexpr (PB) xb = reinterpret_cast<const float *>(reinterpret_cast<const char *>(p.x) + (long long)b * p.x_bs * 2);
    const int KT = p.K;
    const int nsteps = p.nchunks * KT;

    auto acc_row = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * lhalf; };

    // The residual input of the epilogue is x itself (decoder.py:101 `x = xt + x`): the same lines this workgroup is about to stage.
    // Its float4 reads are issued HERE, next to the staging loads, and held in registers (8 * NIT per lane) through both convs:
    // requested 40 us later they had left L2 and came from memory again (4.0 GB of traffic per launch against 3.2 GB compulsory,
    // profiles/r02_b_pmc_traffic.json), and the epilogue waited for them.
    constexpr int E_CW = 32 * NT_W, E_LPR = E_CW / 4, E_RPI = 64 / E_LPR, E_NIT = 8 / E_RPI;
    const bool fast_tile = p.fast_epi && (n0 + NOUT <= p.T);
    const bool e_live = (wn * E_CW + (lane % E_LPR) * 4) < NOUT;
    float4 r4[4][E_NIT];
    if (fast_tile && e_live && p.res) {
        const float *const resp0 = p.res + (long long)b * p.res_bs;
        const long long goff0 = (long long)(wm * 32 + lane / E_LPR) * p.T + n0 + wn * E_CW + (lane % E_LPR) * 4;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
            for (int it = 0; it < E_NIT; ++it) {
                if constexpr (PB)
                    r4[ps][it] = bf4_to_f4(*reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(p.res) + (long long)b * p.res_bs +
                                                                            goff0 + (long long)(8 * ps + it * E_RPI) * p.T));
                else r4[ps][it] = *reinterpret_cast<const float4 *>(resp0 + goff0 + (long long)(8 * ps + it * E_RPI) * p.T);
            }
    }

    // ------------------------------------------------------------------------------------------- phase 1: conv1(lrelu(x))
    f32x16 acc[NT_W];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = p.bias1[wm * 32 + acc_row(r)];
#pragma unroll
        for (int j = 0; j < NT_W; ++j) acc[j][r] = bv;
    }
    float st[4][CIT];
    const __amdgpu_buffer_rsrc_t xsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, (int)((long long)p.C * p.T * (PB ? 2 : 4)), 0x00020000);
    auto stage_load = [&](int chunk) __attribute__((always_inline)) {
        const int nbase = xg0 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (PB) {
                const int voff = ((chunk * CK + 4 * wave + j) * p.T + nbase) * 2;
#pragma unroll
                for (int i = 0; i < CIT; ++i)
                    st[j][i] = u2f((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xsrc, voff + i * 128, 0, 0) << 16);
            } else {
                const int voff = ((chunk * CK + 4 * wave + j) * p.T + nbase) * 4;
#pragma unroll
                for (int i = 0; i < CIT; ++i)
                    st[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, voff + i * 256, 0, 0));
            }
        }
    };
    const bool time_edge = (xg0 < 0) || (xg0 + W > p.T);
    auto stage_store = [&](unsigned *buf) __attribute__((always_inline)) {
        auto run = [&](auto edge_tag) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            unsigned *const dst0 = buf + ((wave >> 1) * W + lane) * 4 + (wave & 1) * 2;
#pragma unroll
            for (int i = 0; i < CIT; ++i) {
                const int col = lane + 64 * i;
                const int n = xg0 + col;
                const bool okn = (n >= 0) && (n < p.T);
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = st[j][i];
                    if constexpr (EDGE) v[j] = okn ? v[j] : 0.f;
                    v[j] = fmaxf(v[j], 0.1f * v[j]);
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
        if (time_edge) run(std::true_type{});
        else run(std::false_type{});
    };

    // A fragments: NPL 16-byte loads per (chunk, tap) step, the next step's requested at the start of the current one
    u32x4 a0[NPL], a1[NPL];
    const u32x4 *wbase = reinterpret_cast<const u32x4 *>(p.ws1) + (long long)wm * KT * p.nchunks * (NPL * 64) + lane;
    auto load_a = [&](u32x4 (&dst)[NPL], int chunk, int tap) __attribute__((always_inline)) {
        const u32x4 *src = wbase + ((long long)tap * p.nchunks + chunk) * (NPL * 64);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) dst[pl] = src[pl * 64];
    };
    int pc = 0, pt = 0;
    auto advance = [&]() __attribute__((always_inline)) { if (++pt == KT) { pt = 0; ++pc; } };
    load_a(a0, pc, pt); advance();

    pstamp(p, 0);
    stage_load(0);
    stage_store(lbuf0);
    if (p.nchunks > 1) stage_load(1);
    __syncthreads();
    pstamp(p, 1);

    // one (chunk, tap) step: NT_W column tiles x 6 cross products; xs = LDS address (dwords) of this lane's 16-byte cell of the
    // first tile in plane 0, plsz = plane pitch; the planes of tile j+1 are read under the MFMAs of tile j
    auto mma_step = [&](const u32x4 (&acur)[NPL], const unsigned *xs, int plsz) __attribute__((always_inline)) {
        u32x4 bf[NPL], bn[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) bf[pl] = *reinterpret_cast<const u32x4 *>(xs + pl * plsz);
#pragma unroll
        for (int j = 0; j < NT_W; ++j) {
            if (j + 1 < NT_W) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) bn[pl] = *reinterpret_cast<const u32x4 *>(xs + pl * plsz + (j + 1) * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
            auto mm = [&](int ta, int tb) __attribute__((always_inline)) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, acur[ta]), __builtin_bit_cast(bf16x8, bf[tb]),
                                                                 acc[j], 0, 0, 0);
            };
            if constexpr (TERMS == 6) { mm(1, 1); mm(2, 0); mm(0, 2); mm(1, 0); mm(0, 1); }      // smallest terms first
            mm(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) bf[pl] = bn[pl];
        }
    };

    int chunk = 0, tap = 0, s = 0;
    auto step1 = [&](u32x4 (&acur)[NPL], u32x4 (&apre)[NPL]) __attribute__((always_inline)) {
        const unsigned *cur = (chunk & 1) ? lbuf1 : lbuf0;
        load_a(apre, min(pc, p.nchunks - 1), pt); advance();      // (unconditional, clamped: see resblock_f16.hip -- behind a branch hipcc waits vmcnt(0) for it)
        if (tap == 0) {
            if (chunk + 1 < p.nchunks) stage_store((chunk & 1) ? lbuf0 : lbuf1);
            if (chunk + 2 < p.nchunks) stage_load(chunk + 2);
        }
        mma_step(acur, cur + (lhalf * W + wn * (NT_W * 32) + l31 + tap * p.d1) * 4, PLSZ);
        if (++tap == KT) {
            __syncthreads();
            tap = 0;
            ++chunk;
        }
        ++s;
    };
    while (s < nsteps) {
        step1(a0, a1);
        if (s < nsteps) step1(a1, a0);
    }
    // (the barrier after the last tap of the last chunk: every wave is done with the staging buffers)
    pstamp(p, 2);

    // ------------------------------------------------------------------ intermediate tile -> LDS, as conv2's B operand
    const int KG = p.C / 8;                                     // channel groups of the intermediate
    const int TPL = KG * WT * 4;                                // dwords per plane of the tile
    unsigned *const Tb = reinterpret_cast<unsigned *>(smem);    // [plane][group][column][4 dwords]
#pragma unroll
    for (int j = 0; j < NT_W; ++j) {
        const int col = wn * (NT_W * 32) + j * 32 + l31;
        const int gpos = tg0 + col;
        const bool inside = (gpos >= 0) && (gpos < p.T);        // conv2 pads the SEQUENCE with zeros, not the tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = acc[j][4 * g + q];
                v[q] = inside ? fmaxf(t, 0.1f * t) : 0.f;
            }
            unsigned d0[NPL], d1[NPL];
            split_pair<NPL>(v[0], v[1], d0);
            split_pair<NPL>(v[2], v[3], d1);
            unsigned *dst = Tb + ((wm * 4 + g) * WT + col) * 4 + lhalf * 2;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint2 *>(dst + pl * TPL) = make_uint2(d0[pl], d1[pl]);
        }
    }
    // columns BN .. WT-1 feed only discarded outputs, but must be finite numbers
    for (int e = tid; e < NPL * KG * QHALO; e += 256) {
        const int rowi = e / QHALO, c = BN + (e % QHALO);       // rowi = plane * KG + group
        *reinterpret_cast<u32x4 *>(Tb + (rowi * WT + c) * 4) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    pstamp(p, 8);
    // ------------------------------------------------------------------------------------------- phase 2: conv2 from LDS
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = p.bias2[wm * 32 + acc_row(r)];
#pragma unroll
        for (int j = 0; j < NT_W; ++j) acc[j][r] = bv;
    }
    wbase = reinterpret_cast<const u32x4 *>(p.ws2) + (long long)wm * KT * p.nchunks * (NPL * 64) + lane;
    pc = 0; pt = 0; chunk = 0; tap = 0; s = 0;
    load_a(a0, pc, pt); advance();
    auto step2 = [&](u32x4 (&acur)[NPL], u32x4 (&apre)[NPL]) __attribute__((always_inline)) {
        load_a(apre, min(pc, p.nchunks - 1), pt); advance();      // (unconditional, clamped: see resblock_f16.hip -- behind a branch hipcc waits vmcnt(0) for it)
        mma_step(acur, Tb + ((chunk * 2 + lhalf) * WT + wn * (NT_W * 32) + l31 + tap) * 4, TPL);
        if (++tap == KT) { tap = 0; ++chunk; }
        ++s;
    };
    while (s < nsteps) {
        step2(a0, a1);
        if (s < nsteps) step2(a1, a0);
    }
    __syncthreads();                                            // the tile in LDS is consumed: its space becomes the epilogue's
    pstamp(p, 9);

    // ------------------------------------------------------------------------------------------- epilogue: + x [+ acc] [* scale]
    const int tile_row0 = wm * 32;
    const bool has_res = p.res != nullptr, has_acc = p.acc != nullptr;
    float *const yb = p.y + (long long)b * p.y_bs;
    const float *const resp = has_res ? p.res + (long long)b * p.res_bs : nullptr;
    const float *const accp = has_acc ? p.acc + (long long)b * p.acc_bs : nullptr;
    if (fast_tile) {
        constexpr int CW = E_CW, LPR = E_LPR, RPI = E_RPI, NIT = E_NIT;
        float *const Lw = smem + wave * 8 * CW;
        const int lrow = lane / LPR, c4 = (lane % LPR) * 4;
        const int ctile = wn * CW + c4;                          // column within the tile
        const bool live = ctile < NOUT;                          // NOUT % 4 == 0: a float4 is all in or all out
        const long long goff0 = (long long)(tile_row0 + lrow) * p.T + n0 + ctile;
        float4 a4[4][NIT];
        if (live && has_acc) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    if constexpr (PB)
                        a4[ps][it] = bf4_to_f4(*reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(p.acc) + (long long)b * p.acc_bs +
                                                                                goff0 + (long long)(8 * ps + it * RPI) * p.T));
                    else a4[ps][it] = *reinterpret_cast<const float4 *>(accp + goff0 + (long long)(8 * ps + it * RPI) * p.T);
                }
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < NT_W; ++j) Lw[(q + 4 * lhalf) * CW + 32 * j + l31] = acc[j][4 * ps + q];
            if (live) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    float4 v = *reinterpret_cast<const float4 *>(Lw + (it * RPI + lrow) * CW + c4);
                    if (has_res) { v.x += r4[ps][it].x; v.y += r4[ps][it].y; v.z += r4[ps][it].z; v.w += r4[ps][it].w; }
                    if (has_acc) { v.x += a4[ps][it].x; v.y += a4[ps][it].y; v.z += a4[ps][it].z; v.w += a4[ps][it].w; }
                    if (p.scale != 1.f) { v.x *= p.scale; v.y *= p.scale; v.z *= p.scale; v.w *= p.scale; }
                    if constexpr (PB)
                        *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(p.y) + (long long)b * p.y_bs + goff0 +
                                                   (long long)(8 * ps + it * RPI) * p.T) = f4_to_bf4(v);
                    else *reinterpret_cast<float4 *>(yb + goff0 + (long long)(8 * ps + it * RPI) * p.T) = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT_W; ++j) {
            const int ctile = wn * (NT_W * 32) + j * 32 + l31;
            const int n = n0 + ctile;
            const bool okc = (ctile < NOUT) && (n < p.T);
            const int nc = min(n, p.T - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long off = (long long)(tile_row0 + acc_row(r)) * p.T + nc;
                float v = acc[j][r];
                if constexpr (PB) {
                    if (has_res) v += u2f((unsigned)(reinterpret_cast<const unsigned short *>(p.res) + (long long)b * p.res_bs)[off] << 16);
                    if (has_acc) v += u2f((unsigned)(reinterpret_cast<const unsigned short *>(p.acc) + (long long)b * p.acc_bs)[off] << 16);
                    v *= p.scale;
                    if (okc) (reinterpret_cast<unsigned short *>(p.y) + (long long)b * p.y_bs)[off] = (unsigned short)(rne_bf16(v) >> 16);
                } else {
                    if (has_res) v += resp[off];
                    if (has_acc) v += accp[off];
                    v *= p.scale;
                    if (okc) yb[off] = v;
                }
            }
        }
    }
    if (p.stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        pstamp(p, 3);
    }
}

template <int NT_W, int WAVES_M, int WAVES_N, int TERMS, bool PB = false>
static int launch_pair_split_cfg(PairSplitParams p, hipStream_t s) {
    constexpr int BN = 32 * NT_W * WAVES_N, WT = BN + QHALO, NPL = (TERMS == 6) ? 3 : 1;
    auto kern = respair_split_kernel<NT_W, WAVES_M, WAVES_N, TERMS, PB>;
    p.W1 = BN + (p.K - 1) * p.d1;
    const size_t lds = std::max<size_t>({(size_t)2 * NPL * 2 * p.W1 * 16, (size_t)NPL * (p.C / 8) * WT * 16,
                                         (size_t)4 * 8 * (32 * NT_W) * sizeof(float)});
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.T, BN - p.PH), 1, (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    set_last_kernel("respair_split_kernel<%d, %d, %d, %d, %s>", NT_W, WAVES_M, WAVES_N, TERMS, PB ? "true" : "false");     // (as rocprofv3 prints it)
    return VS_OK;
}

}  // namespace vs

using namespace vs;

extern unsigned long long *g_stamp_buf;   // conv_engine.hip (vs_debug_set_stamp_buffer)

// called by vs_respair_forward (resblock_pair.hip) when both convs run on the bf16 matrix pipe (both split-bf16 x6, or both bf16)
int vs_respair_split_launch(const vs_conv *c1, const vs_conv *c2, const vs_conv_io_t *io, int fast_epi, hipStream_t s) {
    PairSplitParams p;
    memset(&p, 0, sizeof(p));
    const int C = c1->c_in;
    const long long dflt = (long long)C * io->T;
    p.x = io->x; p.x_bs = io->x_bs ? io->x_bs : dflt;
    p.ws1 = c1->ws.p; p.bias1 = c1->biasp.as<float>();
    p.ws2 = c2->ws.p; p.bias2 = c2->biasp.as<float>();
    const vs_conv_out_t &o = io->out[0];
    p.y = o.y; p.res = o.res; p.acc = o.acc;
    p.y_bs = o.y_bs ? o.y_bs : dflt; p.res_bs = o.res_bs ? o.res_bs : dflt; p.acc_bs = o.acc_bs ? o.acc_bs : dflt;
    p.scale = (o.scale == 0.f) ? 1.f : o.scale;
    p.B = (int)io->B; p.C = C; p.T = (int)io->T; p.K = c1->k; p.d1 = c1->dil; p.nchunks = c1->nchunks;
    p.PH = ((c1->k - 1) + 3) & ~3;
    p.fast_epi = fast_epi;
    p.stamps = g_stamp_buf;
    if (io->x_dtype == VS_DTYPE_BF16) {        // (vs_respair_forward: x and y types agree, plain-bf16 arithmetic)
        if (C == 32) return launch_pair_split_cfg<2, 1, 4, 1, true>(p, s);
        return launch_pair_split_cfg<2, 2, 2, 1, true>(p, s);
    }
    if (c1->math == VS_MATH_BF16) {
        if (C == 32) return launch_pair_split_cfg<2, 1, 4, 1>(p, s);
        return launch_pair_split_cfg<2, 2, 2, 1>(p, s);
    }
    if (C == 32) return launch_pair_split_cfg<2, 1, 4, 6>(p, s);
    return launch_pair_split_cfg<2, 2, 2, 6>(p, s);
}
