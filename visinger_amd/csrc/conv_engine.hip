// conv_engine.hip -- the implicit-GEMM 1-D convolution engine of the VISinger hot path on gfx950.
//
// One kernel family serves every nn.Conv1d / nn.ConvTranspose1d of the path (reference sites listed in
// include/visinger_hip.h).  It is an implicit GEMM on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (bit-identical to a k-ordered fmaf chain, 64 FLOP/clk/SIMD = the fp32 peak of the chip):
//
//     D[m, n] = sum_{tap} sum_{ci}  Wp[tap][m][ci] * f(x[b, ci, n + off(tap)])
//
//   * A operand = weights, pre-packed ONCE on the device into fragment order Wp[m_tile][tap][chunk][quad][64 lanes][4]
//     (lane l holds W[m_tile*32 + (l&31)][2*ci_pair + (l>>5)] for the four ci_pairs of the quad) so a wave fetches four
//     fragments with one coalesced 1-KiB load (served by L2/L1: the weights of a conv are <= 3 MB and shared by every
//     workgroup);
//   * B operand = activations, staged ONCE per (ci-chunk, time tile + halo) into LDS as Xs[ci][n]; every tap reads
//     a shifted window of the same LDS rows (conflict-free ds_read_b32: 32 consecutive dwords per half-wave), so
//     HBM/L2 sees each activation once per M-block no matter how many taps the conv has;
//   * transposed convs run as a polyphase conv over "virtual rows" m = phase*C_out + co with per-tile tap
//     ranges (no multiplies by structural zeros) and an interleaving store;
//   * the input transform (leaky-relu / mask) is applied while staging; bias, conditioning bias, residual add,
//     accumulate, scale, tanh, mask, WaveNet gate, res/skip split and the affine-coupling update (+ log-det
//     wave-shuffle reduction) are fused into the epilogue, so no elementwise kernel ever touches HBM.
//
// Tile: 4 waves (256 threads), each wave owns MT_W x NT_W accumulator tiles of 32x32 (16 VGPRs each); two
// workgroups per CU (2 waves/SIMD) hide the staging of one behind the MFMAs of the other.
#include "conv_common.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <unordered_set>
#include <new>
#include <type_traits>
#include <vector>

namespace vs {

thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- dispatch switches: see vs_internal.h (enum Opt).  Read from the environment ONCE, when the library is loaded.
struct OptEntry { const char *name; long long dflt; long long value; };
static OptEntry g_opts[OPT_COUNT] = {
    {"VS_CONV_MATH", -1, -1}, {"VS_NO_SMALL_CONV", 0, 0}, {"VS_NO_FAST_EPI", 0, 0}, {"VS_WINO_FORCE", 0, 0}, {"VS_NO_WINO", 0, 0},
    {"VS_NO_WINO_K7", 0, 0}, {"VS_WINO_DBG", 0, 0}, 
    {"VS_NO_SMALL_GRID", 0, 0}, {"VS_SMALL_GRID_T6", 512, 512}, {"VS_CONV_CFG", -1, -1}, {"VS_SPLIT_DBG", 0, 0}, {"VS_TRACE", 0, 0},
    {"VS_NO_BF16_ATTN", 0, 0}, {"VS_NO_SPLIT_ATTN", 0, 0}, {"VS_NO_WGRAD_SPLIT", 0, 0}, {"VS_RB_TILE256", 0, 0},
    {"VS_NO_ATTN_KVPACK", 0, 0}, {"VS_NO_TR_EPI", 0, 0}, {"VS_NO_KTAP", 0, 0}, {"VS_NO_ATTN_DMA", 0, 0},
    {"VS_ATTN_DMA_ONE_WAVE", 0, 0}, {"VS_ATTN_SPLIT6", 0, 0}, {"VS_NO_T1_CONV", 0, 0},
};
static const bool g_opts_loaded = [] {
    for (OptEntry &e : g_opts) {
        const char *v = getenv(e.name);
        if (v && *v) {
            char *end = nullptr;
            const long long n = strtoll(v, &end, 10);
            e.value = (end != v) ? n : 1;          // (a non-numeric value counts as "set")
        }
    }
    return true;
}();
long long opt(Opt o) { return g_opts[o].value; }

thread_local char g_last_kernel[160] = "";
void set_last_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
    va_end(ap);
}


template <int MT_W, int NT_W, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, 2) conv_mfma_kernel(const ConvParams p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int BN = 32 * NT_W * WAVES_N;
    constexpr int MAXW = BN + MAX_SPAN;
    constexpr int RPW = CK / NW;                 // LDS rows staged per wave
    constexpr int CIT = (MAXW + 63) / 64;        // column iterations per row
    static_assert(CK % NW == 0, "CK must be a multiple of the wave count");
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
    float *const buf0 = smem;
    float *const buf1 = smem + CK * W;
    const float *const xb = p.x + (long long)b * p.x_bs;
    const float *const maskb = p.mask ? p.mask + (long long)b * p.Tin : nullptr;

    // ---- tap range of this wave (polyphase transposed conv: skip the structurally-zero taps) ----
    int tap_b = 0, tap_e = p.KT;
    if (p.kind == VS_CONV_TRANSPOSE1D && (p.c_out & 31) == 0) {
        // every 32-row tile has one phase; MT_W tiles of a wave may differ -> union
        int lo_t = p.KT, hi_t = 0;
#pragma unroll
        for (int i = 0; i < MT_W; ++i) {
            const int phase = ((mt0 + i) * 32) / p.c_out;
            if (phase < p.up) {
                // delta range with 0 <= delta*up + phase + pad < K
                const int num_lo = -(phase + p.uppad);                    // delta >= ceil(num_lo / up)
                const int dlo = (num_lo >= 0) ? (num_lo + p.up - 1) / p.up : -((-num_lo) / p.up);
                const int num_hi = p.upK - 1 - phase - p.uppad;           // delta <= floor(num_hi / up)
                const int dhi = (num_hi >= 0) ? num_hi / p.up : -((-num_hi + p.up - 1) / p.up);
                lo_t = min(lo_t, dlo - p.dmin);
                hi_t = max(hi_t, dhi - p.dmin + 1);
            }
        }
        tap_b = max(0, lo_t);
        tap_e = min(p.KT, hi_t);
        // a wave whose tiles are all padding (M tiles rounded up to the workgroup) has no tap of its own, but it still owns
        // LDS rows of every chunk and a seat at every chunk barrier: give it one tap (its packed weights are zeros)
        if (tap_e <= tap_b) { tap_b = 0; tap_e = 1; }
    }

    // The accumulators start from the bias (+ the per-item conditioning bias) of their row instead of zero: the same 128
    // v_mov, and the epilogue loses one VALU add per element -- a VALU instruction of a wave in its epilogue waits for a gap
    // in the co-resident workgroup's MFMA stream (~one MFMA slot each, tools/conv_stamps.py), so epilogue time is
    // proportional to its VALU count.
    const float *const bbias = p.bias_b ? p.bias_b + (long long)b * p.bias_b_bs : nullptr;
    f32x16 acc[MT_W][NT_W];
#pragma unroll
    for (int i = 0; i < MT_W; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rt = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int m = (mt0 + i) * 32 + rt;                 // virtual row (biasp is zero-padded to whole tiles)
            float bv = p.biasp[m];
            if (bbias) {
                int row;
                if constexpr (MT_W == 2) {                     // paired rows: tile 2q -> c, tile 2q+1 -> Hh + c
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

    float st[RPW][CIT];
    float mk[CIT];
    const int in_act = p.in_act;

    // Staging loads are UNCONDITIONAL buffer loads (one 32-bit offset VGPR per LDS row, the column iterations are
    // immediate offsets; out-of-range bytes of the item's [Cin, Tin] slab read as 0 by the descriptor's bounds check)
    // and the zero-fill of the halo is applied when the registers are written to LDS: a predicated load makes hipcc
    // branch around every load and drain vmcnt(0) per element, and 64-bit per-load addresses cost 2 VGPRs each.
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
        for (int j = 0; j < RPW; ++j) {
            const int ci = min(chunk * CK + wave + NW * j, p.Cin - 1);
            const int voff = (ci * p.Tin + nbase) * 4;
#pragma unroll
            for (int i = 0; i < CIT; ++i)
                st[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, voff + i * 256, 0, 0));
        }
    };
    // The zero-fill of the halo only exists in the first / last time tiles and in a partial last channel chunk, and the input
    // transform is one of four: both are wave-uniform, so the per-element work of the common case (interior tile, leaky-relu)
    // is mul + max + ds_write instead of compare / select / compare / select / multiply-select / predicated write.
    const bool time_edge = (n0 + p.lo < 0) || (n0 + p.lo + W > p.Tin);
    auto stage_store = [&](float *buf, int chunk) __attribute__((always_inline)) {
        auto run = [&](auto edge_tag, auto act_tag) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int i = 0; i < CIT; ++i) {
                const int col = lane + 64 * i;
                const int n = n0 + p.lo + col;
                const bool okn = (n >= 0) && (n < p.Tin);
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    float v = st[j][i];
                    if constexpr (EDGE) v = (okn && (chunk * CK + wave + NW * j < p.Cin)) ? v : 0.f;
                    if constexpr (ACT == VS_IN_LRELU || ACT == VS_IN_LRELU_MASK) v = fmaxf(v, 0.1f * v);
                    if constexpr (ACT >= VS_IN_MASK) v *= mk[i];
                    if (64 * (i + 1) <= BN || col < W) buf[(wave + NW * j) * W + col] = v;      // W >= BN
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
    // The K loop is flattened into steps s = (chunk, tap); a step is 8 groups (one per input-channel pair) of
    // MT_W*NT_W MFMAs.  Software pipeline (hipcc builds none by itself):
    //   * A fragments (packed weights, L2-resident): 3-deep register ring rotated by NAME (the step loop is unrolled
    //     by 3; a rotating copy would read the slot being prefetched); the slot of step s+2 is requested at step s;
    //   * activations: chunk c+2 is requested at the first tap of chunk c, right AFTER that step's A prefetch (vmcnt
    //     retires in issue order: whatever is issued behind them completes behind them) and stays in registers for a
    //     whole chunk; chunk c+1 is written to the other LDS buffer at the same point;
    //   * B fragments of group g+1 are read from LDS under the MFMAs of group g, pinned above them by sched_barrier
    //     (left alone hipcc sinks each ds_read next to its consumer: ds_read2 -> lgkmcnt(0) -> 2 MFMA).
    // Tried and measured worse (tools/conv_stamps.py, per-step shader cycles): inline-asm ring loads with hand-counted
    // vmcnt (same), staging pieces spread over the gaps between groups (+8 %: a lone wave hides only ~64 cycles of
    // non-MFMA issue per gap).
    const int ntaps = tap_e - tap_b;
    const int nsteps = p.nchunks * ntaps;
    const float *wbase[MT_W];
#pragma unroll
    for (int i = 0; i < MT_W; ++i) wbase[i] = p.wp + (long long)(mt0 + i) * p.KT * p.CP * 64 + lane * 4;
    float a0[MT_W][CK / 2], a1[MT_W][CK / 2], a2[MT_W][CK / 2];
    // (the 8 fragments of a step are stored as two lane-interleaved quads: two 16-byte loads, 1 KiB per wave-instruction)
    auto load_a = [&](float (&dst)[MT_W][CK / 2], int chunk, int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT_W; ++i) {
            const float *src = wbase[i] + ((long long)tap * p.nchunks + chunk) * (CK / 2) * 64;
#pragma unroll
            for (int qd = 0; qd < 2; ++qd) {
                const float4 t = *reinterpret_cast<const float4 *>(src + qd * 256);
                dst[i][qd * 4 + 0] = t.x; dst[i][qd * 4 + 1] = t.y; dst[i][qd * 4 + 2] = t.z; dst[i][qd * 4 + 3] = t.w;
            }
        }
    };
    // (chunk, tap) of the step two ahead of the current one
    int pc = 0, pt = tap_b;
    auto advance = [&]() __attribute__((always_inline)) { if (++pt == tap_e) { pt = tap_b; ++pc; } };
    if (nsteps > 0) { load_a(a0, pc, pt); advance(); }
    if (nsteps > 1) { load_a(a1, pc, pt); advance(); }

    stamp(p, 0);
    stage_load(0);
    stage_store(buf0, 0);
    if (p.nchunks > 1) stage_load(1);      // lives in registers during chunk 0
    __syncthreads();
    stamp(p, 1);

    int chunk = 0, tap = tap_b, s = 0;
    auto step = [&](float (&acur)[MT_W][CK / 2], float (&apre)[MT_W][CK / 2]) __attribute__((always_inline)) {
        const float *cur = (chunk & 1) ? buf1 : buf0;
        const bool more = (chunk + 1 < p.nchunks);
        if (s + 2 < nsteps) { load_a(apre, pc, pt); advance(); }
        if (tap == tap_b) {
            // chunk c+1 (requested one whole chunk ago) goes to the other LDS buffer, which every wave finished reading
            // at the barrier that ended chunk c-1; then chunk c+2 is requested: a full chunk of MFMAs hides its latency
            // whatever the tap count (with the request at the first tap and the store at the last one, a 1-tap conv
            // waited for HBM in every step)
            if (more) stage_store((chunk & 1) ? buf0 : buf1, chunk + 1);
            if (chunk + 2 < p.nchunks) stage_load(chunk + 2);
        }

        const float *xs = cur + lhalf * W + wn * (NT_W * 32) + l31 - p.lo + (p.off0 + tap * p.tstep);
        float bf[NT_W], bn[NT_W];
#pragma unroll
        for (int j = 0; j < NT_W; ++j) bf[j] = xs[j * 32];
#pragma unroll
        for (int cp = 0; cp < CK / 2; ++cp) {
            if (cp + 1 < CK / 2) {
#pragma unroll
                for (int j = 0; j < NT_W; ++j) bn[j] = xs[(cp + 1) * 2 * W + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT_W; ++i)
#pragma unroll
                for (int j = 0; j < NT_W; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[i][cp], bf[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NT_W; ++j) bf[j] = bn[j];
        }
        if (++tap == tap_e) {
            __syncthreads();
            tap = tap_b;
            ++chunk;
        }
        ++s;
    };
    while (s < nsteps) {
        step(a0, a2);
        if (s < nsteps) step(a1, a0);
        if (s < nsteps) step(a2, a1);
    }

    stamp(p, 2);
#include "conv_epilogue.inc"
    if (p.stamps) {
        __builtin_amdgcn_s_waitcnt(0);   // all stores acknowledged
        stamp(p, 3);
    }
}

// One half-step of conv_wino_kernel: 4 input-channel pairs of one tap group, F(2,3): four products per pair column.
// x values of channel pair cp + 1 are read from LDS under the MFMAs of pair cp; the V's of a pair are all formed before its
// first MFMA (a VALU result consumed by the very next MFMA costs wait states).  xa/xb: LDS addresses of (x0, x2) and (x1, x3)
// of the wave's two pair tiles -- with dilation 1 the even and odd columns of the window are staged in separate halves of
// each LDS row, so both are unit-stride across lanes (a stride-2 read is a 2-way bank conflict); otherwise xb = xa + DIL.
// (A partial last tap group can run as F(2,2) / direct form into the same accumulators -- k=7: 10 instead of 12 MFMAs -- but
// any RUNTIME control flow with MFMAs on the accumulators in both arms makes hipcc spill them: the generic kernel zero-pads
// the taps, and the k = 7 instances (TG = 3) unroll the six half-steps of a chunk as straight-line code.)
template <int DIL, int TAPS>
__device__ __forceinline__ void wino_half_step(f32x16 (&acc)[4][2], const float (&a)[16], const float *xa0, const float *xb0,
                                               const float *xa1, const float *xb1, int RP) {
    // TAPS == 3: F(2,3), four products.  TAPS == 1 (the last group of k = 7 in the k-specialised instances, where the step
    // sequence of a chunk is straight-line code -- no branch carries the accumulators): the direct form, y[t] += w x0 into M0
    // and y[t+d] += w x1 into M3 (U3 is packed negated): two products instead of the zero-padded four.
    constexpr int S2 = (DIL == 1) ? 1 : 2 * DIL;      // distance x0 -> x2 (and x1 -> x3) in LDS words
    const float *xa[2] = {xa0, xa1}, *xb[2] = {xb0, xb1};
    float xc[2][4], xn[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        xc[j][0] = xa[j][0];
        xc[j][1] = xb[j][0];
        if constexpr (TAPS >= 2) xc[j][2] = xa[j][S2];
        if constexpr (TAPS == 3) xc[j][3] = xb[j][S2];
    }
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) {
        if (cp + 1 < 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                xn[j][0] = xa[j][(cp + 1) * 2 * RP];
                xn[j][1] = xb[j][(cp + 1) * 2 * RP];
                if constexpr (TAPS >= 2) xn[j][2] = xa[j][(cp + 1) * 2 * RP + S2];
                if constexpr (TAPS == 3) xn[j][3] = xb[j][(cp + 1) * 2 * RP + S2];
            }
        }
        float v[4][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (TAPS == 3) {
                v[0][j] = xc[j][0] - xc[j][2];
                v[1][j] = xc[j][1] + xc[j][2];
                v[2][j] = xc[j][2] - xc[j][1];
                v[3][j] = xc[j][1] - xc[j][3];
            } else if constexpr (TAPS == 2) {      // F(2,2): (x0 - x1) w0 -> M0, x1 (w0 + w1) -> M1, (x1 - x2) w1 -> M3
                v[0][j] = xc[j][0] - xc[j][1];
                v[1][j] = xc[j][1];
                v[2][j] = 0.f;
                v[3][j] = xc[j][1] - xc[j][2];
            } else {
                v[0][j] = xc[j][0];
                v[1][j] = 0.f;
                v[2][j] = 0.f;
                v[3][j] = xc[j][1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            if constexpr (TAPS == 1) {
                if (xi == 1 || xi == 2) continue;          // (compile-time after unrolling)
            }
            if constexpr (TAPS == 2) {
                if (xi == 2) continue;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[xi][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp * 4 + xi], v[xi][j], acc[xi][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < TAPS + 1; ++q) xc[j][q] = xn[j][q];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Minimal-filtering (Winograd F(2,3)) variant of the engine for the stride-1 "same" convolutions with >= 3 taps
// (the HiFi-GAN resblock convs k in {3,7,11}, dilation {1,3,5}, decoder.py:72-87 -- 87 % of the synthesis FLOPs -- and the
// FFN k=9 convs, rel_transformer.py:332-333).  The taps are cut into groups of three; for one group and one pair of
// outputs (y[t], y[t+d]) the six products of the direct form become four:
//     V0 = x0 - x2, V1 = x1 + x2, V2 = x2 - x1, V3 = x1 - x3                 (x_j = x[t + (3g + j) d - pad])
//     U0 = w0, U1 = (w0 + w1 + w2)/2, U2 = (w0 - w1 + w2)/2, U3 = w2          (packed once, pack_wino_kernel)
//     M_xi += U_xi . V_xi  (contraction over input channels: the MFMA)        y[t] = M0+M1+M2, y[t+d] = M1-M2-M3
// so the matrix pipe does 4/6 of the direct work (k=9: 12/18; k=11: 16/22 and k=7: 12/14 with the zero-padded last group).  The
// transforms use the points {0, 1, -1, inf} only: coefficients 1 and 1/2, error growth ~2x an fp32 dot product.
// Same skeleton as conv_mfma_kernel: activations staged once per 16-channel chunk in LDS (the V's are formed per
// wave from four shifted LDS reads: 1 ds_read + 1 VALU per MFMA), weights as fragments from L2 through a 3-deep
// register ring, LDS-transposed vector epilogue.  A wave owns 32 rows x PW "pair columns" (c -> outputs t(c), t(c)+d
// with t(c) = (c / d) * 2d + c % d, a contiguous run of 2*PW outputs when d divides PW) x 4 xi = 8 accumulator tiles.
template <int DIL, int WAVES_M, int WAVES_N, int TG, int TT>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, 2) conv_wino_kernel(const ConvParams p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int PW = (64 / DIL) * DIL;         // valid pair columns per wave (of 64)
    constexpr int NBW = 2 * PW;                  // outputs per wave
    constexpr int BN = NBW * WAVES_N;
    constexpr int MAXW = BN + MAX_SPAN;
    constexpr int RPW = CK / NW;
    constexpr int CIT = (MAXW + 63) / 64;
    constexpr int CWP = NBW + 8;                 // row pitch of the epilogue transposition buffer (+ dump columns)
    static_assert(CK % NW == 0, "CK must be a multiple of the wave count");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M;
    const int wn = wave / WAVES_M;
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * BN;
    const int mt0 = blockIdx.y * WAVES_M + wm;
    const int W = p.W;
    const int G = p.KT;                          // tap groups
    // LDS row: dilation 1 -> even columns of the window at [0, Wh), odd columns at [H, H + Wh), H = 16 mod 32 (the staging
    // writes of a half-wave then cover all 32 banks); otherwise the window as it is.
    const int Wh = (W + 1) >> 1;
    const int H = ((Wh + 15) & ~31) + 16;
    const int RP = (DIL == 1) ? H + Wh : W;
    float *const buf0 = smem;
    float *const buf1 = smem + CK * RP;
    const float *const xb = p.x + (long long)b * p.x_bs;
    const float *const maskb = p.mask ? p.mask + (long long)b * p.Tin : nullptr;
    const int lhalf = lane >> 5;
    const int l31 = lane & 31;

    // M0 starts from the row's bias and M3 from its negative (y[t] = M0+M1+M2, y[t+d] = M1-M2-M3): see conv_mfma_kernel
    const float *const bbias = p.bias_b ? p.bias_b + (long long)b * p.bias_b_bs : nullptr;
    f32x16 acc[4][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = mt0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        float bv = p.biasp[row];
        if (bbias) bv += bbias[min(row, p.M - 1)];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[0][j][r] = bv;
            acc[1][j][r] = 0.f;
            acc[2][j][r] = 0.f;
            acc[3][j][r] = -bv;
        }
    }

    // pair column -> first output of the pair, relative to the wave's first output
    int posr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = j * 32 + l31;
        const int t = (c / DIL) * (2 * DIL) + (c % DIL);
        posr[j] = (c < PW) ? t : 0;          // idle columns read a valid LDS address
    }

    float st[RPW][CIT];
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
        for (int j = 0; j < RPW; ++j) {
            const int ci = min(chunk * CK + wave + NW * j, p.Cin - 1);
            const int voff = (ci * p.Tin + nbase) * 4;
#pragma unroll
            for (int i = 0; i < CIT; ++i)
                st[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, voff + i * 256, 0, 0));
        }
    };
    const bool time_edge = (n0 + p.lo < 0) || (n0 + p.lo + W > p.Tin);
    auto stage_store = [&](float *buf, int chunk) __attribute__((always_inline)) {
        auto run = [&](auto edge_tag, auto act_tag) __attribute__((always_inline)) {   // (see conv_mfma_kernel)
            constexpr bool EDGE = decltype(edge_tag)::value;
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int i = 0; i < CIT; ++i) {
                const int col = lane + 64 * i;
                const int n = n0 + p.lo + col;
                const bool okn = (n >= 0) && (n < p.Tin);
                const int idx = (DIL == 1) ? ((col & 1) ? H : 0) + (col >> 1) : col;
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    float v = st[j][i];
                    if constexpr (EDGE) v = (okn && (chunk * CK + wave + NW * j < p.Cin)) ? v : 0.f;
                    if constexpr (ACT == VS_IN_LRELU || ACT == VS_IN_LRELU_MASK) v = fmaxf(v, 0.1f * v);
                    if constexpr (ACT >= VS_IN_MASK) v *= mk[i];
                    if (64 * (i + 1) <= BN || col < W) buf[(wave + NW * j) * RP + idx] = v;      // W >= BN
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

    // ------------------------------------------------------------------------------------------------ main loop
    // step s = ((chunk * G + g) * 2 + half): 4 channel pairs x 4 xi x 2 pair tiles = 32 MFMAs on 16 weight fragments,
    // which are contiguous in the packed array (one base pointer, immediate offsets).
    const int nsteps = p.nchunks * G * 2;
    // (the four xi fragments of a channel pair are interleaved per lane: one 16-byte load each, 1 KiB per wave-instruction)
    const float *const wbase = p.wp + (long long)mt0 * nsteps * (16 * 64) + lane * 4;
    float a0[16], a1[16], a2[16];
    auto load_a = [&](float (&dst)[16], int step_) __attribute__((always_inline)) {
        const float *src = wbase + (long long)step_ * (16 * 64);
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
            const float4 t = *reinterpret_cast<const float4 *>(src + cp * 256);
            dst[cp * 4 + 0] = t.x; dst[cp * 4 + 1] = t.y; dst[cp * 4 + 2] = t.z; dst[cp * 4 + 3] = t.w;
        }
    };
    // ring depth: 3 slots (fragments requested two half-steps ahead), 2 for the widest workgroup shape (its 36 staging
    // registers do not leave room for the third slot)
    constexpr int RING = (WAVES_N == 4 || TG == 4) ? 2 : 3;
    if (nsteps > 0) load_a(a0, 0);
    if (RING == 3 && nsteps > 1) load_a(a1, 1);

    stamp(p, 0);
    stage_load(0);
    stage_store(buf0, 0);
    if (p.nchunks > 1) stage_load(1);
    __syncthreads();
    stamp(p, 1);

    int chunk = 0, g = 0, half = 0, s = 0;
    auto step = [&](auto taps_tag, float (&acur)[16], float (&apre)[16]) __attribute__((always_inline)) {
        constexpr int TAPS = decltype(taps_tag)::value;
        const float *cur = (chunk & 1) ? buf1 : buf0;
        if (s + (RING - 1) < nsteps && !((p.dbg & 1) && s > 3)) load_a(apre, s + (RING - 1));
        if (g == 0 && half == 0) {
            if (chunk + 1 < p.nchunks && !(p.dbg & 2)) stage_store((chunk & 1) ? buf0 : buf1, chunk + 1);
            if (chunk + 2 < p.nchunks && !(p.dbg & 2)) stage_load(chunk + 2);
        }
        const float *xq[4];
        {
            const float *rowp = cur + (half * 8 + lhalf) * RP;
            const int t3 = 3 * g;
            if constexpr (DIL == 1) {
                const int ga = (t3 & 1) ? H + (t3 >> 1) : (t3 >> 1);             // x0, x2
                const int gb = (t3 & 1) ? ((t3 + 1) >> 1) : H + (t3 >> 1);       // x1, x3
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = wn * PW + (posr[j] >> 1);
                    xq[2 * j] = rowp + e + ga;
                    xq[2 * j + 1] = rowp + e + gb;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    xq[2 * j] = rowp + wn * NBW + posr[j] + t3 * DIL;
                    xq[2 * j + 1] = xq[2 * j] + DIL;
                }
            }
        }
        wino_half_step<DIL, TAPS>(acc, acur, xq[0], xq[1], xq[2], xq[3], RP);
        ++s;
        if (++half == 2) {
            half = 0;
            if (++g == G) {
                __syncthreads();
                g = 0;
                ++chunk;
            }
        }
    };
    constexpr std::integral_constant<int, 3> F3{};
    if constexpr (TG == 3) {
        // three tap groups -> six half-steps per chunk, a multiple of the ring period: straight-line code.  k = 7 (TT = 1): the
        // last two half-steps in the direct form on M0 / M3; k = 9 (TT = 3): all six F(2,3).
        static_assert(RING == 3, "the k-specialised instances use the 3-slot ring");
        constexpr std::integral_constant<int, TT> TL{};
        for (int c = 0; c < p.nchunks; ++c) {
            step(F3, a0, a2); step(F3, a1, a0); step(F3, a2, a1); step(F3, a0, a2);
            step(TL, a1, a0); step(TL, a2, a1);
        }
    } else if constexpr (TG == 4) {
        // k = 11: groups (3, 3, 3, 2 taps) -> eight half-steps per chunk on the 2-slot ring, the last two as F(2,2)
        constexpr std::integral_constant<int, TT> TL{};
        for (int c = 0; c < p.nchunks; ++c) {
            step(F3, a0, a1); step(F3, a1, a0); step(F3, a0, a1); step(F3, a1, a0); step(F3, a0, a1); step(F3, a1, a0);
            step(TL, a0, a1); step(TL, a1, a0);
        }
    } else if constexpr (RING == 3) {
        while (s < nsteps) {
            step(F3, a0, a2);
            if (s < nsteps) step(F3, a1, a0);
            if (s < nsteps) step(F3, a2, a1);
        }
    } else {
        while (s < nsteps) {       // nsteps is even
            step(F3, a0, a1);
            step(F3, a1, a0);
        }
    }

    stamp(p, 2);
    // ------------------------------------------------------------------------------------------------- epilogue
    // output transform in place: acc[0] <- y[t(c)], acc[3] <- y[t(c) + d]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m1 = acc[1][j][r], m2 = acc[2][j][r];
            acc[0][j][r] = acc[0][j][r] + m1 + m2;
            acc[3][j][r] = m1 - m2 - acc[3][j][r];
        }

    const OutSpec o = p.out[0];
    const int tile_row0 = mt0 * 32;
    const bool has_res = o.res != nullptr, has_acc = o.acc != nullptr;
    const bool use_mask = (o.out_mask != 0);
    float *const yb = o.y + (long long)b * o.y_bs;
    const float *const resp = has_res ? o.res + (long long)b * o.res_bs : nullptr;
    const float *const accp = has_acc ? o.acc + (long long)b * o.acc_bs : nullptr;
    const int nw = n0 + wn * NBW;                // first output of this wave
    // pair column -> output position, recomputed from an opaque copy of the lane id (see below: keeps it out of the prologue)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    int posw[2], pose[2];
    bool cvalid[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = j * 32 + (lane_e & 31);
        const int t = (c / DIL) * (2 * DIL) + (c % DIL);
        cvalid[j] = c < PW;
        posw[j] = cvalid[j] ? t : NBW;       // idle columns write to the dump columns of the transposition buffer
        pose[j] = cvalid[j] ? t : 0;
    }

    if (p.fast_epi && (n0 + BN <= p.N) && (tile_row0 + 32 <= p.M)) {
        constexpr int VEC = (DIL == 1) ? 4 : 2;
        constexpr int LPR = NBW / VEC;           // lanes per row
        constexpr int RPI = 64 / LPR;            // rows per wave-instruction
        constexpr int NIT = 8 / RPI;
        typedef float vecf __attribute__((ext_vector_type(VEC)));
        float *const Lw = smem + wave * 8 * CWP;
        // Everything below depends only on the lane and the launch parameters, so hipcc computes the 64-bit row addresses of
        // the epilogue BEFORE the main loop and spills them across it (17-48 dwords per lane = 1 KB of scratch traffic per
        // dword and workgroup, written and read back: +15..50 % of a workgroup's HBM traffic).  An opaque copy of the lane id
        // taken after the loop keeps that arithmetic here.
        const int lrow = lane_e / LPR;
        const int cv = (lane_e % LPR) * VEC;
        const bool active = lane_e < LPR * RPI;
        const int colg = nw + cv;
        vecf mv;
#pragma unroll
        for (int e = 0; e < VEC; ++e) mv[e] = 1.f;
        if (use_mask && active) mv = *reinterpret_cast<const vecf *>(maskb + colg);
        // Residual / accumulate reads: every 8-row pass needs one HBM round trip (2-3 us with the chip in its epilogues), and
        // four of them in sequence were 11 us of a 50 us workgroup at k=3 (tools/conv_stamps.py) -- so the reads of ALL
        // passes are issued before the first use (64 registers, free now that the fragment ring and the staging registers
        // are dead); with an accumulate input as well, two passes at a time.  Loads of a batch precede its stores and every
        // lane stores exactly the elements it loaded, so y may alias res / acc.
        // SIMPLE = no accumulate input, no scale, no activation, no mask (the inner resblock convs: 5 of 6 launches): the
        // per-element work is LDS read + residual add + store with no branch in it (every runtime `if` inside these unrolled
        // loops is a scalar branch per element group, and the epilogue's instruction stream is what it costs).
        auto run = [&](auto pb_tag, auto simple_tag, auto res_tag) __attribute__((always_inline)) {
            constexpr int PB = decltype(pb_tag)::value;
            constexpr bool SIMPLE = decltype(simple_tag)::value;
            constexpr bool RES = decltype(res_tag)::value;
#pragma unroll
            for (int pb = 0; pb < 4 / PB; ++pb) {
                vecf r4[PB][NIT], a4[PB][NIT];
#pragma unroll
                for (int u = 0; u < PB; ++u)
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const long long goff = (long long)(tile_row0 + 8 * (pb * PB + u) + it * RPI + lrow) * p.Tout + colg;
                        if (active) {
                            if constexpr (RES) r4[u][it] = *reinterpret_cast<const vecf *>(resp + goff);
                            if constexpr (!SIMPLE) {
                                if (has_acc) a4[u][it] = *reinterpret_cast<const vecf *>(accp + goff);
                            }
                        }
                    }
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    const int ps = pb * PB + u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float y0 = acc[0][j][4 * ps + q], y1 = acc[3][j][4 * ps + q];
                            if constexpr (DIL == 1) {      // the pair is adjacent: one 8-byte write, unit stride across lanes
                                *reinterpret_cast<float2 *>(Lw + (q + 4 * lhalf) * CWP + posw[j]) = make_float2(y0, y1);
                            } else {
                                Lw[(q + 4 * lhalf) * CWP + posw[j]] = y0;
                                Lw[(q + 4 * lhalf) * CWP + posw[j] + DIL] = y1;
                            }
                        }
                    }
                    if (active) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) {
                            const long long goff = (long long)(tile_row0 + 8 * ps + it * RPI + lrow) * p.Tout + colg;
                            vecf v = *reinterpret_cast<const vecf *>(Lw + (it * RPI + lrow) * CWP + cv);
                            if constexpr (RES) v += r4[u][it];
                            if constexpr (!SIMPLE) {
                                if (has_acc) v += a4[u][it];
                                v *= o.scale;
                                if (o.out_act != VS_OUT_NONE) {
#pragma unroll
                                    for (int e = 0; e < VEC; ++e) {
                                        if (o.out_act == VS_OUT_TANH) v[e] = tanh_fast(v[e]);
                                        else if (o.out_act == VS_OUT_RELU) v[e] = fmaxf(v[e], 0.f);
                                    }
                                }
                                v *= mv;
                            }
                            *reinterpret_cast<vecf *>(yb + goff) = v;
                        }
                    }
                }
            }
        };
        const bool simple = !has_acc && o.scale == 1.f && o.out_act == VS_OUT_NONE && !use_mask;
        if (simple) {
            if (has_res) run(std::integral_constant<int, 2>{}, std::true_type{}, std::true_type{});
            else run(std::integral_constant<int, 2>{}, std::true_type{}, std::false_type{});
        } else if (has_res) {
            run(std::integral_constant<int, 2>{}, std::false_type{}, std::true_type{});
        } else {
            run(std::integral_constant<int, 2>{}, std::false_type{}, std::false_type{});
        }
    } else {
        // edge workgroups (ragged last time tile): element-wise, predicated stores, clamped loads
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int col = nw + pose[j] + hh * DIL;
                const bool okc = cvalid[j] && (col < p.N);
                const int colc = min(col, p.Tout - 1);
                const float mval = use_mask ? maskb[colc] : 1.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = tile_row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane_e >> 5);
                    const int rowc = min(row, p.M - 1);
                    const long long off = (long long)rowc * p.Tout + colc;
                    float v = (hh == 0 ? acc[0][j][r] : acc[3][j][r]);
                    if (has_res) v += resp[off];
                    if (has_acc) v += accp[off];
                    v *= o.scale;
                    if (o.out_act == VS_OUT_TANH) v = tanh_fast(v);
                    else if (o.out_act == VS_OUT_RELU) v = fmaxf(v, 0.f);
                    v *= mval;
                    if (okc && row < p.M) yb[off] = v;
                }
            }
        }
    }
    if (p.stamps) {
        stamp(p, 6);                     // epilogue issued
        __builtin_amdgcn_s_waitcnt(0);   // all stores acknowledged
        stamp(p, 3);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight-norm row scales and packing

// scale[r] = g[r] / ||v[r, :]||_2 ; one wave per row
__global__ void rownorm_scale_kernel(const float *__restrict__ v, const float *__restrict__ g, float *__restrict__ scale,
                                     long long rows, long long cols) {
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *vr = v + row * cols;
    float ss = 0.f;
    for (long long c = lane; c < cols; c += 64) ss += vr[c] * vr[c];
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) ss += __shfl_xor(ss, s);
    if (lane == 0) scale[row] = g[row] / sqrtf(ss);
}

__global__ void weightnorm_apply_kernel(const float *__restrict__ v, const float *__restrict__ scale, float *__restrict__ w,
                                        long long rows, long long cols) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * cols) w[i] = v[i] * scale[i / cols];
}

struct PackParams {
    const float *w;      // source weight (or weight_v)
    const float *scale;  // per-dim0 scale (weight norm) or null
    const float *bias;   // source bias or null
    float *wp;
    float *biasp;
    int kind, c_in, c_out, k, up, pad, dmin, KT, CP, MT_alloc, Hh;
    unsigned flags;
    int wino_tail1;      // pack_wino_kernel: the single tap of the last group in the direct form (k = 7 specialised instances)
    unsigned *maxbits, *maxbits_clear;   // split-f16 arithmetic: largest |w| (bits) of this pack -> *maxbits; the other slot is zeroed for the next pack
};

// bias of virtual row m (rows past the real ones: 0)
__device__ __forceinline__ float packed_bias(const PackParams &q, int m) {
    int row = -1;
    if (q.kind == VS_CONV1D) {
        if (m < q.c_out) row = (q.flags & VS_CONV_FLIP_OUT) ? q.c_out - 1 - m : m;
    } else if (q.kind == VS_CONV1D_PAIRED) {
        const int pair = m >> 6, which = (m >> 5) & 1, c = pair * 32 + (m & 31);
        if (c < q.Hh) row = which * q.Hh + ((q.flags & VS_CONV_FLIP_OUT) ? q.Hh - 1 - c : c);
    } else {
        const int phase = m / q.c_out;
        if (phase < q.up) row = m - phase * q.c_out;
    }
    return (row >= 0 && q.bias) ? q.bias[row] : 0.f;
}

// The value of fragment element Wp[m_tile][tap][chunk][quad(2)][64 lanes][4]: lane l of quad qd holds channel pairs
// chunk*8 + qd*4 + (0..3), i.e. row m_tile*32 + (l & 31), (logical) input channel 2 * pair + (l >> 5)
__device__ __forceinline__ float packed_weight(const PackParams &q, int mt, int tap, int chunk, int quad, int lane, int sub) {
    const int cp = chunk * (CK / 2) + quad * 4 + sub;
    const int m = mt * 32 + (lane & 31);
    int ci = cp * 2 + (lane >> 5);
    float val = 0.f;
    if (ci < q.c_in) {
        if (q.flags & VS_CONV_FLIP_IN) ci = q.c_in - 1 - ci;
        if (q.kind == VS_CONV1D) {
            if (m < q.c_out) {
                const int row = (q.flags & VS_CONV_FLIP_OUT) ? q.c_out - 1 - m : m;
                val = (q.flags & VS_CONV_ADJOINT) ? q.w[((long long)ci * q.c_out + row) * q.k + (q.k - 1 - tap)]
                                                  : q.w[((long long)row * q.c_in + ci) * q.k + tap];
                if (q.scale) val *= q.scale[row];
            }
        } else if (q.kind == VS_CONV1D_PAIRED) {
            const int pair = mt >> 1, which = mt & 1, c = pair * 32 + (lane & 31);
            if (c < q.Hh) {
                const int row = which * q.Hh + ((q.flags & VS_CONV_FLIP_OUT) ? q.Hh - 1 - c : c);
                val = q.w[((long long)row * q.c_in + ci) * q.k + tap];
                if (q.scale) val *= q.scale[row];
            }
        } else {  // transposed: w is [c_in, c_out, k]; virtual row m = phase*c_out + co; tap <-> delta = dmin + tap
            const int phase = m / q.c_out;
            if (phase < q.up) {
                const int co = m - phase * q.c_out;
                const int kk = (q.dmin + tap) * q.up + phase + q.pad;
                if (kk >= 0 && kk < q.k) {
                    val = q.w[((long long)ci * q.c_out + co) * q.k + kk];
                    if (q.scale) val *= q.scale[ci];
                }
            }
        }
    }
    return val;
}

// blk of nblk: this block's place among the blocks that pack q (the whole grid of the one- and two-handle launches, a slice of it in the batch launch)
__device__ __forceinline__ void pack_conv_body(const PackParams &q, unsigned blk, unsigned nblk) {
    const long long total = (long long)q.MT_alloc * q.KT * q.CP * 64;
    const long long work = max(total, (long long)q.MT_alloc * 32);
    if (blk == 0 && threadIdx.x == 0 && q.maxbits_clear) *q.maxbits_clear = 0u;
    unsigned m = 0u;
    // grid-stride: at most 1024 blocks however large the weight (the largest |w| below is ONE atomic per block: tens of thousands of
    // same-address atomics -- the discriminators' 1024 x 1024 x 5 convs -- serialised into most of a millisecond per pack)
    for (long long e = (long long)blk * blockDim.x + threadIdx.x; e < work; e += (long long)nblk * blockDim.x) {
        if (e < (long long)q.MT_alloc * 32) q.biasp[e] = packed_bias(q, (int)e);   // bias over virtual rows
        if (e < total) {
            const int sub = (int)(e & 3);
            const int lane = (int)((e >> 2) & 63);
            const int quad = (int)((e >> 8) & 1);
            long long t = e >> 9;
            const int nchunks = q.CP / (CK / 2);
            const int chunk = (int)(t % nchunks);
            t /= nchunks;
            const int tap = (int)(t % q.KT);
            const int mt = (int)(t / q.KT);
            const float v = packed_weight(q, mt, tap, chunk, quad, lane, sub);
            q.wp[e] = v;
            m = f16_maxkey(m, v);      // (finite magnitudes only: a NaN / Inf weight must not set the scale)
        }
    }
    if (q.maxbits) {       // (uniform over the launch) the conv's largest weight, for the scale of the f16 planes
        __shared__ unsigned red[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(q.maxbits, max(max(red[0], red[1]), max(red[2], red[3])));
    }
}
__global__ void pack_conv_kernel(const PackParams q) { pack_conv_body(q, blockIdx.x, gridDim.x); }
// two handles fed from the same weight (a conv and the ADJOINT handle of its grad-input: vs_conv_set_weights_pair) in ONE launch:
// blockIdx.y selects the handle.  A training step packs every weight for both; two launches per handle were 1 068 of its 6 175.
__global__ void pack_conv_pair_kernel(const PackParams q0, const PackParams q1) {
    if (blockIdx.y == 0) pack_conv_body(q0, blockIdx.x, gridDim.x);
    else pack_conv_body(q1, blockIdx.x, gridDim.x);
}
// ANY number of handles in one launch (vs_conv_set_weights_batch: every conv of a network at the top of a training pass): the jobs and the first
// block of each (blk0[n] = the grid) come from a table in device memory; a block finds its job by bisection.
__global__ void pack_conv_multi_kernel(const PackParams *__restrict__ jobs, const unsigned *__restrict__ blk0, int n) {
    int lo = 0, hi = n - 1;
    const unsigned b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blk0[mid] <= b) lo = mid;
        else hi = mid - 1;
    }
    const PackParams q = jobs[lo];
    pack_conv_body(q, b - blk0[lo], blk0[lo + 1] - blk0[lo]);
}

// The same pack for the bf16-pipe engine in ONE launch: thread (cell = (m_tile, tap, chunk), lane) produces the eight values of its bf16
// fragment (pack_split_kernel's mapping: row lane & 31, channels chunk*16 + 8*(lane >> 5) + j), writes them to the fp32 fragment buffer
// Wp (kept: vs_conv_set_math and the lazy F(2,3) transforms are built from it) and, split into planes, to Ws.  A training step re-packs
// every conv's weight twice (forward and grad-input handles): two launches per pack were 1 400 of its 7 000.
__global__ void pack_conv_split_kernel(const PackParams q, void *ws, int npl) {
    const int nchunks = q.CP / (CK / 2);
    const long long total = (long long)q.MT_alloc * q.KT * nchunks * 64;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (long long)q.MT_alloc * 32) q.biasp[e] = packed_bias(q, (int)e);
    if (e >= total) return;
    const int lane = (int)(e & 63);
    const long long cell = e >> 6;
    const int chunk = (int)(cell % nchunks);
    const int tap = (int)((cell / nchunks) % q.KT);
    const int mt = (int)(cell / ((long long)nchunks * q.KT));
    const int row = lane & 31, kg = lane >> 5;
    float *const wpc = q.wp + cell * 512;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cl = 8 * kg + j, cp = cl >> 1, par = cl & 1;
        v[j] = packed_weight(q, mt, tap, chunk, cp >> 2, row + 32 * par, cp & 3);
        wpc[(cp >> 2) * 256 + (row + 32 * par) * 4 + (cp & 3)] = v[j];
    }
    u32x4 *dst = reinterpret_cast<u32x4 *>(ws) + cell * npl * 64 + lane;
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

// Winograd F(2,3) weight transform + fragment packing for conv_wino_kernel:
//   Up[m_tile][chunk][group][half][ci_pair(4)][64 lanes][xi(4)], lane l <-> (row m_tile*32 + (l&31), ci = chunk*16 + half*8 +
//   2*ci_pair + (l>>5)); taps beyond k are zeros (the last group of k = 7 / 11 is partial).
__global__ void pack_wino_kernel(const PackParams q, int G, int nchunks) {
    const long long total = (long long)q.MT_alloc * nchunks * G * 2 * 16 * 64;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int xi = (int)(e & 3);
    const int lane = (int)((e >> 2) & 63);
    const int cp4 = (int)((e >> 8) & 3);
    long long t = e >> 10;
    const int half = (int)(t & 1);
    t >>= 1;
    const int g = (int)(t % G);
    t /= G;
    const int chunk = (int)(t % nchunks);
    const int mt = (int)(t / nchunks);
    const int ci = chunk * CK + half * 8 + cp4 * 2 + (lane >> 5);
    // built from the fp32 fragment buffer Wp (q.w), which already holds flips / weight-norm scale / adjoint and zeros in the padding:
    // the transform is made by the first launch that needs it (the bf16-pipe engine and the training path never do)
    auto wv = [&](int tap) -> float {
        const int cp = ci >> 1;
        return q.w[((((long long)mt * q.KT + tap) * nchunks + (cp >> 3)) * 2 + ((cp & 7) >> 2)) * 256 + ((ci & 1) * 32 + (lane & 31)) * 4 + (cp & 3)];
    };
    float val = 0.f;
    {
        const int k0 = 3 * g;
        const float w0 = (k0 < q.k) ? wv(k0) : 0.f;
        const float w1 = (k0 + 1 < q.k) ? wv(k0 + 1) : 0.f;
        const float w2 = (k0 + 2 < q.k) ? wv(k0 + 2) : 0.f;
        if (q.wino_tail1 && k0 + 1 == q.k) val = (xi == 0) ? w0 : (xi == 3) ? -w0 : 0.f;       // single last tap: direct form
        else if (q.wino_tail1 && k0 + 2 == q.k) val = (xi == 0) ? w0 : (xi == 1) ? (w0 + w1) : (xi == 3) ? w1 : 0.f;   // two: F(2,2)
        else val = (xi == 0) ? w0 : (xi == 1) ? 0.5f * (w0 + w1 + w2) : (xi == 2) ? 0.5f * (w0 - w1 + w2) : w2;
    }
    q.wp[e] = val;
}


// ---------------------------------------------------------------------------------------------------------------
// Convs with <= 4 output channels (HiFi-GAN conv_post 32->1 k7 + tanh, decoder.py:34,55-57; PitchPredictor.linear
// 192->2, predictor.py:14): a 32-row MFMA tile would be 97 % padding (measured 1.7 ms for conv_post at 2 TFLOP/s),
// and the op is a pure HBM stream of its input (1.07 GB for conv_post), so it runs on the VALU: one thread per
// 4 consecutive output frames, weights through the scalar cache, x re-read across taps through L1.
struct SmallParams {
    const float *x;
    long long x_bs;
    const float *w;      // effective weights [c_out][c_in][k]
    const float *bias;   // [c_out] or null
    const float *bias_b;
    long long bias_b_bs;
    const float *mask;   // [B, T]
    float *y;
    long long y_bs;
    int B, Cin, Cout, Tin, Tout, K, dil, pad, in_act, out_act, out_mask;
    float scale;
    int x_bf16;          // x holds bf16 elements (bf16-resident activations: the generator's last stage in BASELINE config 5); x_bs in elements
};

// ---------------------------------------------------------------------------------------------------------------
// 1 x 1 convs over ONE frame per item (the conditioning vectors: WN.cond_layer 256 -> 2 * hidden * n_layers on the speaker embedding,
// modules/visinger/wavenet.py cond_layer; the g-convs of the flow and the generator, decoder.py:38-39): y[b][m] = bias[m] + sum_c W[m][c] x[b][c].
// On the tile kernels the items are grid.z and the time axis the tile's columns -- one valid column in 128 or 256, a full K loop per
// item and row tile: 63 us for 256 -> 1536 at B = 32 (tools/conv_census.py, round 6).  Here a workgroup takes one 32-row tile of W for up
// to 32 items: x staged once in LDS (zero-padded to the chunk grid), the weights read straight from the fp32 fragment buffer Wp (lane =
// row: consecutive rows are consecutive float4s), fp32 FMAs; VS_MATH_BF16 rounds both operands to bf16 first, as its matrix kernels do.
struct T1Params {
    const float *x;
    long long x_bs;
    const float *wp;     // Wp[m_tile][chunk][quad(2)][64][4] (k = 1: one tap)
    const float *biasp;  // packed bias [MT_alloc * 32]
    const float *bias_b;
    long long bias_b_bs;
    float *y;
    long long y_bs;
    int B, Cin, c_out, nchunks, bf16;
    float scale;
};
__global__ void __launch_bounds__(256) conv_t1_kernel(const T1Params p) {
    extern __shared__ __attribute__((aligned(16))) float t1_xs[];       // [items of this workgroup][nchunks * 16]
    const int tid = threadIdx.x, row = tid & 31, bg = tid >> 5;
    const int mt = blockIdx.x, b0 = blockIdx.y * 32;
    const int nb = min(32, p.B - b0), CP = p.nchunks * 16;
    // (sixteen loads per thread in flight, unconditional on clamped indices: one load per loop trip was 32 round trips to a cold L2, 22 of the
    //  launch's 27 us)
    for (int e0 = 0; e0 < nb * CP; e0 += 256 * 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = min(e0 + tid + 256 * u, nb * CP - 1), bi = e / CP, c = e - bi * CP;
            v[u] = p.x[(long long)(b0 + bi) * p.x_bs + min(c, p.Cin - 1)];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = e0 + tid + 256 * u;
            if (e < nb * CP) {
                float t = (e % CP < p.Cin) ? v[u] : 0.f;
                if (p.bf16) t = u2f(rne_bf16(t) & 0xffff0000u);
                t1_xs[e] = t;
            }
        }
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float4 *w = reinterpret_cast<const float4 *>(p.wp) + (long long)mt * p.nchunks * 128;
    // the weights of eight chunks are requested before any is used: the launch is a few dozen workgroups, each a chain of loads from a cold L2
    for (int c0 = 0; c0 < p.nchunks; c0 += 8) {
        float4 wq[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 *wc = w + min(c0 + u, p.nchunks - 1) * 128;
            wq[u][0] = wc[row]; wq[u][1] = wc[row + 32]; wq[u][2] = wc[64 + row]; wq[u][3] = wc[64 + row + 32];      // (quad, parity): channels 8 quad + 2 i + parity
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (c0 + u < p.nchunks) {
                if (p.bf16) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        wq[u][t].x = u2f(rne_bf16(wq[u][t].x) & 0xffff0000u); wq[u][t].y = u2f(rne_bf16(wq[u][t].y) & 0xffff0000u);
                        wq[u][t].z = u2f(rne_bf16(wq[u][t].z) & 0xffff0000u); wq[u][t].w = u2f(rne_bf16(wq[u][t].w) & 0xffff0000u);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int bi = bg + 8 * i;
                    if (bi < nb) {
                        const float4 *xv = reinterpret_cast<const float4 *>(t1_xs + bi * CP + (c0 + u) * 16);
                        const float4 x0 = xv[0], x1 = xv[1], x2 = xv[2], x3 = xv[3];
                        float a = 0.f;                                     // (a chunk's sixteen products first, then onto the running sum: shorter rounding chains)
                        a += wq[u][0].x * x0.x; a += wq[u][1].x * x0.y; a += wq[u][0].y * x0.z; a += wq[u][1].y * x0.w;
                        a += wq[u][0].z * x1.x; a += wq[u][1].z * x1.y; a += wq[u][0].w * x1.z; a += wq[u][1].w * x1.w;
                        a += wq[u][2].x * x2.x; a += wq[u][3].x * x2.y; a += wq[u][2].y * x2.z; a += wq[u][3].y * x2.w;
                        a += wq[u][2].z * x3.x; a += wq[u][3].z * x3.y; a += wq[u][2].w * x3.z; a += wq[u][3].w * x3.w;
                        acc[i] += a;
                    }
                }
            }
        }
    }
    const int m = mt * 32 + row;
    if (m >= p.c_out) return;
    const float bias = p.biasp[m];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int bi = bg + 8 * i;
        if (bi < nb) {
            float v = acc[i] + bias;
            if (p.bias_b) v += p.bias_b[(long long)(b0 + bi) * p.bias_b_bs + m];
            p.y[(long long)(b0 + bi) * p.y_bs + m] = v * p.scale;
        }
    }
}

// KT > 0: compile-time taps KT and padding PAD (PAD <= 4, KT-1-PAD <= 4), dilation 1, T % 4 == 0, 16-B aligned rows: each thread
// produces 4 consecutive frames from three aligned float4 loads per input channel (its own 16 bytes, coalesced, plus
// its two neighbours' through L1), activates every value once and keeps the window in registers for all taps.
// KT == 0: generic runtime taps / dilation / alignment, one scalar load per (tap, frame).
template <int COUT, int KT, int PAD>
__global__ void __launch_bounds__(256) conv_small_kernel(const SmallParams p) {
    constexpr int NQ = 4;
    const int b = blockIdx.y;
    const int n0 = (blockIdx.x * 256 + threadIdx.x) * NQ;
    if (n0 >= p.Tout) return;
    const float *xb = p.x + (long long)b * p.x_bs;
    const unsigned short *xh = reinterpret_cast<const unsigned short *>(p.x) + (long long)b * p.x_bs;     // the same rows as bf16 elements
    const bool xbf = p.x_bf16 != 0;
    const float *mb = p.mask ? p.mask + (long long)b * p.Tin : nullptr;
    const bool act_lrelu = (p.in_act == VS_IN_LRELU || p.in_act == VS_IN_LRELU_MASK);
    const bool act_mask = (p.in_act >= VS_IN_MASK);
    float acc[COUT][NQ];
#pragma unroll
    for (int c = 0; c < COUT; ++c)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[c][q] = 0.f;
    if constexpr (KT > 0) {
        // window = frames n0-4 .. n0+7 ; frame n0 + q + k - pad is window[q + k - pad + 4]
        const bool okl = (n0 >= 4), okr = (n0 + 8 <= p.Tin);
        const int nl = okl ? n0 - 4 : n0, nr = okr ? n0 + 4 : n0;
        float mw[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) mw[i] = 1.f;
        if (act_mask) {
            const float4 a = *reinterpret_cast<const float4 *>(mb + nl), c4 = *reinterpret_cast<const float4 *>(mb + n0),
                         d = *reinterpret_cast<const float4 *>(mb + nr);
            mw[0] = a.x; mw[1] = a.y; mw[2] = a.z; mw[3] = a.w; mw[4] = c4.x; mw[5] = c4.y; mw[6] = c4.z; mw[7] = c4.w;
            mw[8] = d.x; mw[9] = d.y; mw[10] = d.z; mw[11] = d.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { mw[i] = okl ? mw[i] : 0.f; mw[8 + i] = okr ? mw[8 + i] : 0.f; }
        for (int ci = 0; ci < p.Cin; ++ci) {
            const float *xr = xb + (long long)ci * p.Tin;
            float4 a, c4, d;
            if (xbf) {
                const unsigned short *xq = xh + (long long)ci * p.Tin;
                a = bf4_to_f4(*reinterpret_cast<const uint2 *>(xq + nl));
                c4 = bf4_to_f4(*reinterpret_cast<const uint2 *>(xq + n0));
                d = bf4_to_f4(*reinterpret_cast<const uint2 *>(xq + nr));
            } else {
                a = *reinterpret_cast<const float4 *>(xr + nl);
                c4 = *reinterpret_cast<const float4 *>(xr + n0);
                d = *reinterpret_cast<const float4 *>(xr + nr);
            }
            float xw[12] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w, d.x, d.y, d.z, d.w};
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                float v = xw[i];
                if (act_lrelu) v = lrelu(v);
                xw[i] = v * mw[i];
            }
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const float wv = (c < p.Cout) ? p.w[((long long)c * p.Cin + ci) * KT + k] : 0.f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[c][q] += wv * xw[q + k - PAD + 4];
                }
            }
        }
    } else {
        for (int ci = 0; ci < p.Cin; ++ci) {
            const float *xr = xb + (long long)ci * p.Tin;
            for (int k = 0; k < p.K; ++k) {
                const int off = k * p.dil - p.pad;
                float xv[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int n = n0 + q + off;
                    const bool ok = (n >= 0 && n < p.Tin);
                    float v = ok ? (xbf ? u2f((unsigned)(xh + (long long)ci * p.Tin)[n] << 16) : xr[n]) : 0.f;
                    if (act_lrelu) v = lrelu(v);
                    if (act_mask) v *= ok ? mb[n] : 0.f;
                    xv[q] = v;
                }
#pragma unroll
                for (int c = 0; c < COUT; ++c) {
                    const float wv = (c < p.Cout) ? p.w[((long long)c * p.Cin + ci) * p.K + k] : 0.f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[c][q] += wv * xv[q];
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        if (c < p.Cout) {
            float bv = p.bias ? p.bias[c] : 0.f;
            if (p.bias_b) bv += p.bias_b[(long long)b * p.bias_b_bs + c];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int n = n0 + q;
                if (n < p.Tout) {
                    float v = (acc[c][q] + bv) * p.scale;
                    if (p.out_act == VS_OUT_TANH) v = tanhf(v);
                    else if (p.out_act == VS_OUT_RELU) v = fmaxf(v, 0.f);
                    if (p.out_mask) v *= mb[n];
                    p.y[(long long)b * p.y_bs + (long long)c * p.Tout + n] = v;
                }
            }
        }
    }
}

// effective (weight-norm folded) weights in the reference layout, for the VALU path
__global__ void fold_weights_kernel(const float *__restrict__ w, const float *__restrict__ scale, float *__restrict__ out,
                                    long long rows, long long cols) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * cols) out[i] = scale ? w[i] * scale[i / cols] : w[i];
}

// logdet[b] += sum of the item's per-tile partials, in a fixed order: lane l adds slots l, l + 64, ... sequentially, then a wavefront-shuffle
// butterfly (commutative per step: every lane ends with the same bits).  One wave per item; a few hundred slots.
__global__ void logdet_reduce_kernel(const float *__restrict__ part, int slots, float *__restrict__ logdet) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int i = lane; i < slots; i += 64) s += part[(long long)b * slots + i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) logdet[b] += s;
}

static inline int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
static inline int ceil_div_i(int a, int b) { return -floor_div(-a, b); }

}  // namespace vs

// ---------------------------------------------------------------------------------------------------------------
// C ABI

unsigned long long *g_stamp_buf = nullptr;   // debug hook, see vs_debug_set_stamp_buffer (also read by resblock_pair_split.hip)

using namespace vs;

template <int MT_W, int NT_W, int WAVES_M, int WAVES_N>
static int launch_cfg(const ConvParams &p, hipStream_t s) {
    constexpr int BN = 32 * NT_W * WAVES_N;
    constexpr int BM_TILES = MT_W * WAVES_M;
    auto kern = conv_mfma_kernel<MT_W, NT_W, WAVES_M, WAVES_N>;
    const size_t lds = (size_t)2 * CK * p.W * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.N, BN), (unsigned)ceil_div(p.MT, BM_TILES), (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_M * WAVES_N), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    set_last_kernel("conv_mfma_kernel<%d, %d, %d, %d>", MT_W, NT_W, WAVES_M, WAVES_N);
    return VS_OK;
}

template <int DIL, int WAVES_M, int WAVES_N, int TG = 0, int TT = 3>
static int launch_wino(const ConvParams &p, hipStream_t s) {
    constexpr int NBW = 2 * ((64 / DIL) * DIL);
    constexpr int BN = NBW * WAVES_N;
    auto kern = conv_wino_kernel<DIL, WAVES_M, WAVES_N, TG, TT>;
    const int Wh = (p.W + 1) >> 1, H = ((Wh + 15) & ~31) + 16;
    const int RP = (DIL == 1) ? H + Wh : p.W;      // LDS row pitch, as computed by the kernel
    const size_t lds = sizeof(float) * std::max<size_t>((size_t)2 * CK * RP, (size_t)WAVES_M * WAVES_N * 8 * (NBW + 8));
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.N, BN), (unsigned)ceil_div(p.MT, WAVES_M), (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_M * WAVES_N), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    set_last_kernel("conv_wino_kernel<%d, %d, %d, %d, %d>", DIL, WAVES_M, WAVES_N, TG, TT);
    return VS_OK;
}

template <int DIL>
static int launch_wino_dil(ConvParams &p, int MT, int span_w, int spec, hipStream_t s) {
    // spec: 0 generic (zero-padded last group), 7 / 11: the k-specialised straight-line instances
    // (k = 9 through the same structure, TG = 3 / TT = 3, measured +1 %: not instantiated)
    constexpr int NBW = 2 * ((64 / DIL) * DIL);
    if (MT % 4 == 0) {
        p.W = NBW + span_w;
        if (spec == 7) return launch_wino<DIL, 4, 1, 3, 1>(p, s);
        if (spec == 11) return launch_wino<DIL, 4, 1, 4, 2>(p, s);
        return launch_wino<DIL, 4, 1>(p, s);
    }
    if (MT % 2 == 0) {
        p.W = 2 * NBW + span_w;
        if (spec == 7) return launch_wino<DIL, 2, 2, 3, 1>(p, s);
        if (spec == 11) return launch_wino<DIL, 2, 2, 4, 2>(p, s);
        return launch_wino<DIL, 2, 2>(p, s);
    }
    p.W = 4 * NBW + span_w;
    if (spec == 11 && DIL == 1) return launch_wino<1, 1, 4, 4, 2>(p, s);
    return launch_wino<DIL, 1, 4>(p, s);
}

extern "C" {

const char *vs_last_error(void) { return g_err; }
const char *vs_last_kernel_name(void) { return g_last_kernel; }
int vs_abi_version(void) { return 7; }

int vs_set_option(const char *name, long long value) {
    VS_REQUIRE(name, "vs_set_option: NULL name");
    for (OptEntry &e : g_opts)
        if (strcmp(e.name, name) == 0) { e.value = value; return VS_OK; }
    set_error("vs_set_option: unknown option %s", name);
    return VS_EINVAL;
}
int vs_get_option(const char *name, long long *value) {
    VS_REQUIRE(name && value, "vs_get_option: NULL argument");
    for (const OptEntry &e : g_opts)
        if (strcmp(e.name, name) == 0) { *value = e.value; return VS_OK; }
    set_error("vs_get_option: unknown option %s", name);
    return VS_EINVAL;
}
int vs_reset_option(const char *name) {          /* back to the built-in default (not to the environment's value) */
    VS_REQUIRE(name, "vs_reset_option: NULL name");
    for (OptEntry &e : g_opts)
        if (strcmp(e.name, name) == 0) { e.value = e.dflt; return VS_OK; }
    set_error("vs_reset_option: unknown option %s", name);
    return VS_EINVAL;
}

int vs_device_info(char *buf, size_t n) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) cnt = 0;
    if (buf && n) {
        buf[0] = 0;
        if (cnt > 0) {
            hipDeviceProp_t pr;
            if (hipGetDeviceProperties(&pr, 0) == hipSuccess) snprintf(buf, n, "%s (%s)", pr.name, pr.gcnArchName);
        }
    }
    return cnt;
}

// Debug only (not part of the public header): per-workgroup phase stamps of the next conv launches are written
// to buf[64 * n_workgroups] (s_memrealtime, 100 MHz): [0] start, [1] first chunk staged, [2] main loop done, [3] epilogue
// issued, [7] HW ids.  NULL disables.
__attribute__((visibility("default"))) void vs_debug_set_stamp_buffer(void *buf) { g_stamp_buf = (unsigned long long *)buf; }

int vs_weightnorm_fold(const float *v, const float *g, float *w, int64_t rows, int64_t cols, void *stream) {
    VS_REQUIRE(v && g && w && rows > 0 && cols > 0, "vs_weightnorm_fold: bad arguments");
    hipStream_t s = as_stream(stream);
    float *scale = nullptr;
    VS_CHECK_HIP(hipMallocAsync((void **)&scale, sizeof(float) * rows, s));
    hipLaunchKernelGGL(rownorm_scale_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, v, g, scale,
                       (long long)rows, (long long)cols);
    hipLaunchKernelGGL(weightnorm_apply_kernel, dim3((unsigned)ceil_div(rows * cols, 256)), dim3(256), 0, s, v, scale, w,
                       (long long)rows, (long long)cols);
    VS_CHECK_HIP(hipGetLastError());
    VS_CHECK_HIP(hipFreeAsync(scale, s));
    return VS_OK;
}

int vs_conv_create(vs_conv_t **out, int kind, int c_in, int c_out, int k, int dil, int pad, unsigned flags) {
    VS_REQUIRE(out, "vs_conv_create: out is NULL");
    *out = nullptr;
    VS_REQUIRE(kind >= VS_CONV1D && kind <= VS_CONV1D_PAIRED, "vs_conv_create: unknown kind %d", kind);
    VS_REQUIRE(c_in > 0 && c_out > 0 && k > 0 && dil > 0 && pad >= 0, "vs_conv_create: bad dims");
    VS_REQUIRE(kind != VS_CONV1D_PAIRED || (c_out % 2 == 0), "vs_conv_create: PAIRED needs even c_out");
    VS_REQUIRE((flags & ~(VS_CONV_FLIP_IN | VS_CONV_FLIP_OUT | VS_CONV_ADJOINT)) == 0, "vs_conv_create: unknown flags %u", flags);
    VS_REQUIRE(!(flags & VS_CONV_ADJOINT) || (kind == VS_CONV1D && !(flags & (VS_CONV_FLIP_IN | VS_CONV_FLIP_OUT))),
               "vs_conv_create: VS_CONV_ADJOINT goes with a plain VS_CONV1D");
    vs_conv *h = new (std::nothrow) vs_conv();
    if (!h) { set_error("out of host memory"); return VS_ENOMEM; }
    h->kind = kind; h->c_in = c_in; h->c_out = c_out; h->k = k; h->dil = dil; h->pad = pad; h->flags = flags;
    h->Hh = c_out / 2;
    if (kind == VS_CONV_TRANSPOSE1D) {
        const int u = dil;
        h->M = u * c_out;
        h->dmin = ceil_div_i(-(u - 1) - pad, u);
        const int dmax = floor_div(k - 1 - pad, u);
        h->KT = dmax - h->dmin + 1;
        h->off0 = -h->dmin;      // x index = q - delta = q - dmin - tap
        h->tstep = -1;
    } else {
        h->M = (kind == VS_CONV1D_PAIRED) ? 2 * 32 * (int)ceil_div(h->Hh, 32) : c_out;
        h->KT = k;
        h->off0 = -pad;
        h->tstep = dil;
        h->dmin = 0;
    }
    const int off_last = h->off0 + (h->KT - 1) * h->tstep;
    h->lo = std::min(h->off0, off_last);
    h->span = std::max(h->off0, off_last) - h->lo;
    if (h->span > MAX_SPAN) {
        set_error("vs_conv_create: receptive span %d exceeds the LDS window (%d)", h->span, MAX_SPAN);
        delete h;
        return VS_EUNSUPPORTED;
    }
    h->MT = (int)ceil_div(h->M, 32);
    h->MT_alloc = (int)ceil_div(h->MT, MT_ALLOC) * MT_ALLOC;
    h->nchunks = (int)ceil_div(c_in, CK);
    h->CP = h->nchunks * (CK / 2);
    // F(2,3) minimal-filtering path: stride-1 "same" convs with >= 3 taps, dilation 1/3/5, whole 32-row tiles
    if (kind == VS_CONV1D && k >= 3 && (k & 1) && (dil == 1 || dil == 3 || dil == 5) && pad == dil * (k - 1) / 2 &&
        c_out % 32 == 0 && 3 * (int)ceil_div(k, 3) * dil <= MAX_SPAN)
        h->wino_groups = (int)ceil_div(k, 3);
    h->wino_k7 = h->wino_groups == 3 && k == 7 && (h->MT % 2 == 0) && !opt(OPT_NO_WINO_K7);
    h->wino_k11 = h->wino_groups == 4 && k == 11 && ((h->MT & 1) == 0 || dil == 1) && !opt(OPT_NO_WINO_K7);
    // Default arithmetic: the split-f16 x3 engine (two f16 planes under a power-of-two scale per staged tile, three cross products) --
    // measured x1.3 .. x1.45 the rate of the split-bf16 x6 engine on the 128- / 256-channel convs (tools/conv_bench.py) and CLOSER to the
    // fp64 result than it and than the fp32 MFMA on every case of tools/split3_check.py / tools/conv_accuracy.py (half the products summed
    // in fp32).  VS_CONV_MATH=0 / 1 / 3 / 6: process-wide A/B switch for handles created from here on.
    h->math = VS_MATH_SPLIT3;
    {
        const int m = (int)opt(OPT_CONV_MATH);
        if (m == VS_MATH_F32 || m == VS_MATH_BF16 || m == VS_MATH_SPLIT6 || m == VS_MATH_SPLIT3) h->math = m;
    }
    *out = h;
    return VS_OK;
}

// {s_w, 1 / s_w, max |w| bits of the even packs, of the odd packs}
static int reserve_wscale(vs_conv *h, hipStream_t s) {
    if (h->wsc.p) return VS_OK;
    VS_TRY(h->wsc.reserve(32));      // {s_w, 1 / s_w, max |w| bits of even / odd packs, scratch maximum of a vs_conv_set_math re-pack, 3 spare words}
    VS_CHECK_HIP(hipMemsetAsync(h->wsc.p, 0, 32, s));
    return VS_OK;
}

// maxbits_ready: the fp32 pack of THIS weight version has just left the conv's largest |w| in its slot (vs_conv_set_weights); otherwise
// (vs_conv_set_math on a bound handle) pack_split finds it with a pass of its own over the fp32 fragments
static int pack_split_planes(vs_conv *h, hipStream_t s, bool maxbits_ready = false) {
    const int npl = split_planes(h->math);
    VS_TRY(h->ws.reserve((size_t)h->MT_alloc * h->KT * h->nchunks * npl * 64 * 16));
    vs_split_pack q;
    q.wp = h->wp.as<float>(); q.ws = h->ws.p; q.MT_alloc = h->MT_alloc; q.KT = h->KT; q.nchunks = h->nchunks; q.terms = h->math;
    q.wscale = nullptr;
    q.maxbits = nullptr;
    q.scratch = nullptr;
    if (h->math == VS_MATH_SPLIT3) {
        VS_TRY(reserve_wscale(h, s));
        q.wscale = h->wsc.as<float>();
        q.maxbits = maxbits_ready ? reinterpret_cast<const unsigned *>(q.wscale + 2) + (h->pack_gen & 1) : nullptr;
        q.scratch = reinterpret_cast<unsigned *>(q.wscale + 4);      // the handle's own word: no allocation on any pack path (stream capture)
    }
    return pack_split(q, s);
}

int vs_conv_set_math(vs_conv_t *h, int math, void *stream) {
    VS_REQUIRE(h, "vs_conv_set_math: NULL handle");
    VS_REQUIRE(math == VS_MATH_F32 || math == VS_MATH_BF16 || math == VS_MATH_SPLIT6 || math == VS_MATH_SPLIT3, "vs_conv_set_math: unknown arithmetic %d", math);
    if (h->math == math) return VS_OK;
    h->math = math;
    if (math && h->weights_set) VS_TRY(pack_split_planes(h, as_stream(stream)));
    return VS_OK;
}

int vs_conv_get_math(const vs_conv_t *h) { return h ? h->math : -1; }

void vs_conv_destroy(vs_conv_t *h) { delete h; }

int64_t vs_conv_out_len(const vs_conv_t *h, int64_t T) {
    if (!h) return -1;
    if (h->kind == VS_CONV_TRANSPOSE1D) return (T - 1) * h->dil - 2 * h->pad + h->k;
    return T + 2 * h->pad - (int64_t)h->dil * (h->k - 1);
}

int vs_conv_set_weights(vs_conv_t *h, const float *w, const float *g, const float *bias, void *stream) {
    VS_REQUIRE(h && w, "vs_conv_set_weights: NULL handle or weight");
    VS_REQUIRE(!(h->flags & VS_CONV_ADJOINT) || !g, "vs_conv_set_weights: an ADJOINT handle takes the plain forward weight (g must be NULL)");
    hipStream_t s = as_stream(stream);
    const size_t n = (size_t)h->MT_alloc * h->KT * h->CP * 64;
    VS_TRY(h->wp.reserve(n * sizeof(float)));
    VS_TRY(h->biasp.reserve((size_t)h->MT_alloc * 32 * sizeof(float)));
    const float *scale = nullptr;
    if (g) {
        // weight_norm(dim=0): rows = dim 0 of the weight (c_out for Conv1d, c_in for ConvTranspose1d)
        const long long rows = (h->kind == VS_CONV_TRANSPOSE1D) ? h->c_in : h->c_out;
        const long long cols = (long long)h->c_in * h->c_out * h->k / rows;
        VS_TRY(h->scale.reserve(rows * sizeof(float)));
        hipLaunchKernelGGL(rownorm_scale_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, s, w, g,
                           h->scale.as<float>(), rows, cols);
        scale = h->scale.as<float>();
    }
    PackParams q;
    q.w = w; q.scale = scale; q.bias = bias; q.wp = h->wp.as<float>(); q.biasp = h->biasp.as<float>();
    q.kind = h->kind; q.c_in = h->c_in; q.c_out = h->c_out; q.k = h->k; q.up = h->dil; q.pad = h->pad; q.dmin = h->dmin;
    q.KT = h->KT; q.CP = h->CP; q.MT_alloc = h->MT_alloc; q.Hh = h->Hh; q.flags = h->flags; q.wino_tail1 = 0;
    q.maxbits = q.maxbits_clear = nullptr;
    if (h->math == VS_MATH_SPLIT3) {
        // split-f16: the planes need the largest weight first (one scale per conv): fp32 fragments, then the scaled planes
        const long long total = std::max<long long>((long long)n, (long long)h->MT_alloc * 32);
        VS_TRY(reserve_wscale(h, s));
        ++h->pack_gen;
        q.maxbits = reinterpret_cast<unsigned *>(h->wsc.as<float>() + 2) + (h->pack_gen & 1);
        q.maxbits_clear = reinterpret_cast<unsigned *>(h->wsc.as<float>() + 2) + ((h->pack_gen + 1) & 1);
        hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 1024)), dim3(256), 0, s, q);
        VS_CHECK_HIP(hipGetLastError());
        VS_TRY(pack_split_planes(h, s, true));
    } else if (h->math) {
        // bf16-pipe arithmetic: fp32 fragments + bf16 planes in one launch
        const int npl = split_planes(h->math);
        VS_TRY(h->ws.reserve((size_t)h->MT_alloc * h->KT * h->nchunks * npl * 64 * 16));
        const long long total = std::max<long long>((long long)h->MT_alloc * h->KT * h->nchunks * 64, (long long)h->MT_alloc * 32);
        hipLaunchKernelGGL(pack_conv_split_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, q, h->ws.p, npl);
    } else {
        const long long total = std::max<long long>((long long)n, (long long)h->MT_alloc * 32);
        hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 1024)), dim3(256), 0, s, q);
    }
    h->wino_packed = false;        // (the F(2,3) transform of the fp32 engine is rebuilt from Wp by the first launch that uses it)
    if (h->kind == VS_CONV1D && h->c_out <= 4 && !(h->flags & VS_CONV_ADJOINT)) {
        const long long cols = (long long)h->c_in * h->k;
        VS_TRY(h->weff.reserve((size_t)h->c_out * cols * sizeof(float)));
        hipLaunchKernelGGL(fold_weights_kernel, dim3((unsigned)ceil_div(h->c_out * cols, 256)), dim3(256), 0, s, w, scale,
                           h->weff.as<float>(), (long long)h->c_out, cols);
        h->has_bias = bias != nullptr;
        if (bias) {
            VS_TRY(h->beff.reserve((size_t)h->c_out * sizeof(float)));
            VS_CHECK_HIP(hipMemcpyAsync(h->beff.p, bias, (size_t)h->c_out * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
    }
    VS_CHECK_HIP(hipGetLastError());
    h->weights_set = true;
    return VS_OK;
}

// fp32 fragments + bias of handle h from (w, bias): the PackParams of vs_conv_set_weights for a plain (already folded) weight
static void fill_pack_params(vs_conv *h, const float *w, const float *bias, PackParams &q) {
    q.w = w; q.scale = nullptr; q.bias = bias; q.wp = h->wp.as<float>(); q.biasp = h->biasp.as<float>();
    q.kind = h->kind; q.c_in = h->c_in; q.c_out = h->c_out; q.k = h->k; q.up = h->dil; q.pad = h->pad; q.dmin = h->dmin;
    q.KT = h->KT; q.CP = h->CP; q.MT_alloc = h->MT_alloc; q.Hh = h->Hh; q.flags = h->flags; q.wino_tail1 = 0;
    ++h->pack_gen;
    q.maxbits = reinterpret_cast<unsigned *>(h->wsc.as<float>() + 2) + (h->pack_gen & 1);
    q.maxbits_clear = reinterpret_cast<unsigned *>(h->wsc.as<float>() + 2) + ((h->pack_gen + 1) & 1);
}

int vs_conv_set_weights_pair(vs_conv_t *h0, vs_conv_t *h1, const float *w, const float *bias0, void *stream) {
    VS_REQUIRE(h0 && h1 && w, "vs_conv_set_weights_pair: NULL handle or weight");
    // h1 reads the SAME weight tensor as the grad-input conv of h0: anything else would index `w` out of bounds in the pack
    VS_REQUIRE((h1->flags & VS_CONV_ADJOINT) && !(h0->flags & VS_CONV_ADJOINT) && h1->c_in == h0->c_out && h1->c_out == h0->c_in && h1->k == h0->k &&
                   (h0->kind == VS_CONV_TRANSPOSE1D || h1->dil == h0->dil),
               "vs_conv_set_weights_pair: the second handle is not the VS_CONV_ADJOINT counterpart of the first (c_in / c_out swapped, same k and dilation)");
    // the fused form serves the training step's common case; everything else is the two plain calls
    const bool fused = h0->math == VS_MATH_SPLIT3 && h1->math == VS_MATH_SPLIT3 && !(h0->kind == VS_CONV1D && h0->c_out <= 4) &&
                       !(h1->kind == VS_CONV1D && h1->c_out <= 4 && !(h1->flags & VS_CONV_ADJOINT));
    if (!fused) {
        VS_TRY(vs_conv_set_weights(h0, w, nullptr, bias0, stream));
        return vs_conv_set_weights(h1, w, nullptr, nullptr, stream);
    }
    hipStream_t s = as_stream(stream);
    PackParams q[2];
    vs_conv *hs[2] = {h0, h1};
    long long work = 0;
    for (int i = 0; i < 2; ++i) {
        vs_conv *h = hs[i];
        const size_t n = (size_t)h->MT_alloc * h->KT * h->CP * 64;
        VS_TRY(h->wp.reserve(n * sizeof(float)));
        VS_TRY(h->biasp.reserve((size_t)h->MT_alloc * 32 * sizeof(float)));
        VS_TRY(reserve_wscale(h, s));
        VS_TRY(h->ws.reserve((size_t)h->MT_alloc * h->KT * h->nchunks * 2 * 64 * 16));
        fill_pack_params(h, w, i == 0 ? bias0 : nullptr, q[i]);
        work = std::max<long long>(work, std::max<long long>((long long)n, (long long)h->MT_alloc * 32));
    }
    hipLaunchKernelGGL(pack_conv_pair_kernel, dim3((unsigned)std::min<long long>(ceil_div(work, 256), 1024), 2), dim3(256), 0, s, q[0], q[1]);
    VS_CHECK_HIP(hipGetLastError());
    vs_split_pack sp[2];
    for (int i = 0; i < 2; ++i) {
        vs_conv *h = hs[i];
        sp[i].wp = h->wp.as<float>(); sp[i].ws = h->ws.p; sp[i].MT_alloc = h->MT_alloc; sp[i].KT = h->KT; sp[i].nchunks = h->nchunks; sp[i].terms = 3;
        sp[i].wscale = h->wsc.as<float>();
        sp[i].maxbits = reinterpret_cast<const unsigned *>(sp[i].wscale + 2) + (h->pack_gen & 1);
        sp[i].scratch = nullptr;
        h->wino_packed = false;
        h->weights_set = true;
    }
    return pack_split_pair(sp[0], sp[1], s);
}

// ---- vs_conv_set_weights_batch: the table of one batch (PackParams x n | vs_split_pack x n | first pack block x (n + 1) | first split block x (n + 1))
// goes to the device through one of a ring of pinned staging buffers, each with a device twin and an event recorded behind the launches that read
// it: a slot is reused only when its event has completed (normally long ago: a training step makes two or three batches).
namespace {
struct BatchSlot {
    void *host = nullptr, *dev = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
    int device = -1;          // the device `dev` and `ev` belong to (the ring served whichever device was current at first use)
};
// marks a slot busy from the moment its host buffer has been handed to an asynchronous copy, whatever happens afterwards (on an error behind the copy
// the slot stayed "free" and its host buffer could be rewritten while the copy was still in flight)
struct SlotInFlight {
    BatchSlot *sl;
    hipStream_t s;
    ~SlotInFlight() {
        if (hipEventRecord(sl->ev, s) == hipSuccess) sl->pending = true;
        else (void)hipStreamSynchronize(s);
    }
};
struct BatchArena {
    std::mutex mu;
    static constexpr int NSLOT = 8;
    BatchSlot slot[NSLOT];
    int next = 0;
};
BatchArena g_batch;

int batch_slot(size_t bytes, BatchSlot **out) {
    BatchSlot &sl = g_batch.slot[g_batch.next];
    g_batch.next = (g_batch.next + 1) % BatchArena::NSLOT;
    if (sl.pending) {
        VS_CHECK_HIP(hipEventSynchronize(sl.ev));
        sl.pending = false;
    }
    int cur = 0;
    VS_CHECK_HIP(hipGetDevice(&cur));
    if (sl.device != cur) {       // first use, or the process moved to another device: this slot's device buffer and event are re-made there
        if (sl.ev) (void)hipEventDestroy(sl.ev);
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
        sl.ev = nullptr;
        sl.host = sl.dev = nullptr;
        sl.cap = 0;
        sl.device = cur;
    }
    if (!sl.ev) VS_CHECK_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (bytes > sl.cap) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
        sl.host = sl.dev = nullptr;
        sl.cap = 0;
        const size_t cap = std::max<size_t>(bytes * 2, 64 * 1024);
        VS_CHECK_HIP(hipHostMalloc(&sl.host, cap, hipHostMallocDefault));
        VS_CHECK_HIP(hipMalloc(&sl.dev, cap));
        sl.cap = cap;
    }
    *out = &sl;
    return VS_OK;
}
}  // namespace

int vs_conv_set_weights_batch(vs_conv_t *const *handles, const float *const *w, const float *const *bias, int n, void *stream) {
    VS_REQUIRE(n >= 0 && (n == 0 || (handles && w)), "vs_conv_set_weights_batch: NULL array");
    hipStream_t s = as_stream(stream);
    // handles outside the batch kernels' case (other arithmetic, the <= 4-row VALU convs) take the plain call
    std::vector<int> idx;
    idx.reserve(n);
    for (int i = 0; i < n; ++i) {
        vs_conv *h = handles[i];
        VS_REQUIRE(h && w[i], "vs_conv_set_weights_batch: entry %d: NULL handle or weight", i);
        const bool batched = h->math == VS_MATH_SPLIT3 && !(h->kind == VS_CONV1D && h->c_out <= 4 && !(h->flags & VS_CONV_ADJOINT));
        if (!batched) VS_TRY(vs_conv_set_weights(h, w[i], nullptr, bias ? bias[i] : nullptr, stream));
        else idx.push_back(i);
    }
    const int m = (int)idx.size();
    if (m == 0) return VS_OK;
    if (m == 1) return vs_conv_set_weights(handles[idx[0]], w[idx[0]], nullptr, bias ? bias[idx[0]] : nullptr, stream);
    {   // no handle twice in one batch: each job flips its handle's weight-maximum slot (pack_gen), so a duplicate would make one job clear the slot the other reads
        std::unordered_set<const vs_conv *> seen;
        for (int j = 0; j < m; ++j)
            VS_REQUIRE(seen.insert(handles[idx[j]]).second, "vs_conv_set_weights_batch: entry %d: the same handle appears twice in one batch", idx[j]);
    }
    std::lock_guard<std::mutex> lock(g_batch.mu);
    const size_t off_sp = (size_t)m * sizeof(PackParams), off_b0 = off_sp + (size_t)m * sizeof(vs_split_pack),
                 off_b1 = off_b0 + (size_t)(m + 1) * sizeof(unsigned), bytes = off_b1 + (size_t)(m + 1) * sizeof(unsigned);
    BatchSlot *sl = nullptr;
    VS_TRY(batch_slot(bytes, &sl));
    char *hb = static_cast<char *>(sl->host);
    PackParams *q = reinterpret_cast<PackParams *>(hb);
    vs_split_pack *sp = reinterpret_cast<vs_split_pack *>(hb + off_sp);
    unsigned *b0 = reinterpret_cast<unsigned *>(hb + off_b0), *b1 = reinterpret_cast<unsigned *>(hb + off_b1);
    unsigned nb0 = 0, nb1 = 0;
    for (int j = 0; j < m; ++j) {
        vs_conv *h = handles[idx[j]];
        const size_t nel = (size_t)h->MT_alloc * h->KT * h->CP * 64;
        VS_TRY(h->wp.reserve(nel * sizeof(float)));
        VS_TRY(h->biasp.reserve((size_t)h->MT_alloc * 32 * sizeof(float)));
        VS_TRY(reserve_wscale(h, s));
        VS_TRY(h->ws.reserve((size_t)h->MT_alloc * h->KT * h->nchunks * 2 * 64 * 16));
        fill_pack_params(h, w[idx[j]], bias ? bias[idx[j]] : nullptr, q[j]);
        const long long work = std::max<long long>((long long)nel, (long long)h->MT_alloc * 32);
        b0[j] = nb0;
        nb0 += (unsigned)std::min<long long>(ceil_div(work, 256), 1024);
        sp[j].wp = h->wp.as<float>(); sp[j].ws = h->ws.p; sp[j].MT_alloc = h->MT_alloc; sp[j].KT = h->KT; sp[j].nchunks = h->nchunks; sp[j].terms = 3;
        sp[j].wscale = h->wsc.as<float>();
        sp[j].maxbits = reinterpret_cast<const unsigned *>(sp[j].wscale + 2) + (h->pack_gen & 1);
        sp[j].scratch = nullptr;
        b1[j] = nb1;
        nb1 += (unsigned)ceil_div((long long)h->MT_alloc * h->KT * h->nchunks * 64, 256);
        h->wino_packed = false;
        h->weights_set = true;
    }
    b0[m] = nb0;
    b1[m] = nb1;
    VS_CHECK_HIP(hipMemcpyAsync(sl->dev, sl->host, bytes, hipMemcpyHostToDevice, s));
    SlotInFlight in_flight{sl, s};         // (records the slot's event behind everything launched below, on every way out)
    const char *db = static_cast<const char *>(sl->dev);
    hipLaunchKernelGGL(pack_conv_multi_kernel, dim3(nb0), dim3(256), 0, s, reinterpret_cast<const PackParams *>(db),
                       reinterpret_cast<const unsigned *>(db + off_b0), m);
    VS_CHECK_HIP(hipGetLastError());
    VS_TRY(pack_split_multi(reinterpret_cast<const vs_split_pack *>(db + off_sp), reinterpret_cast<const unsigned *>(db + off_b1), m, nb1, s));
    return VS_OK;
}

int vs_conv_forward(vs_conv_t *h, const vs_conv_io_t *io, void *stream) {
    VS_REQUIRE(h && io, "vs_conv_forward: NULL handle or io");
    VS_REQUIRE(h->weights_set, "vs_conv_forward: weights not set");
    VS_REQUIRE(io->x && io->B > 0 && io->T > 0, "vs_conv_forward: bad input");
    VS_REQUIRE(io->out[0].y, "vs_conv_forward: out[0].y is NULL");
    const int64_t Tout = vs_conv_out_len(h, io->T);
    VS_REQUIRE(Tout > 0, "vs_conv_forward: empty output");
    VS_REQUIRE(io->B <= 65535, "vs_conv_forward: B too large for grid.z");
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.x = io->x;
    p.x_bs = io->x_bs ? io->x_bs : (long long)h->c_in * io->T;
    p.wp = h->wp.as<float>();
    p.biasp = h->biasp.as<float>();
    p.bias_b = io->bias_b;
    p.bias_b_bs = io->bias_b_bs ? io->bias_b_bs : h->c_out;
    p.mask = io->mask;
    p.logdet = io->logdet;
    p.kind = h->kind;
    p.pair_mode = io->pair_mode;
    p.in_act = io->in_act;
    p.B = (int)io->B; p.Cin = h->c_in; p.Tin = (int)io->T;
    p.M = h->M; p.MT = h->MT; p.c_out = h->c_out; p.Hh = h->Hh;
    p.Tout = (int)Tout;
    p.N = (h->kind == VS_CONV_TRANSPOSE1D) ? (int)ceil_div(Tout, h->dil) : (int)Tout;
    p.KT = h->KT; p.CP = h->CP; p.nchunks = h->nchunks;
    p.off0 = h->off0; p.tstep = h->tstep; p.lo = h->lo;
    p.stamps = g_stamp_buf;
    p.up = (h->kind == VS_CONV_TRANSPOSE1D) ? h->dil : 1;
    p.upK = h->k; p.uppad = h->pad; p.dmin = h->dmin;
    const int rows_out = (h->kind == VS_CONV1D_PAIRED) ? h->Hh : h->c_out;
    p.split_row = (io->split_row > 0 && io->split_row < rows_out && h->kind != VS_CONV1D_PAIRED) ? io->split_row : 0;
    VS_REQUIRE((io->x_dtype == VS_DTYPE_F32 || io->x_dtype == VS_DTYPE_BF16) && (io->y_dtype == VS_DTYPE_F32 || io->y_dtype == VS_DTYPE_BF16),
               "vs_conv_forward: unknown element type");
    p.x_bf16 = io->x_dtype == VS_DTYPE_BF16;
    p.y_bf16 = io->y_dtype == VS_DTYPE_BF16;
    if (p.x_bf16 || p.y_bf16) {
        VS_REQUIRE(h->math == VS_MATH_BF16, "vs_conv_forward: bf16-resident tensors need the plain-bf16 arithmetic (VS_MATH_BF16)");
        VS_REQUIRE(h->kind != VS_CONV1D_PAIRED && !p.split_row && (io->out[0].mode == VS_OUT_LINEAR || !p.y_bf16),
                   "vs_conv_forward: bf16-resident tensors go with plain LINEAR launches (no PAIRED kind, no split_row)");
    }
    bool need_mask = (io->in_act >= VS_IN_MASK);
    bool need_out_mask = false;      // a mask indexed by OUTPUT position: only defined where T_out == T (never for a transposed conv)
    VS_REQUIRE(io->in_act >= VS_IN_NONE && io->in_act <= VS_IN_LRELU_MASK, "vs_conv_forward: bad in_act");
    for (int s = 0; s < 2; ++s) {
        const vs_conv_out_t &o = io->out[s];
        OutSpec &d = p.out[s];
        const int rows = (s == 0) ? (p.split_row ? p.split_row : rows_out) : (rows_out - p.split_row);
        const long long dflt = (long long)rows * Tout;
        d.y = o.y; d.res = o.res; d.acc = o.acc;
        d.y_bs = o.y_bs ? o.y_bs : dflt; d.res_bs = o.res_bs ? o.res_bs : dflt; d.acc_bs = o.acc_bs ? o.acc_bs : dflt;
        d.scale = (o.scale == 0.f) ? 1.f : o.scale;   // 0 = unset
        d.out_act = o.out_act; d.out_mask = o.out_mask; d.mode = o.mode;
        d.rows = rows;
        if (s == 0 || p.split_row) {
            need_out_mask |= (o.out_mask != 0) || (o.mode != VS_OUT_LINEAR);
            VS_REQUIRE(o.mode == VS_OUT_LINEAR || o.res, "vs_conv_forward: coupling mode needs res (x1)");
        }
    }
    VS_REQUIRE(!p.split_row || io->out[1].y, "vs_conv_forward: split_row set but out[1].y is NULL");
    VS_REQUIRE(h->kind != VS_CONV_TRANSPOSE1D || (!io->out[0].res && !io->out[0].acc && !p.split_row),
               "vs_conv_forward: res / acc / split_row are not supported with a transposed conv");
    VS_REQUIRE((long long)h->c_in * io->T * 4 < (1ll << 31) && (long long)rows_out * Tout * 4 < (1ll << 31),
               "vs_conv_forward: one item's tensor exceeds the 2 GiB buffer-descriptor range");
    if (h->kind == VS_CONV1D_PAIRED) {
        VS_REQUIRE(io->pair_mode >= VS_PAIR_GATE && io->pair_mode <= VS_PAIR_COUPLING_INV, "bad pair_mode");
        if (io->pair_mode != VS_PAIR_GATE) {
            VS_REQUIRE(io->out[0].res, "vs_conv_forward: coupling pair mode needs out[0].res (x1)");
            need_out_mask = true;
        }
    }
    need_mask |= need_out_mask;
    VS_REQUIRE(!need_mask || io->mask, "vs_conv_forward: mask required but NULL");
    // (an INPUT mask -- VS_IN_MASK / VS_IN_LRELU_MASK, indexed by input frame while staging -- is fine with any kind)
    VS_REQUIRE(Tout == io->T || !need_out_mask, "vs_conv_forward: an output mask needs T_out == T (not with a transposed / unpadded conv)");
    hipStream_t s = as_stream(stream);
    const bool trace = opt(OPT_TRACE) != 0;   // debug: one line per launch on stderr
    if (trace)
        fprintf(stderr, "[vs_conv_forward] kind %d %d->%d k%d d%d pad%d flags%u B%d T%d Tout%d in_act%d split%d mode%d,%d res%d acc%d "
                        "mask%d bias_b%d pair%d x%p y%p y1%p\n", h->kind, h->c_in, h->c_out, h->k, h->dil, h->pad, h->flags, p.B, p.Tin,
                p.Tout, p.in_act, p.split_row, p.out[0].mode, p.out[1].mode, p.out[0].res != nullptr, p.out[0].acc != nullptr,
                p.mask != nullptr, p.bias_b != nullptr, p.pair_mode, (const void *)p.x, (void *)p.out[0].y, (void *)p.out[1].y);

    // (the VALU kernel reduces over c_in * k serially per thread: right for conv_post 32 -> 1 and the pitch head 192 -> 2, which
    // stream a long input once; the discriminators' 1024 -> 1 conv_post over a few thousand positions needs the parallelism of
    // the MFMA tiles even at 1 valid row in 32)
    if (h->kind == VS_CONV1D && h->c_out <= 4 && h->c_in * h->k <= 2048 && !(h->flags & (VS_CONV_FLIP_IN | VS_CONV_FLIP_OUT | VS_CONV_ADJOINT)) && !p.split_row &&
        !p.y_bf16 &&
        !io->out[0].res && !io->out[0].acc && io->out[0].mode == VS_OUT_LINEAR && !opt(OPT_NO_SMALL_CONV)) {
        SmallParams q;
        q.x = p.x; q.x_bs = p.x_bs; q.w = h->weff.as<float>(); q.bias = h->has_bias ? h->beff.as<float>() : nullptr;
        q.bias_b = p.bias_b; q.bias_b_bs = p.bias_b_bs; q.mask = p.mask; q.y = p.out[0].y; q.y_bs = p.out[0].y_bs;
        q.B = p.B; q.Cin = h->c_in; q.Cout = h->c_out; q.Tin = p.Tin; q.Tout = p.Tout; q.K = h->k; q.dil = h->dil;
        q.pad = h->pad; q.in_act = p.in_act; q.out_act = p.out[0].out_act; q.out_mask = p.out[0].out_mask;
        q.scale = p.out[0].scale;
        q.x_bf16 = p.x_bf16;
        dim3 grid((unsigned)ceil_div(p.Tout, 256 * 4), (unsigned)p.B);
        auto al16 = [](const void *q2) { return (reinterpret_cast<uintptr_t>(q2) & 15u) == 0; };
        const bool vec_ok = (p.Tin % 4 == 0) && (Tout == io->T) && al16(q.x) && (q.x_bs % 4 == 0) && (!q.mask || al16(q.mask));
        const bool k7 = vec_ok && (h->k == 7 && h->dil == 1 && h->pad == 3), k1 = vec_ok && (h->k == 1 && h->pad == 0);
#define VS_SMALL(CO)                                                                                       \
        do {                                                                                               \
            if (k7) hipLaunchKernelGGL((conv_small_kernel<CO, 7, 3>), grid, dim3(256), 0, s, q);           \
            else if (k1) hipLaunchKernelGGL((conv_small_kernel<CO, 1, 0>), grid, dim3(256), 0, s, q);      \
            else hipLaunchKernelGGL((conv_small_kernel<CO, 0, 0>), grid, dim3(256), 0, s, q);              \
            set_last_kernel("conv_small_kernel<%d, %d, %d>", CO, k7 ? 7 : (k1 ? 1 : 0), k7 ? 3 : 0);       \
        } while (0)
        if (h->c_out == 1) VS_SMALL(1);
        else if (h->c_out == 2) VS_SMALL(2);
        else VS_SMALL(4);
#undef VS_SMALL
        VS_CHECK_HIP(hipGetLastError());
        return VS_OK;
    }

    // one frame per item, k = 1 (conditioning vectors): conv_t1_kernel -- the tile kernels would compute one valid column in 128
    if (h->kind == VS_CONV1D && h->k == 1 && io->T == 1 && h->flags == 0 && !p.split_row && !p.x_bf16 && !p.y_bf16 && io->in_act == VS_IN_NONE &&
        !io->out[0].res && !io->out[0].acc && io->out[0].mode == VS_OUT_LINEAR && io->out[0].out_act == VS_OUT_NONE && !io->out[0].out_mask &&
        h->nchunks * 16 * 32 * 4 <= 64 * 1024 && !opt(OPT_NO_T1_CONV)) {
        T1Params q;
        q.x = static_cast<const float *>(p.x); q.x_bs = p.x_bs; q.wp = h->wp.as<float>(); q.biasp = p.biasp; q.bias_b = p.bias_b; q.bias_b_bs = p.bias_b_bs;
        q.y = static_cast<float *>(p.out[0].y); q.y_bs = p.out[0].y_bs; q.B = p.B; q.Cin = h->c_in; q.c_out = h->c_out; q.nchunks = h->nchunks;
        q.bf16 = (h->math == VS_MATH_BF16); q.scale = p.out[0].scale;
        const int nbmax = std::min(32, p.B);
        hipLaunchKernelGGL(conv_t1_kernel, dim3((unsigned)h->MT, (unsigned)ceil_div(p.B, 32)), dim3(256), (size_t)nbmax * h->nchunks * 16 * sizeof(float), s, q);
        VS_CHECK_HIP(hipGetLastError());
        set_last_kernel("conv_t1_kernel");
        return VS_OK;
    }

    if (h->kind == VS_CONV1D_PAIRED) {
        p.row_lo = 0;
        p.row_hi = h->c_out;
        const bool want_ld = (io->pair_mode == VS_PAIR_COUPLING_FWD) && io->logdet;
        if (want_ld) {
            // log-det = sum of logs over (channel, frame): every wave leaves the sum of its tile in its own slot, then one workgroup per item
            // adds the slots in index order onto logdet[b] -- the same bits every run (an atomicAdd per tile summed in retirement order)
            p.ld_nt = (int)ceil_div(p.N, 32) + 64;                      // (+64: first-tile indices of the trailing waves of the widest tile shape)
            p.ld_slots = p.ld_nt * (int)ceil_div(h->MT, 2);
            VS_TRY(h->ldpart.reserve((size_t)p.B * p.ld_slots * sizeof(float)));
            p.ld_part = h->ldpart.as<float>();
            VS_CHECK_HIP(hipMemsetAsync(p.ld_part, 0, (size_t)p.B * p.ld_slots * sizeof(float), s));
        }
        int rc;
        if (h->math) {
            p.wp = h->ws.as<float>();
            p.wscale = h->wsc.as<float>();
            if (!opt(OPT_NO_KTAP) && p.Cin % CK == 0 && ktap_pair_instance(h->math, h->KT, (p.x_bf16 ? 1 : 0) | (p.y_bf16 ? 2 : 0), p.in_act, h->MT))
                rc = launch_ktap_pair(p, h->math, s);      // (conv_ktap_pair.hip: taps unrolled, staging in the MFMA shadows; bit-identical)
            else
            rc = launch_split(p, (h->MT >= 4) ? 4 : 5, h->math, h->span, s);
        } else if (h->MT >= 4) {
            p.W = 128 + h->span;
            rc = launch_cfg<2, 2, 2, 2>(p, s);
        } else {
            p.W = 256 + h->span;
            rc = launch_cfg<2, 2, 1, 4>(p, s);
        }
        if (rc == VS_OK && want_ld) {
            hipLaunchKernelGGL(logdet_reduce_kernel, dim3((unsigned)p.B), dim3(64), 0, s, p.ld_part, p.ld_slots, p.logdet);
            VS_CHECK_HIP(hipGetLastError());
        }
        return rc;
    }
    {
        auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
        bool ok = (h->kind == VS_CONV1D) && (Tout % 4 == 0) && !opt(OPT_NO_FAST_EPI);
        for (int s2 = 0; s2 < (p.split_row ? 2 : 1) && ok; ++s2) {
            const OutSpec &d = p.out[s2];
            ok = ok && d.mode == VS_OUT_LINEAR && al16(d.y) && (d.y_bs % 4 == 0) && (!d.res || (al16(d.res) && d.res_bs % 4 == 0)) &&
                 (!d.acc || (al16(d.acc) && d.acc_bs % 4 == 0));
        }
        ok = ok && (!p.mask || al16(p.mask)) && (!p.split_row || p.split_row % 32 == 0);
        p.fast_epi = ok ? 1 : 0;
    }
    // F(2,3) path where it measured faster than the direct engine (tools/conv_bench.py, B=32 production shapes): with an even
    // number of 32-row tiles every dilation-1 conv (k=3: +10..17 %, k=7: +0..2 %, k=9: +36..44 %, k=11: +21..25 %) and the
    // k >= 9 convs at any dilation (+7..16 %); with an odd tile count (C_out = 32: the 32 x 512 workgroup shape, 2-slot
    // fragment ring) only k >= 9 at dilation 1 (+10 %).  The dilated k=3 / k=7 convs lose 2..25 % (idle pair columns, 8-byte
    // epilogue runs, fewer MFMAs to hide the same staging behind) and stay on the direct engine.
    // VS_WINO_FORCE=1 / VS_NO_WINO=1: test / A-B switches.
    // k = 7 through the straight-line instances (direct-form last tap, 10/14 of the direct MFMAs): +11..22 % at every dilation;
    // k = 11 likewise with an F(2,2) last group (15/22): another 6 %.
    const bool wino_pays = (h->MT & 1) ? (h->k >= 9 && h->dil == 1) : (h->dil == 1 || h->k >= 9 || h->wino_k7);
    if (!h->math && h->wino_groups && (wino_pays || opt(OPT_WINO_FORCE)) && !p.split_row && io->out[0].mode == VS_OUT_LINEAR && !opt(OPT_NO_WINO)) {
        ConvParams q = p;
        if (!h->wino_packed) {
            const size_t nw = (size_t)h->MT_alloc * h->nchunks * h->wino_groups * 2 * 16 * 64;
            VS_TRY(h->wpw.reserve(nw * sizeof(float)));
            PackParams qw;
            memset(&qw, 0, sizeof(qw));
            qw.w = h->wp.as<float>(); qw.wp = h->wpw.as<float>(); qw.k = h->k; qw.KT = h->KT; qw.MT_alloc = h->MT_alloc;
            qw.wino_tail1 = (h->wino_k7 || h->wino_k11) ? 1 : 0;
            hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)ceil_div((long long)nw, 256)), dim3(256), 0, s, qw, h->wino_groups, h->nchunks);
            VS_CHECK_HIP(hipGetLastError());
            h->wino_packed = true;
        }
        q.wp = h->wpw.as<float>();
        q.KT = h->wino_groups;
        q.lo = -h->pad;
        q.dbg = (int)opt(OPT_WINO_DBG);
        // the vector epilogue stores float4 (dilation 1) / float2 runs: same alignment preconditions as the direct engine's
        // (Tout % 4 == 0 checked above covers both)
        const int span_w = 3 * h->wino_groups * h->dil;
        const int spec = h->wino_k7 ? 7 : (h->wino_k11 ? 11 : 0);
        if (h->dil == 1) return launch_wino_dil<1>(q, h->MT, span_w, spec, s);
        if (h->dil == 3) return launch_wino_dil<3>(q, h->MT, span_w, spec, s);
        return launch_wino_dil<5>(q, h->MT, span_w, spec, s);
    }
    // tile shape: 128-row blocks unless that would leave a half-empty M block (6 tiles = 192 rows: the q/k/v/o, FFN-out,
    // coupling `pre` and last res/skip convs) AND the launch is short (T_mel-sized): there 64 x 256 blocks waste no MFMA
    // rows and give 1.5x the workgroups (a 256-workgroup launch fills only one slot per CU)
    int cfg;   // 0: <1,8,4,1> 128x256   1: <1,8,2,2> 64x512   2: <1,4,1,4> 32x512   3: <1,4,2,2> 64x256
    // (6 tiles on a LONG launch -- the stride-3 stage of the hop-300 generator, 192 virtual rows: 64 x 512 blocks, three in M)
    if (h->MT >= 3) cfg = ((h->MT % 4) == 2) ? (((long long)p.N * p.B <= 65536) ? 3 : 1) : 0;
    else cfg = (h->MT == 2) ? 1 : 2;
    // Short launches (single utterances, T_mel-sized tensors): a 128 x 256 tile grid of fewer workgroups than CUs leaves most of
    // the chip idle while each workgroup runs its full K loop -- the launch lasts as long as ONE workgroup.  On the bf16-pipe
    // engine take 64-row (then 32-row) tiles until the grid covers the CUs: the same kernel family, 2x / 4x the workgroups, each
    // with half / a quarter of the MFMAs per wave (B=1, T_mel=1024 synthesis latency: DESIGN.md 4.2).
    if (h->math && !opt(OPT_NO_SMALL_GRID) && !p.x_bf16 && !p.y_bf16) {      // (bf16-resident tensors: the 128-row instance only)
        const long long ncol = ceil_div(p.N, 256) * p.B;
        if (cfg == 0 && ncol * ceil_div(h->MT, 4) < 256) cfg = 3;
        if (cfg == 3 && ncol * ceil_div(h->MT, 2) < 256 && (long long)p.N * p.B <= 65536) cfg = 2;
        const long long t6 = opt(OPT_SMALL_GRID_T6);
        if (cfg == 2 && ncol * h->MT < t6 && h->kind != VS_CONV_TRANSPOSE1D) cfg = 6;     // 128-column tiles: twice the workgroups again
    }
    if (opt(OPT_CONV_CFG) >= 0 && h->MT >= 3) cfg = (opt(OPT_CONV_CFG) == 3) ? 3 : (opt(OPT_CONV_CFG) == 1 ? 1 : 0);   // A/B switch
    if ((opt(OPT_CONV_CFG) == 2 || opt(OPT_CONV_CFG) == 6) && !p.x_bf16 && !p.y_bf16 && h->kind != VS_CONV_TRANSPOSE1D) cfg = (int)opt(OPT_CONV_CFG);   // (tools/ktap_tile_sweep.py)
    if (h->math) p.wp = h->ws.as<float>();
    if (h->math) p.dbg = (int)opt(OPT_SPLIT_DBG);
    p.wscale = h->wsc.as<float>();
    // the wide stride-1 convs of the split-f16 arithmetic (generator resblocks at 128 / 256 channels, conv_pre, FFN conv_1): taps unrolled, the staging of
    // the next chunk dealt out over the MFMA gaps of the current one (conv_ktap.inc; bit-identical to the tile kernels, VS_NO_KTAP=1: A/B); likewise the
    // plain-bf16 arithmetic (BASELINE configs[4]: the hidden-512 transformer convs on fp32 tensors, the generator's wide convs on bf16-resident tensors)
    // ... and the 64 x 256 / 32 x 128 tiles of short launches (T_mel-sized tensors, the training step): conv_ktap_small.hip
    // (plain bf16 behind a MASKED input transform: a launch whose chosen tile shape has no conv_ktap instance takes another shape that has one before it falls back to
    //  the tile kernel -- round 5 did this for safety, the tile kernel's masked instances gave wrong lanes then; round 6 root-caused and fixed that, DESIGN.md 4.5,
    //  tests/test_conv_mask_race_gpu.py, and the preference stays because the conv_ktap instances are faster)
    if (h->math == VS_MATH_BF16 && p.in_act >= VS_IN_MASK && h->kind == VS_CONV1D && !opt(OPT_NO_KTAP) && p.Cin % CK == 0 && !p.x_bf16 && !p.y_bf16 &&
        !ktap_instance(h->math, cfg, h->KT, 0, p.in_act)) {
        for (int alt : {3, 6, 0})
            if (ktap_instance(h->math, alt, h->KT, 0, p.in_act)) { cfg = alt; break; }
    }
    if ((h->math == VS_MATH_SPLIT3 || h->math == VS_MATH_BF16) && h->kind == VS_CONV1D && !opt(OPT_NO_KTAP) && p.Cin % CK == 0 &&
        ktap_instance(h->math, cfg, h->KT, (p.x_bf16 ? 1 : 0) | (p.y_bf16 ? 2 : 0), p.in_act) && !(p.split_row && (p.split_row % 32) != 0)) {
        p.row_lo = 0;
        p.row_hi = h->c_out;
        return h->math == VS_MATH_SPLIT3 ? launch_ktap(p, cfg, s) : launch_ktap_bf16(p, cfg, s);
    }
    if (h->kind == VS_CONV_TRANSPOSE1D && !opt(OPT_NO_KTAP) && !opt(OPT_NO_TR_EPI) && ktap_tr_instance(p, h->math, cfg)) {      // the generator's upsamplers (k = 2 * stride): conv_ktap.inc, IO bit 2
        p.row_lo = 0;
        p.row_hi = h->c_out;
        return launch_ktap_tr(p, s);
    }
    auto launch = [&](const ConvParams &q) -> int {
        if (h->math) return launch_split(q, cfg, h->math, h->span, s);
        switch (cfg) {
            case 0: return launch_cfg<1, 8, 4, 1>(q, s);
            case 1: return launch_cfg<1, 8, 2, 2>(q, s);
            case 3: return launch_cfg<1, 4, 2, 2>(q, s);
            default: return launch_cfg<1, 4, 1, 4>(q, s);
        }
    };
    p.W = ((cfg == 0 || cfg == 3) ? 256 : 512) + h->span;
    p.row_lo = 0;
    p.row_hi = h->c_out;
    if (p.split_row && (p.split_row % 32) != 0) {
        // a 32-row tile would straddle the two destinations: store them in two passes (odd sizes only; the
        // production split is hidden_channels = 192 = 6 tiles)
        ConvParams q = p;
        q.fast_epi = 0;
        q.split_row = 0;
        q.row_hi = p.split_row;
        VS_TRY(launch(q));
        q.out[0] = p.out[1];
        q.row_lo = p.split_row;
        q.row_hi = h->c_out;
        // rows are addressed relative to split_row in the second destination
        q.out[0].rows = h->c_out;
        q.out[0].y -= (long long)p.split_row * p.Tout;
        if (q.out[0].res) q.out[0].res -= (long long)p.split_row * p.Tout;
        if (q.out[0].acc) q.out[0].acc -= (long long)p.split_row * p.Tout;
        return launch(q);
    }
    return launch(p);
}

}  // extern "C"
