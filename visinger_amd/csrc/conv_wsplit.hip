// conv_wsplit.hip -- minimal filtering F(2,3) ON the split-bf16 x6 arithmetic: the stride-1 "same" convolutions with >= 3 taps and
// whole 128-row blocks (the HiFi-GAN resblock convs of the 128- and 256-channel stages, decoder.py:72-87, and the FFN k = 9 convs,
// rel_transformer.py:332-333) with 4 matrix products per pair of outputs and group of three taps instead of 6, each of them the
// six bf16 cross products of conv_split.hip.
//
//   taps in groups of three (the last group zero-padded: k = 3, 9: 4/6 of the direct matrix work, k = 11: 16/22, k = 7: 12/14);
//   for a pair of outputs (y[t], y[t+d]), x_j = x[t + (3g + j) d - pad]:
//       V0 = x0 - x2, V1 = x1 + x2, V2 = x2 - x1, V3 = x1 - x3
//       U0 = w0, U1 = (w0 + w1 + w2)/2, U2 = (w0 - w1 + w2)/2, U3 = w2          (fp32, once per weight version, then split exactly)
//       M_xi += U_xi . V_xi over input channels and groups;  y[t] = M0 + M1 + M2,  y[t+d] = M1 - M2 - M3
//   The transforms use the points {0, 1, -1, inf}: coefficients 1 and 1/2; V is formed in fp32 from the staged activations and
//   THEN split exactly into three bf16 planes (the split is not linear: planes of a difference are not differences of planes).
//
// What follows from the split for the structure (conv_wino_kernel forms V per wave from shifted fp32 LDS reads; here a B fragment
// must be ONE ds_read_b128 of eight bf16 channels of a plane):
//   * three transformed arrays live in LDS, over all window positions n:  A[n] = x[n] - x[n+2d]  (V0 at n, V3 at n+d),
//     P[n] = x[n+d] + x[n+2d]  (V1),  Q[n] = x[n+2d] - x[n+d]  (V2), each as [plane(3)][k-group(2)][n][8 channels bf16] --
//     9 plane arrays instead of 3: 54 KB for a 128-output tile, so the tile is SINGLE-buffered (two barriers per 16-channel
//     chunk) and two workgroups per CU overlap one's staging with the other's MFMAs;
//   * a wave stages 4 channels: raw (activated, masked) fp32 values go to a wave-private LDS strip, come back shifted by d and
//     2d, are combined and split -- no barrier inside the transform;
//   * a wave owns 32 rows x 64 pair columns x 4 xi = 8 accumulator tiles (128 registers), so a weight fragment feeds two column
//     tiles only: 3 x 16 B per 12 MFMAs from L2, four times the direct engine's rate -- requested three sub-steps ahead through a
//     4-slot register ring (inline asm, hand-counted vmcnt);
//   * one sub-step = (chunk, group, xi): 2 column tiles x 6 cross products; the body of a chunk is straight-line code (G is a
//     template argument).
// Epilogue: the output transform in registers, then the LDS-transposed vector epilogue of conv_wino_kernel (pairs interleaved on
// the way into the transposition buffer).
#include "conv_common.h"

#include <algorithm>
#include <type_traits>

namespace vs {

struct WsplitPack {
    const float *wp;       // fp32 fragment-order weights Wp[m_tile][tap][chunk][quad(2)][64][4] (pack_conv_kernel: weight norm folded in)
    void *ws;              // Us[m_tile][chunk][group][xi][plane(3)][64 lanes][8 bf16]
    int KT, MT_alloc, nchunks, G;
};

// lane l <-> row m_tile*32 + (l & 31), channels chunk*16 + 8*(l >> 5) + j.  Reads the handle's own fp32 fragments (lane l of quad qd,
// element e <-> row l&31, channel chunk*16 + 2*(4*qd + e) + (l>>5)), so the transform can be (re)built whenever it is first needed:
// after vs_conv_set_weights, after a change of arithmetic -- and never for handles whose launches stay on the direct kernel (the
// training path re-packs every conv every step).
__global__ void pack_wsplit_kernel(const WsplitPack q) {
    const long long total = (long long)q.MT_alloc * q.nchunks * q.G * 4 * 64;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int lane = (int)(e & 63);
    const int xi = (int)((e >> 6) & 3);
    long long t = e >> 8;
    const int g = (int)(t % q.G);
    t /= q.G;
    const int chunk = (int)(t % q.nchunks);
    const int mt = (int)(t / q.nchunks);
    const int row = lane & 31;
    float u[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cl = 8 * (lane >> 5) + j, cp = cl >> 1, par = cl & 1;
        float w[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int tap = 3 * g + i;
            w[i] = (tap < q.KT) ? q.wp[(((long long)mt * q.KT + tap) * q.nchunks + chunk) * 512 + (cp >> 2) * 256 + (row + 32 * par) * 4 + (cp & 3)] : 0.f;
        }
        u[j] = (xi == 0) ? w[0] : (xi == 1) ? 0.5f * (w[0] + w[1] + w[2]) : (xi == 2) ? 0.5f * (w[0] - w[1] + w[2]) : w[2];
    }
    unsigned d[4][3];
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) split_pair<3>(u[2 * tq], u[2 * tq + 1], d[tq]);
    u32x4 *dst = reinterpret_cast<u32x4 *>(q.ws) + (e >> 6) * (3 * 64) + lane;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        u32x4 o; o.x = d[0][pl]; o.y = d[1][pl]; o.z = d[2][pl]; o.w = d[3][pl];
        dst[pl * 64] = o;
    }
}

int pack_wsplit(const float *wp, void *ws, int KT, int MT_alloc, int nchunks, int G, hipStream_t s) {
    WsplitPack q;
    q.wp = wp; q.ws = ws; q.KT = KT; q.MT_alloc = MT_alloc; q.nchunks = nchunks; q.G = G;
    const long long total = (long long)MT_alloc * nchunks * G * 4 * 64;
    hipLaunchKernelGGL(pack_wsplit_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, q);
    VS_CHECK_HIP(hipGetLastError());
    return VS_OK;
}

size_t wsplit_bytes(int MT_alloc, int nchunks, int G) { return (size_t)MT_alloc * nchunks * G * 4 * 3 * 64 * 16; }

constexpr int WS_MAXWX = 128 + MAX_SPAN;      // staged raw window: outputs of the tile + 3 G d  (<= 192)
constexpr int WS_CIT = WS_MAXWX / 64;         // column iterations per staged row
constexpr int WS_WR = WS_MAXWX + 4;           // pitch of the wave-private raw strip (floats)

// WN = 1: four waves stacked in M (128 rows), each with both 32-lane pair tiles of the workgroup's columns; WN = 2: 2 x 2 waves (64
// rows: the 64-channel stage), each wave with ONE pair tile -- a weight fragment then feeds a single column tile.
template <int DIL, int G, int WN>
__global__ void __launch_bounds__(256, 2) conv_wsplit_kernel(const ConvParams p) {
    constexpr int NTW = 2 / WN;                  // pair tiles per wave
    constexpr int WM = 4 / WN;                   // waves along M
    constexpr int PW = (WN == 1) ? (64 / DIL) * DIL : (32 / DIL) * DIL;     // valid pair columns per wave
    constexpr int NBW = 2 * PW;                  // outputs per wave
    constexpr int BN = NBW * WN;                 // outputs per workgroup
    constexpr int W = BN + 3 * (G - 1) * DIL + DIL;   // positions of the transformed arrays that are read
    constexpr int WX = W + 2 * DIL;              // raw window = BN + 3 G d
    constexpr int CIT = (WX + 63) / 64;
    constexpr int CWP = NBW + 8;                 // row pitch of the epilogue transposition buffer (+ dump columns)
    constexpr int PLSZ = 2 * W * 4;              // dwords per plane: [k-group][n][4 dwords]
    constexpr int ARSZ = 3 * PLSZ;               // dwords per transformed array
    static_assert(WX <= WS_MAXWX && CIT <= WS_CIT, "window exceeds the staging budget");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z;
    const int n0 = blockIdx.x * BN;
    const int wm = wave % WM, wn = wave / WM;
    const int mt0 = blockIdx.y * WM + wm;
    unsigned *const Vb = reinterpret_cast<unsigned *>(smem);             // [array(3)][plane(3)][k-group(2)][W][4 dwords]
    float *const RAWw = smem + 3 * ARSZ + wave * 4 * WS_WR;              // this wave's [4 channels][WS_WR] raw strip
    const float *const xb = p.x + (long long)b * p.x_bs;
    const float *const maskb = p.mask ? p.mask + (long long)b * p.Tin : nullptr;
    const int lhalf = lane >> 5;
    const int l31 = lane & 31;

    // M0 starts from the row's bias and M3 from its negative (y[t] = M0+M1+M2, y[t+d] = M1-M2-M3)
    const float *const bbias = p.bias_b ? p.bias_b + (long long)b * p.bias_b_bs : nullptr;
    f32x16 acc[4][NTW];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = mt0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        float bv = p.biasp[row];
        if (bbias) bv += bbias[min(row, p.M - 1)];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            acc[0][j][r] = bv;
            acc[1][j][r] = 0.f;
            acc[2][j][r] = 0.f;
            acc[3][j][r] = -bv;
        }
    }

    // pair column c -> first output of the pair t(c) = (c / d) 2d + c % d, relative to the tile's first output
    int posr[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int c = j * 32 + l31;
        const int t = (c / DIL) * (2 * DIL) + (c % DIL);
        posr[j] = wn * NBW + ((c < PW) ? t : 0);          // idle columns read a valid LDS address
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
    const bool time_edge = (n0 + p.lo < 0) || (n0 + p.lo + WX > p.Tin);
    auto stage_store = [&](int chunk) __attribute__((always_inline)) {
        auto run = [&](auto edge_tag, auto act_tag) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            constexpr int ACT = decltype(act_tag)::value;
            // 1. activated / masked fp32 values -> the wave's raw strip (columns beyond the window are never read back)
#pragma unroll
            for (int i = 0; i < CIT; ++i) {
                const int col = lane + 64 * i;
                const int n = n0 + p.lo + col;
                const bool okn = (n >= 0) && (n < p.Tin);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = st[j][i];
                    if constexpr (EDGE) v = (okn && (chunk * CK + 4 * wave + j < p.Cin)) ? v : 0.f;
                    if constexpr (ACT == VS_IN_LRELU || ACT == VS_IN_LRELU_MASK) v = fmaxf(v, 0.1f * v);
                    if constexpr (ACT >= VS_IN_MASK) v *= mk[i];
                    st[j][i] = v;
                    if (64 * (i + 1) <= WX || col < WX) RAWw[j * WS_WR + col] = v;
                }
            }
            // the strip is private to the wave: LDS operations of one wave execute in order, no barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // 2. x[n + d], x[n + 2d] back from the strip; A = x0 - x2, P = x1 + x2, Q = x2 - x1; exact split; one ds_write_b64 per plane
            unsigned *const dst0 = Vb + ((wave >> 1) * W + lane) * 4 + (wave & 1) * 2;
#pragma unroll
            for (int i = 0; i < CIT; ++i) {
                const int col = lane + 64 * i;
                if (64 * (i + 1) <= W || col < W) {
                    float a[4], pp[4], qq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x0 = st[j][i];
                        const float x1 = RAWw[j * WS_WR + col + DIL];
                        const float x2 = RAWw[j * WS_WR + col + 2 * DIL];
                        a[j] = x0 - x2;
                        pp[j] = x1 + x2;
                        qq[j] = x2 - x1;
                    }
                    unsigned d0[3], d1[3];
                    split_pair<3>(a[0], a[1], d0);
                    split_pair<3>(a[2], a[3], d1);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2 *>(dst0 + pl * PLSZ + i * 256) = make_uint2(d0[pl], d1[pl]);
                    split_pair<3>(pp[0], pp[1], d0);
                    split_pair<3>(pp[2], pp[3], d1);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2 *>(dst0 + ARSZ + pl * PLSZ + i * 256) = make_uint2(d0[pl], d1[pl]);
                    split_pair<3>(qq[0], qq[1], d0);
                    split_pair<3>(qq[2], qq[3], d1);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2 *>(dst0 + 2 * ARSZ + pl * PLSZ + i * 256) = make_uint2(d0[pl], d1[pl]);
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

    // ---------------------------------------------------------------------------------------------- main loop
    // sub-step ss = (chunk * G + g) * 4 + xi: three 16-byte weight fragments (planes h, m, l of U_xi), 2 column tiles x 6 MFMAs.
    // Fragments are requested RING - 1 = 3 sub-steps ahead; 4 G sub-steps per chunk are a multiple of the ring period, so the
    // body of a chunk is straight-line code with compile-time ring slots.
    //
    // The weight loads are inline asm with HAND-COUNTED vmcnt waits (as in conv_split_kernel: left to hipcc the wait in front of a
    // sub-step's first MFMA is vmcnt(0) or vmcnt(1), i.e. it also waits for the fragments requested a few instructions earlier for
    // sub-step ss + 3 -- one exposed L2 round trip per sub-step).  vmcnt counts vector-memory operations in issue order, so "the
    // fragments of THIS sub-step have landed" = at most (everything issued after them) outstanding: the three younger requests (9
    // loads; the request is unconditional, clamped to the last sub-step, so the count is a constant) plus, in the first four
    // sub-steps of a chunk, the activation loads of chunk + 2, which are issued right AFTER the request of sub-step 0 so that they
    // stay younger than the fragments of sub-steps 0 .. 3 (csrc/build.py fails the build if this kernel uses scratch: a spill
    // would be a vector-memory instruction behind the count's back).
    constexpr int SPC = 4 * G;                                   // sub-steps per chunk
    const int nss = p.nchunks * SPC;
    const u32x4 *const wbase = reinterpret_cast<const u32x4 *>(p.wp) + (long long)mt0 * nss * (3 * 64) + lane;
    u32x4 ar[4][3];
    auto load_a = [&](u32x4 (&dst)[3], int ss) __attribute__((always_inline)) {
        const u32x4 *src = wbase + (long long)min(ss, nss - 1) * (3 * 64);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst[0]) : "v"(src) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=&v"(dst[1]) : "v"(src) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=&v"(dst[2]) : "v"(src) : "memory");
    };
    const int NY = ((in_act >= VS_IN_MASK) ? 5 : 4) * CIT;       // activation (+ mask) loads of one stage_load
    auto wait_a = [&](u32x4 (&a)[3], bool staged) __attribute__((always_inline)) {
        if (staged) {
            if (NY == 4 * CIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(9 + 4 * CIT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(9 + 5 * CIT) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(a[pl]));
    };
    load_a(ar[0], 0);
    load_a(ar[1], 1);
    load_a(ar[2], 2);

    stamp(p, 0);
    // Two workgroups share a CU and each alternates a VALU / LDS phase (transform) with an MFMA phase per chunk: started together
    // they transform together and then compete for the matrix pipe.  p.dbg > 0 (VS_WSPLIT_STAGGER, experiment): the workgroups of
    // every other dispatch round of 256 start p.dbg x 64 cycles late.
    if (p.dbg > 0 && (((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) >> 8) & 1)) {
        for (int i = 0; i < p.dbg; i += 32) __builtin_amdgcn_s_sleep(32);
    }
    stage_load(0);
    stage_store(0);                 // (hipcc waits vmcnt(0) for the staged registers: covers the three requests above as well)
    __syncthreads();
    stamp(p, 1);

    // B fragments of (sub-step u, tile j): V0 = A[n], V1 = P[n], V2 = Q[n], V3 = A[n + d] with n = t(c) + 3 g d
    auto read_b = [&](u32x4 (&bf)[3], int u, int j) __attribute__((always_inline)) {
        const int g = u >> 2, xi = u & 3;
        const int arr = (xi == 1) ? 1 : (xi == 2) ? 2 : 0;
        const int shift = 3 * g * DIL + ((xi == 3) ? DIL : 0);
        const unsigned *xs = Vb + arr * ARSZ + (lhalf * W + posr[j] + shift) * 4;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bf[pl] = *reinterpret_cast<const u32x4 *>(xs + pl * PLSZ);
    };
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const bool stage_next = chunk + 1 < p.nchunks;
        u32x4 b0[3], b1[3];
        read_b(b0, 0, 0);
#pragma unroll
        for (int u = 0; u < SPC; ++u) {
            const int xi = u & 3;
            const int ss = chunk * SPC + u;
            load_a(ar[(u + 3) & 3], ss + 3);
            if (u == 0 && stage_next) stage_load(chunk + 1);          // (registers free: chunk's own data went to LDS before the barrier)
            if constexpr (NTW == 2) read_b(b1, u, 1);
            wait_a(ar[u & 3], stage_next && u < 4);
            const u32x4(&a)[3] = ar[u & 3];
            auto mm = [&](f32x16 &c, const u32x4(&bf)[3], int ta, int tb) __attribute__((always_inline)) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ta]), __builtin_bit_cast(bf16x8, bf[tb]), c, 0, 0, 0);
            };
            if constexpr (NTW == 2) {
                __builtin_amdgcn_sched_barrier(0);
                // smallest terms first
                mm(acc[xi][0], b0, 1, 1); mm(acc[xi][0], b0, 2, 0); mm(acc[xi][0], b0, 0, 2);
                mm(acc[xi][0], b0, 1, 0); mm(acc[xi][0], b0, 0, 1); mm(acc[xi][0], b0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (u + 1 < SPC) read_b(b0, u + 1, 0);                // the next sub-step's first tile, under this one's second
                __builtin_amdgcn_sched_barrier(0);
                mm(acc[xi][NTW - 1], b1, 1, 1); mm(acc[xi][NTW - 1], b1, 2, 0); mm(acc[xi][NTW - 1], b1, 0, 2);
                mm(acc[xi][NTW - 1], b1, 1, 0); mm(acc[xi][NTW - 1], b1, 0, 1); mm(acc[xi][NTW - 1], b1, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // one pair tile per wave: the planes of the next sub-step are read under this one's MFMAs (b0 / b1 alternate)
                u32x4(&bc)[3] = (u & 1) ? b1 : b0;
                u32x4(&bx)[3] = (u & 1) ? b0 : b1;
                if (u + 1 < SPC) read_b(bx, u + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                mm(acc[xi][0], bc, 1, 1); mm(acc[xi][0], bc, 2, 0); mm(acc[xi][0], bc, 0, 2);
                mm(acc[xi][0], bc, 1, 0); mm(acc[xi][0], bc, 0, 1); mm(acc[xi][0], bc, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (chunk == 1) stamp(p, 8);
        __syncthreads();                                  // every wave is done with the transformed arrays of this chunk
        if (stage_next) {
            stage_store(chunk + 1);
            if (chunk == 1) stamp(p, 9);
            __syncthreads();
            if (chunk == 1) stamp(p, 10);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of the asm loads may be in flight past here

    stamp(p, 2);
    // ------------------------------------------------------------------------------------------------- epilogue
    // output transform in place: acc[0] <- y[t(c)], acc[3] <- y[t(c) + d]
#pragma unroll
    for (int j = 0; j < NTW; ++j)
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
    int lane_e = lane;                           // opaque copy: keeps the epilogue's address arithmetic out of the prologue
    asm volatile("" : "+v"(lane_e));
    int posw[NTW], pose[NTW];
    bool cvalid[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
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
        // lanes beyond LPR * RPI (dilation 3 / 5: 63 or 60 of 64) repeat the last lane's loads and stores (identical values to identical
        // addresses) instead of being masked off: with 64-row workgroups they hold VALID pair columns of the other lane half, and
        // their transposition writes of the first pass of a batch were lost when the load batch in front of it was predicated
        const int lane_a = min(lane_e, LPR * RPI - 1);
        const int lrow = lane_a / LPR;
        const int cv = (lane_a % LPR) * VEC;
        constexpr bool active = true;
        const int colg = nw + cv;
        vecf mv;
#pragma unroll
        for (int e = 0; e < VEC; ++e) mv[e] = 1.f;
        if (use_mask && active) mv = *reinterpret_cast<const vecf *>(maskb + colg);
        // (as conv_wino_kernel: the residual / accumulate reads of two 8-row passes in flight at once; loads of a batch precede its
        // stores and every lane stores exactly the elements it loaded, so y may alias res / acc)
        auto run = [&](auto simple_tag, auto res_tag) __attribute__((always_inline)) {
            constexpr int PB = 2;
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
                        for (int j = 0; j < NTW; ++j) {
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
            if (has_res) run(std::true_type{}, std::true_type{});
            else run(std::true_type{}, std::false_type{});
        } else if (has_res) {
            run(std::false_type{}, std::true_type{});
        } else {
            run(std::false_type{}, std::false_type{});
        }
    } else {
        // edge workgroups (ragged last time tile): element-wise, predicated stores, clamped loads
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
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
        __builtin_amdgcn_s_waitcnt(0);
        stamp(p, 3);
    }
}

template <int DIL, int G, int WN>
static int launch_wsplit_cfg(const ConvParams &p, hipStream_t s) {
    constexpr int PW = (WN == 1) ? (64 / DIL) * DIL : (32 / DIL) * DIL, NBW = 2 * PW, BN = NBW * WN;
    constexpr int W = BN + 3 * (G - 1) * DIL + DIL, CWP = NBW + 8;
    auto kern = conv_wsplit_kernel<DIL, G, WN>;
    const size_t lds = std::max<size_t>((size_t)4 * (3 * 3 * 2 * W * 4 + 4 * 4 * WS_WR), (size_t)4 * 4 * 8 * CWP);
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)ceil_div(p.N, BN), (unsigned)ceil_div(p.MT, 4 / WN), (unsigned)p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    set_last_kernel("conv_wsplit_kernel<%d, %d, %d>", DIL, G, WN);
    return VS_OK;
}

bool wsplit_instance(int dil, int G) { return (dil == 1 || dil == 3 || dil == 5) && (G == 1 || G == 3 || G == 4) && 3 * G * dil <= MAX_SPAN; }

// MT % 4 == 0: 128-row workgroups (WN = 1); otherwise MT % 2 == 0: 64-row workgroups (WN = 2)
int launch_wsplit(const ConvParams &p, int dil, int G, hipStream_t s) {
    const bool wide = (p.MT % 4) == 0;
#define VS_WS(D, GG) if (dil == D && G == GG) return wide ? launch_wsplit_cfg<D, GG, 1>(p, s) : launch_wsplit_cfg<D, GG, 2>(p, s)
    VS_WS(1, 1); VS_WS(1, 3); VS_WS(1, 4);
    VS_WS(3, 1); VS_WS(3, 3); VS_WS(3, 4);
    VS_WS(5, 1); VS_WS(5, 3); VS_WS(5, 4);
#undef VS_WS
    set_error("launch_wsplit: no instance for dilation %d, %d tap groups", dil, G);
    return VS_EUNSUPPORTED;
}

}  // namespace vs
