// conv_pipe.hip -- conv_pipe_kernel: the 128 x 256 tile of the split-f16 x3 conv engine (conv_split_kernel<1, 8, 4, 1, 3>) as a PERSISTENT,
// SOFTWARE-PIPELINED kernel: one workgroup per CU walks a sequence of column tiles, and the epilogue of tile i (accumulators -> LDS
// transposition -> + residual [+ MRF accumulator] -> 16-byte stores) is executed in slices INSIDE the main loop of tile i + 1.
//
// Why: the per-launch time of conv_split_kernel<1, 8, 4, 1, 3> is t_bytes + k * t_tap with BOTH terms at their
// own ceilings (the three tensor passes at ~6 TB/s, the tap slope at 95 % of a bare MFMA loop) -- the memory phases of a tile (3 us of
// exposed prologue round trips, an 18-26 us epilogue) never overlap matrix work: a wave is either in its main loop or in its epilogue, two
// co-resident workgroups share the matrix pipe while both are in their main loops (tools/conv_stamps.py: 69 us main loops at 128 channels,
// k = 7, against 27.5 us of MFMA issue) and idle it together in their epilogues.  Here no wave ever has a memory-only phase:
//   * every wave keeps TWO accumulator sets: `acc` (the tile being computed) and `prev` (the finished tile, copied at the tile boundary);
//   * the 16 row-pair items of the finished tile (one ds_read_b128 + residual / accumulate loads + one 16-byte store each) are issued one
//     every `istep` steps of the next tile's (chunk, tap) loop, their loads two items ahead: the HBM traffic of the epilogue is spread
//     evenly over the whole tile time instead of arriving as a burst that every CU sends at once;
//   * the staging stream (activations of chunk c + 2 requested, chunk c + 1 split into f16 planes) runs across tile boundaries, so a tile
//     has no prologue: the first chunks of tile i + 1 are requested during the last chunks of tile i.
// Eight waves (4 row tiles x 2 column halves, 32 x 128 outputs = 64 accumulators per wave and set) so that two waves share a SIMD and
// fill each other's bubbles (LDS latency, barriers, scalar bookkeeping) within the 256 registers a wave then owns.
//
// ALL vector-memory instructions of the loop are inline asm and waited for by COUNT: a wave numbers its VMEM operations (`vm_seq`, a
// scalar), remembers the number of the last operation of every outstanding load group, and waits with `s_waitcnt vmcnt(vm_seq - seq)`
// (vmcnt counts loads and stores together, in issue order: MI355X_MICROARCH.md "s_waitcnt vmcnt(N)"); the immediate comes from a
// computed jump into a table of 64 s_waitcnt instructions.  hipcc sees no vector-memory instruction in the loop, so it inserts no wait of
// its own (its vmcnt(0) in front of a conditional block would drain the whole pipeline once per chunk).
//
// Arithmetic, staging layout, weight-fragment layout and epilogue arithmetic are those of conv_split_body.inc / conv_epilogue.inc
// (fast path), operation for operation: the outputs are BIT-IDENTICAL to conv_split_kernel<1, 8, 4, 1, 3> (tests/test_conv_pipe_gpu.py).
// Reference work being replaced: modules/visinger/decoder.py:91-104 (ResBlock1 convs at 128 / 256 channels).
#include "conv_common.h"

#include <algorithm>
#include <type_traits>
#include <utility>

namespace vs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// wait until at most n (wave-uniform, clamped to 0..63) vector-memory operations of this wave are outstanding
__device__ __forceinline__ void wait_vm(int n) {
    n = __builtin_amdgcn_readfirstlane(n);
    asm volatile(
        "s_min_u32 %0, %0, 63\n\t"
        "s_lshl_b32 s92, %0, 3\n\t"
        "s_getpc_b64 s[90:91]\n\t"
        "s_add_u32 s90, s90, s92\n\t"
        "s_addc_u32 s91, s91, 0\n\t"
        "s_add_u32 s90, s90, 20\n\t"       // bytes from the instruction after s_getpc_b64 to the table (4 x 4-byte instructions + s_setpc)
        "s_addc_u32 s91, s91, 0\n\t"
        "s_setpc_b64 s[90:91]\n\t"
#define VS_WE(k) "s_waitcnt vmcnt(" #k ")\n\ts_branch 1f\n\t"
        VS_WE(0) VS_WE(1) VS_WE(2) VS_WE(3) VS_WE(4) VS_WE(5) VS_WE(6) VS_WE(7) VS_WE(8) VS_WE(9) VS_WE(10) VS_WE(11) VS_WE(12) VS_WE(13)
        VS_WE(14) VS_WE(15) VS_WE(16) VS_WE(17) VS_WE(18) VS_WE(19) VS_WE(20) VS_WE(21) VS_WE(22) VS_WE(23) VS_WE(24) VS_WE(25) VS_WE(26)
        VS_WE(27) VS_WE(28) VS_WE(29) VS_WE(30) VS_WE(31) VS_WE(32) VS_WE(33) VS_WE(34) VS_WE(35) VS_WE(36) VS_WE(37) VS_WE(38) VS_WE(39)
        VS_WE(40) VS_WE(41) VS_WE(42) VS_WE(43) VS_WE(44) VS_WE(45) VS_WE(46) VS_WE(47) VS_WE(48) VS_WE(49) VS_WE(50) VS_WE(51) VS_WE(52)
        VS_WE(53) VS_WE(54) VS_WE(55) VS_WE(56) VS_WE(57) VS_WE(58) VS_WE(59) VS_WE(60) VS_WE(61) VS_WE(62) VS_WE(63)
#undef VS_WE
        "1:\n\t"
        : "+s"(n) : : "s90", "s91", "s92", "scc", "memory");
}

__device__ __forceinline__ const char *uniform_ptr(const void *q) {
    const unsigned long long u = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
}

// Timing-only perturbations (results are garbage): build with -DVS_PIPE_PERTURB and set VS_SPLIT_DBG to a sum of 1 = no epilogue events,
// 2 = no staging (no activation loads, no split + LDS writes, no exponent exchange), 4 = no wait for the weight fragments, 8 = no
// weight-fragment loads after the first, 16 = no B-fragment reads.  Compiled out by default.
#ifdef VS_PIPE_PERTURB
#define PIPE_PERTURB(bit) ((p.dbg & (bit)) != 0)
// shader-clock stamps of the first 100 steps of workgroup (0, 0), waves 0 and 4 (one SIMD): slot [wave >> 2][step][k]
#define PSTAMP(k)                                                                                                         \
    do {                                                                                                                  \
        if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && (wave & 3) == 0 && s < 100) {                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                   \
            if (lane == 0) p.stamps[(size_t)(wave >> 2) * 1024 + s * 8 + (k)] = t_;                                       \
        }                                                                                                                 \
    } while (0)
#else
#define PSTAMP(k)
#define PIPE_PERTURB(bit) false
#endif

constexpr int PIPE_BN = 256;             // columns of a workgroup tile
constexpr int PIPE_CW = 128;             // columns of a wave's tile
constexpr int PIPE_NTW = 4;              // 32-column accumulator tiles per wave
constexpr int PIPE_EVENTS = 18;          // epilogue events per tile: 16 items + 2 events of load lead

// wait until the `n` youngest vector-memory operations are the only ones outstanding; the common count of the main loop (2: the two
// fragment loads just issued for the next tap) takes a compare and a fixed immediate instead of the computed jump
__device__ __forceinline__ void wait_vm_fast2(int n) {
    n = __builtin_amdgcn_readfirstlane(n);
    if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else wait_vm(n);
}

// index k of the chunk's epilogue event that sits at tap T (event k at tap k * KT / 3 + shift, taken modulo KT), or -1
constexpr int pipe_event_at(int T, int KT, int shift) {
    for (int k = 0; k < 3; ++k)
        if (((k * KT) / 3 + shift) % KT == T) return k;
    return -1;
}

template <class F, int... Is>
__device__ __forceinline__ void for_each_tap(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}

template <int KT, bool HAS_ACC>
__global__ void __launch_bounds__(512) conv_pipe_kernel(const ConvParams p) {
    constexpr int NPL = 2;
    constexpr int NT_W = PIPE_NTW, CW = PIPE_CW;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3;                 // row tile of the workgroup's 128 rows
    const int wn = wave >> 2;                // column half
    const int sch = wave >> 1;               // staging: channels 4 * sch .. 4 * sch + 3 of every chunk ...
    const int spar = wave & 1;               // ... on the column iterations of this parity (even waves: 0, 2, 4; odd: 1, 3)
    const int mt0 = blockIdx.y * 4 + wm;
    const int W = p.W;
    const int PLSZ = 2 * W * 4;              // dwords per plane: [k-group][column][4 dwords = 8 f16]
    unsigned *const lbuf0 = reinterpret_cast<unsigned *>(smem);
    unsigned *const lbuf1 = lbuf0 + NPL * PLSZ;
    int *const smax = reinterpret_cast<int *>(lbuf1 + NPL * PLSZ);       // [2][8] biased exponents of the waves' staged maxima
    float *const sbias = reinterpret_cast<float *>(smax + 16);            // [128] bias of the workgroup's rows
    float *const Lw0 = sbias + 128 + wave * 2048;                         // this wave's two 8-row x 128-column transposition regions
    float *const rdma = sbias + 128 + 8 * 2048 + wave * 768;              // this wave's landing slots of the epilogue's loads: res x 2, acc (1 KB each)
    const unsigned rdma_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)rdma);
    const int lhalf = lane >> 5, l31 = lane & 31;

    // ---- this workgroup's tile sequence: tiles blockIdx.x, blockIdx.x + gridDim.x, ... of ncol * B column tiles (item-major)
    const int ncol = p.N >> 8;
    const int ntiles = ncol * p.B;
    const int ntl = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nchunks = p.nchunks;
    const int G = ntl * nchunks;                                 // chunks this workgroup consumes
    const int epc = (PIPE_EVENTS + nchunks - 1) / nchunks;       // epilogue events per chunk (<= 3: the host requires nchunks >= 6)
    auto coord = [&](int i, int &b, int &n0) __attribute__((always_inline)) {
        const int t = (int)blockIdx.x + i * (int)gridDim.x;
        b = __builtin_amdgcn_readfirstlane(t / ncol);
        n0 = __builtin_amdgcn_readfirstlane((t - b * ncol) << 8);
    };

    if (tid < 128) sbias[tid] = p.biasp[blockIdx.y * 128 + tid];
    float wscale_inv = p.wscale[1];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wscale_inv) : : "memory");      // (hipcc must not sink this load into the loop: its wait there would be a vmcnt(0))
    const OutSpec &o = p.out[0];
    const bool has_res = o.res != nullptr;
    constexpr bool has_acc = HAS_ACC;
    // values only needed at tile boundaries (tensor bases and item strides) are re-read from the kernel-argument segment there (scalar
    // loads) instead of occupying SGPRs through the loop, where spilled scalars cost VGPR lanes
    typedef const __attribute__((address_space(4))) ConvParams *kargs_t;
    const kargs_t kargs = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    const int in_act = p.in_act;

    f32x16 acc[NT_W], prev[NT_W];
#pragma unroll
    for (int j = 0; j < NT_W; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; prev[j][r] = 0.f; }

    // ---- the wave's VMEM scoreboard
    int vm_seq = 0;                 // vector-memory operations issued so far
    int seqX = 0;                   // number of the last staging load in flight
    int seqR0 = 0, seqR1 = 0;       // ... of the last residual load of landing slot 0 / 1
    int seqAcc = 0;                 // ... of the accumulate-input load
    int seqA0 = 0, seqA1 = 0;       // ... of the fragment loads into set 0 / 1

    // ---- staging stream: loads (chunk l_chunk of tile l_i) two chunks ahead of consumption, stores (s_chunk of tile s_i) one ahead
    float st[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) st[j][q] = 0.f;
    int l_i = 0, l_chunk = 0, l_b, l_n0;
    coord(0, l_b, l_n0);
    int s_i = 0, s_chunk = 0, s_n0 = l_n0, s_cnt = 0;
    auto x_rsrc = [&](int b) __attribute__((always_inline)) {
        kargs_t q = kargs;
        asm volatile("" : "+s"(q));
        return __builtin_amdgcn_make_buffer_rsrc((void *)uniform_ptr(q->x + (long long)b * q->x_bs), 0, (int)((long long)p.Cin * p.Tin * 4), 0x00020000);
    };
    __amdgpu_buffer_rsrc_t xsrc = x_rsrc(l_b);
    // (every wave issues 12 loads: the third column iteration of the odd waves lies beyond the staged window and is never used -- a
    //  constant count keeps the scoreboard arithmetic out of the loop)
    auto stage_load = [&]() __attribute__((always_inline)) {
        const int nb4 = (l_n0 + p.lo + lane) * 4 + spar * 256;
        int voff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) voff[j] = min(l_chunk * CK + 4 * sch + j, p.Cin - 1) * p.Tin * 4 + nb4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(st[j][0]) : "v"(voff[j]), "s"(xsrc) : "memory");
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:512" : "=v"(st[j][1]) : "v"(voff[j]), "s"(xsrc) : "memory");
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:1024" : "=v"(st[j][2]) : "v"(voff[j]), "s"(xsrc) : "memory");
        }
        vm_seq += 12;
        seqX = vm_seq;
        if (++l_chunk == nchunks) {
            l_chunk = 0;
            if (++l_i < ntl) { coord(l_i, l_b, l_n0); xsrc = x_rsrc(l_b); }
        }
    };
    // the tile's running scale (conv_split_body.inc): eb_st = biased exponent of the largest magnitude staged so far for the tile in the
    // STAGING stream; `pend` = what the accumulators of the consumed tile are multiplied by at the next chunk boundary; eb_fin = the final
    // exponent of the tile whose staging has ended (taken over by the epilogue at its tile boundary)
    int eb_st = F16_EB_MIN, eb_fin = F16_EB_MIN;
    float pend = 1.f;
    auto stage_store = [&](unsigned *buf) __attribute__((always_inline)) {
        wait_vm(vm_seq - seqX);
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(st[j][0]), "+v"(st[j][1]), "+v"(st[j][2]));
        auto run = [&](auto edge_tag, auto act_tag) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            constexpr bool LRELU = decltype(act_tag)::value;
            unsigned *const dst0 = buf + ((sch >> 1) * W + lane) * 4 + (sch & 1) * 2;
            unsigned mkey = 0u;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q == 2 && spar) break;
                const int col = lane + 64 * (spar + 2 * q);
                const int n = s_n0 + p.lo + col;
                const bool okn = (n >= 0) && (n < p.Tin);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = st[j][q];
                    if constexpr (EDGE) v = (okn && (s_chunk * CK + 4 * sch + j < p.Cin)) ? v : 0.f;
                    if constexpr (LRELU) v = fmaxf(v, 0.1f * v);
                    st[j][q] = v;
                    if (col < W) mkey = f16_maxkey(mkey, v);
                }
            }
            const int ebw = wave_max_u8(f16_key_exponent(mkey));
            int *const slot = smax + (s_cnt & 1) * 8;
            if (lane == 0) slot[wave] = ebw;
            __syncthreads();
            const int4 sa = *reinterpret_cast<const int4 *>(slot), sb = *reinterpret_cast<const int4 *>(slot + 4);
            const int eb = max(max(max(max(sa.x, sa.y), max(sa.z, sa.w)), max(max(sb.x, sb.y), max(sb.z, sb.w))), F16_EB_MIN);
            if (s_chunk == 0) { eb_fin = eb_st; eb_st = eb; pend = 1.f; }
            else if (eb > eb_st) { pend = __builtin_ldexpf(1.f, eb_st - eb); eb_st = eb; }
            const float sx = f16_scale(eb_st);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q == 2 && spar) break;
                const int col = lane + 64 * (spar + 2 * q);
                unsigned d0[2], d1[2];
                split_pair_h(st[0][q] * sx, st[1][q] * sx, d0);
                split_pair_h(st[2][q] * sx, st[3][q] * sx, d1);
                if (col < W) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        *reinterpret_cast<uint2 *>(dst0 + pl * PLSZ + (spar + 2 * q) * 256) = make_uint2(d0[pl], d1[pl]);
                }
            }
        };
        const bool edge = (s_n0 + p.lo < 0) || (s_n0 + p.lo + W > p.Tin) || (s_chunk * CK + CK > p.Cin);
        const bool lrelu = (in_act == VS_IN_LRELU);
        if (edge) {
            if (lrelu) run(std::true_type{}, std::true_type{});
            else run(std::true_type{}, std::false_type{});
        } else {
            if (lrelu) run(std::false_type{}, std::true_type{});
            else run(std::false_type{}, std::false_type{});
        }
        ++s_cnt;
        if (++s_chunk == nchunks) {
            s_chunk = 0;
            if (++s_i < ntl) { int sb_; coord(s_i, sb_, s_n0); }
        }
    };

    // ---- weight fragments: [m_tile][tap][chunk][plane(2)][64 lanes][8 f16], one 16-byte load per plane and tap, one tap ahead, into two
    // NAMED register sets: tap t of a chunk computes from set t & 1.  With an odd KT the last tap of a chunk and the first tap of the next
    // would want the same set: the fragments of a chunk's tap 0 are requested into set 1 and moved to set 0 once they have landed.
    const char *const wsb = uniform_ptr(reinterpret_cast<const char *>(p.wp) + (long long)mt0 * KT * nchunks * (NPL * 64 * 16));
    const int la = lane * 16;
    u32x4 a0[NPL], a1[NPL];
    constexpr int FIRST_SET = KT & 1;          // the set the fragments of a chunk's tap 0 are loaded into
    auto load_a = [&](auto set_tag, int chunk, int tap) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_tag)::value;
        const char *src = wsb + (long long)(tap * nchunks + chunk) * (NPL * 64 * 16);
        if constexpr (SET) {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(a1[0]) : "v"(la), "s"(src) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(a1[1]) : "v"(la), "s"(src) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(a0[0]) : "v"(la), "s"(src) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(a0[1]) : "v"(la), "s"(src) : "memory");
        }
        vm_seq += 2;
        if constexpr (SET) seqA1 = vm_seq; else seqA0 = vm_seq;
    };

    // ---- epilogue of the PREVIOUS tile, in PIPE_EVENTS events: event ev finishes item ev - 2 (LDS transposition of the accumulators 8 rows
    // at a time, + residual [+ accumulate input], one 16-byte store per lane) and requests the residual of item ev (LDS-DMA into one of two
    // wave-private 1 KB slots; the accumulate input of item ev - 1 into a third).  Through registers those loads cost 12 VGPRs and were
    // WRONG: hipcc copies a loop-carried asm destination at control-flow joins before the data has landed (cdna_hip_programming.md 5.7
    // item 1); an LDS landing slot has no register for the compiler to move.
    int ev = PIPE_EVENTS;                                            // (no finished tile yet)
    float acc_inv_prev = 1.f;
    const char *ypv = nullptr, *rpv = nullptr, *apv = nullptr;      // y / res / acc at (item pv_b, row mt0 * 32, column pv_n0 + wn * 128)
    const int lrow = lane >> 5;                                      // 32 lanes per 128-column row: two rows per instruction
    const int c4 = (lane & 31) * 4;
    const int lane_off = (lrow * p.Tout + c4) * 4;                   // bytes
    auto write_pass = [&](auto ps_tag) __attribute__((always_inline)) {
        constexpr int PS = decltype(ps_tag)::value;
        float *const Lw = Lw0 + (PS & 1) * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < NT_W; ++j) Lw[(q + 4 * lhalf) * CW + 32 * j + l31] = prev[j][4 * PS + q];
    };
    auto epi_event = [&]() __attribute__((always_inline)) {
        const int sl = ev & 1;                                       // landing slot of this event's item and of the residual requested here
        const int e = ev - 2;
        if (e >= 0) {
            const int ps = e >> 2, it = e & 3;
            if (it == 0) {
                if (ps == 0) write_pass(std::integral_constant<int, 0>{});
                else if (ps == 1) write_pass(std::integral_constant<int, 1>{});
                else if (ps == 2) write_pass(std::integral_constant<int, 2>{});
                else write_pass(std::integral_constant<int, 3>{});
            }
            if constexpr (HAS_ACC) wait_vm(vm_seq - seqAcc);      // (the accumulate-input load of this item is younger than its residual load: one wait)
            else if (has_res) wait_vm(vm_seq - (sl ? seqR1 : seqR0));
            f32x4 r4 = f32x4{0.f, 0.f, 0.f, 0.f}, a4 = r4;
            if (has_res) r4 = *reinterpret_cast<const f32x4 *>(rdma + sl * 256 + lane * 4);
            if constexpr (HAS_ACC) a4 = *reinterpret_cast<const f32x4 *>(rdma + 512 + lane * 4);
            f32x4 v = *reinterpret_cast<const f32x4 *>(Lw0 + (ps & 1) * 1024 + (it * 2 + lrow) * CW + c4);
            const float bb = sbias[wm * 32 + 8 * ps + 2 * it + lrow];
            v.x = fmaf(v.x, acc_inv_prev, bb); v.y = fmaf(v.y, acc_inv_prev, bb); v.z = fmaf(v.z, acc_inv_prev, bb); v.w = fmaf(v.w, acc_inv_prev, bb);
            if (has_res) { v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
            if (has_acc) { v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w; }
            if (o.scale != 1.f) { v.x *= o.scale; v.y *= o.scale; v.z *= o.scale; v.w *= o.scale; }
            if (o.out_act == VS_OUT_TANH) {
                v.x = tanh_fast(v.x); v.y = tanh_fast(v.y); v.z = tanh_fast(v.z); v.w = tanh_fast(v.w);
            } else if (o.out_act == VS_OUT_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            const char *yp = ypv + (long long)(8 * ps + 2 * it) * p.Tout * 4;
            asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(lane_off), "v"(v), "s"(yp) : "memory");
            vm_seq += 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the slots read above are about to be targeted again)
        const int l = ev;
        if (l < 16 && has_res) {
            const char *rp = rpv + (long long)(8 * (l >> 2) + 2 * (l & 3)) * p.Tout * 4;
            glds16(lane_off, rp, rdma_lds + sl * 1024);
            vm_seq += 1;
            if (sl) seqR1 = vm_seq; else seqR0 = vm_seq;
        }
        if constexpr (HAS_ACC) {
            const int la_ = ev - 1;               // item finished by the NEXT event
            if (la_ >= 0 && la_ < 16) {
                const char *ap = apv + (long long)(8 * (la_ >> 2) + 2 * (la_ & 3)) * p.Tout * 4;
                glds16(lane_off, ap, rdma_lds + 2048);
                vm_seq += 1;
                seqAcc = vm_seq;
            }
        }
        ++ev;
    };

    // ---- consumption stream
    int c_i = 0, c_b = l_b, c_n0 = l_n0;
    int chunk = 0;                                 // chunk of the consumed tile

    auto tile_end = [&]() __attribute__((always_inline)) {
        // (every event of the tile finished before has run: nchunks * epc >= PIPE_EVENTS)
#pragma unroll
        for (int j = 0; j < NT_W; ++j) {
            prev[j] = acc[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
        // the staging stream is one chunk ahead: if a next tile exists its chunk 0 has been stored and took this tile's exponent over
        const int eb_done = (c_i + 1 < ntl) ? eb_fin : eb_st;
        acc_inv_prev = f16_inv_scale(eb_done) * wscale_inv;
        const long long eoff = (long long)mt0 * 32 * p.Tout + c_n0 + wn * CW;
        kargs_t q = kargs;
        asm volatile("" : "+s"(q));
        ypv = uniform_ptr(q->out[0].y + (long long)c_b * q->out[0].y_bs + eoff);
        rpv = has_res ? uniform_ptr(q->out[0].res + (long long)c_b * q->out[0].res_bs + eoff) : ypv;
        apv = has_acc ? uniform_ptr(q->out[0].acc + (long long)c_b * q->out[0].acc_bs + eoff) : ypv;
        ev = 0;
        if (++c_i < ntl) coord(c_i, c_b, c_n0);
    };

    // ---- prologue: chunk 0 of the first tile into LDS, chunk 1 in flight, the fragments of tap 0 in flight
    __syncthreads();                                 // (sbias)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    load_a(std::integral_constant<int, FIRST_SET>{}, 0, 0);
    stage_load();
    stage_store(lbuf0);
    stage_load();
    __syncthreads();

    // one tap: fragments of the next tap requested, B fragments of the four 32-column tiles read one tile ahead, 12 MFMAs
    auto tap_body = [&](auto t_tag, const unsigned *cur, int g) __attribute__((always_inline)) {
        constexpr int T = decltype(t_tag)::value;
        constexpr int SET = T & 1;
        u32x4 (&acur)[NPL] = *(SET ? &a1 : &a0);
        if constexpr (T == 0) {
            if (s_i < ntl && !PIPE_PERTURB(2)) stage_store((g & 1) ? lbuf0 : lbuf1);
            if constexpr (FIRST_SET == 1) {          // odd KT: tap 0's fragments arrived in set 1 (see load_a): move them, then set 1 is free for tap 1
                wait_vm(vm_seq - seqA1);
                asm volatile("" : "+v"(a1[0]), "+v"(a1[1]));
                a0[0] = a1[0];
                a0[1] = a1[1];
                asm volatile("" : "+v"(a0[0]), "+v"(a0[1]));
                seqA0 = seqA1;
            }
        }
        // the next tap's fragments (the last tap of the last chunk requests nothing)
        if constexpr (T + 1 < KT) {
            load_a(std::integral_constant<int, (T + 1) & 1>{}, chunk, T + 1);
        } else {
            if (g + 1 < G) load_a(std::integral_constant<int, FIRST_SET>{}, (chunk + 1 == nchunks) ? 0 : chunk + 1, 0);
        }
        if constexpr (T == 0) {
            if (l_i < ntl && !PIPE_PERTURB(2)) stage_load();
        }
        const unsigned *xs = cur + (lhalf * W + wn * CW + l31 - p.lo + (p.off0 + T * p.tstep)) * 4;
        // B fragments of the four 32-column tiles, two tiles (one PAIR) at a time, the next pair read under the MFMAs of the current one.
        // The six MFMAs of a pair alternate between its two accumulator tiles: l x h (0), l x h (1), h x l (0), h x l (1), h x h (0), h x h (1)
        // -- every accumulator still sees its three terms in the order of conv_split_body.inc (bit-identical sums), but two consecutive
        // MFMAs are independent: a lone wave issuing three dependent MFMAs back to back reaches ~70 % of the matrix pipe
        // (tools/pipe_perturb.py: one wave per SIMD), and its SIMD partner is not always in its own MFMA section to fill the gaps.
        u32x4 bA[2][NPL], bB[2][NPL];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) bA[jj][pl] = *reinterpret_cast<const u32x4 *>(xs + pl * PLSZ + jj * 128);
        if (!PIPE_PERTURB(4)) wait_vm_fast2(vm_seq - (SET ? seqA1 : seqA0));
        asm volatile("" : "+v"(acur[0]), "+v"(acur[1]));
        auto pair_mfma = [&](int j0, u32x4 (&b)[2][NPL]) __attribute__((always_inline)) {
            if (PIPE_PERTURB(256) && wn) return;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                acc[j0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, acur[1]), __builtin_bit_cast(f16x8, b[jj][0]), acc[j0 + jj], 0, 0, 0);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                acc[j0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, acur[0]), __builtin_bit_cast(f16x8, b[jj][1]), acc[j0 + jj], 0, 0, 0);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                acc[j0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, acur[0]), __builtin_bit_cast(f16x8, b[jj][0]), acc[j0 + jj], 0, 0, 0);
        };
        if (!PIPE_PERTURB(16)) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) bB[jj][pl] = *reinterpret_cast<const u32x4 *>(xs + pl * PLSZ + (2 + jj) * 128);
        }
        __builtin_amdgcn_sched_barrier(0);
        pair_mfma(0, bA);
        __builtin_amdgcn_sched_barrier(0);
        {
            // the chunk's epilogue events sit at fixed taps: event k of the chunk at tap k * KT / 3 for the waves of column half 0 and one tap
            // later for those of half 1
            constexpr int K0 = pipe_event_at(T, KT, 0), K1 = pipe_event_at(T, KT, 1);
            if constexpr (K0 >= 0 || K1 >= 0) {
                const int kk = wn ? K1 : K0;
                if (kk >= 0 && kk < epc && ev < PIPE_EVENTS && !PIPE_PERTURB(1)) epi_event();
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        pair_mfma(2, bB);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int g = 0; g < G; ++g) {
        const unsigned *cur = (g & 1) ? lbuf1 : lbuf0;
        for_each_tap([&](auto t_tag) __attribute__((always_inline)) { tap_body(t_tag, cur, g); }, std::make_integer_sequence<int, KT>{});
        __syncthreads();
        // the accumulators follow the tile's running scale (pend = 1 unless the chunk stored during this one raised the largest magnitude):
        // unconditional, so that the accumulators stay in place (a conditional multiply made hipcc copy all 64 at every chunk boundary)
#pragma unroll
        for (int j = 0; j < NT_W; ++j) acc[j] *= pend;
        pend = 1.f;
        if (++chunk == nchunks) {
            chunk = 0;
            tile_end();
        }
    }
    // ---- the last tile's epilogue (nothing left to hide it under)
    while (ev < PIPE_EVENTS && !PIPE_PERTURB(1)) epi_event();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// p: as prepared by vs_conv_forward for the 128 x 256 tile (cfg 0) of the split-f16 arithmetic; the caller has checked pipe_eligible()
int launch_pipe(ConvParams p, int span, int ncu, hipStream_t s) {
    p.W = PIPE_BN + span;
    const size_t lds = (size_t)2 * 2 * 2 * p.W * 16 + 64 + 512 + (size_t)8 * 2 * 8 * PIPE_CW * sizeof(float) + (size_t)8 * 3 * 1024;

    const int mb = p.MT / 4;
    const int ntiles = (p.N / PIPE_BN) * p.B;
    dim3 grid((unsigned)std::max(1, std::min(ntiles, ncu / mb)), (unsigned)mb, 1);
    auto go = [&](auto kt_tag, auto acc_tag) -> int {
        constexpr int KT = decltype(kt_tag)::value;
        constexpr bool HA = decltype(acc_tag)::value;
        auto kern = conv_pipe_kernel<KT, HA>;
        static bool attr_set = false;
        if (!attr_set) {
            VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
        VS_CHECK_HIP(hipGetLastError());
        set_last_kernel("conv_pipe_kernel<%d, %s>", KT, HA ? "true" : "false");
        return VS_OK;
    };
    const bool ha = p.out[0].acc != nullptr;
    switch (p.KT) {
#define VS_PIPE_CASE(K) case K: return ha ? go(std::integral_constant<int, K>{}, std::true_type{}) : go(std::integral_constant<int, K>{}, std::false_type{});
        VS_PIPE_CASE(3) VS_PIPE_CASE(5) VS_PIPE_CASE(7) VS_PIPE_CASE(9) VS_PIPE_CASE(11)
#undef VS_PIPE_CASE
        default: break;
    }
    set_error("launch_pipe: no instance for %d taps", p.KT);
    return VS_EUNSUPPORTED;
}

}  // namespace vs
