// attention_dma.hip -- relattn_dma_kernel: MultiHeadAttention.attention (reference modules/rel_transformer.py:148-179 + 181-243) for the
// long-form configuration (BASELINE config 5: T_mel 4096, hidden 512 -> 2 heads of 256 channels, plain-bf16 arithmetic), round 4.
// Same algorithm, layouts and arithmetic as relattn_bf16_kernel<DT, 32, 1, true> (attention_bf16.hip: streaming softmax over 32-key
// tiles, -1e4 mask fill, banded relative terms, K / V tiles pre-packed as LDS images by attn_pack_kv_kernel); what changes is how a tile
// reaches LDS and how the matrix instructions are fed (263 TFLOP/s = 0.105 of the bf16 peak, one wave per SIMD,
// 132 B of scratch, every load round trip exposed):
//   * the image of a tile (K, V, key mask: 36 992 contiguous bytes) goes from global memory STRAIGHT into LDS by LDS-DMA
//     (global_load_lds_dwordx4: no register destination) into a ring of three slots, TWO tiles ahead of the one being used: no staging
//     registers (72 VGPRs in the old kernel, which then used all 512 registers and moved 1 186 values between the two register files),
//     no LDS stores, and an L2 / HBM round trip has two whole tiles to complete.  One barrier per tile: the wave's own pieces are waited
//     for by count (s_waitcnt vmcnt(n): the next tile's 9 or 10 pieces may stay in flight), the barrier makes every wave's pieces visible
//     and retires every read of the slot that is refilled right after it;
//   * S^T = K^T Q accumulates even and odd k-steps in TWO accumulator tiles (added once per tile): a chain of 16 dependent MFMAs on one
//     tile issues at ~70 % of the pipe from a lone wave (round 4, tools/pipe_perturb.py); the P V MFMAs of a k-step already walk eight
//     independent output tiles;
//   * every LDS fragment is read one MFMA ahead of its use.
// Results differ from relattn_bf16_kernel only by the summation order of S^T (even + odd k-steps): both are held to the same bounds
// against the fp32 / fp64 restatements (tests/test_conv_split_gpu.py::test_bf16_attention_matches_fp32_kernel, test_attention_dma_gpu.py).
#include "attn_common.h"
#include "conv_common.h"

namespace vs {

// A/B knobs of the fragment pipeline (defaults = what measured fastest on the box)
#ifndef ATT_KD
#define ATT_KD 4        // K fragments in flight ahead of their MFMA
#endif
#ifndef ATT_VD
#define ATT_VD 3        // V fragments in flight
#endif
#ifndef ATT_PIN
#define ATT_PIN 0       // 1: query fragments pinned in the accumulator file
#endif
#ifndef ATT_SB
#define ATT_SB 1        // 1: sched_barrier between (MFMA, next read) steps
#endif

// WPQ = waves per group of 32 queries.  1: the round-4 kernel, a wave owns all DT output tiles of its queries (128 accumulators at 256 channels: one wave per SIMD,
// nothing to overlap its own S^T -> softmax -> P V chain with).  2 (round 6): a PAIR of waves (w, w + 4) shares a query group and splits the
// head's channels: each holds HALF of the query fragments and accumulates HALF of the output tiles (DT / 2: 64 accumulators at 256 channels).  S^T = K^T Q
// contracts over the channels, so each wave forms the partial scores of its half, the two exchange them through LDS (one more barrier per tile) and add --
// a + b in one wave, b + a in the other: the same bits, so both run the SAME softmax and stay in step without exchanging anything else.  A workgroup is eight
// waves, two per SIMD: the matrix pipe sees two independent MFMA chains, the VALU two softmax streams; all eight move the tile images (half the pieces each).
template <int DT, int WPQ>
__global__ void __launch_bounds__(256 * WPQ, WPQ) relattn_dma_kernel(const AttnParams p) {
    static_assert(WPQ == 1 || (WPQ == 2 && DT % 2 == 0), "one wave per query group, or a pair splitting an even number of output tiles");
    constexpr int DH = DT / WPQ;                     // output tiles (of 32 channels) per wave
    constexpr int AKT = 32;
    constexpr int DKR = DT * 32;                     // padded head dim
    constexpr int NKS = DKR / 16;                    // k-steps of S^T = K^T Q
    constexpr int AVP = AKT / 2 + 4;                 // V row pitch in dwords
    constexpr int KPL = (DKR / 8) * AKT * 4;         // dwords of the K image
    constexpr int VPL = DKR * AVP;                   // dwords of the V image
    constexpr int IMG = KPL + VPL + AKT;             // dwords of a tile image: K, V, key mask
    constexpr int NU4 = IMG / 4;                     // 16-byte units of an image
    constexpr int PIECE = 256 * WPQ;                 // 16-byte units of a piece: one per thread of the workgroup (round 6: ALL waves move the image -- with the pair
                                                     // form four DMA waves kept their partners waiting at the score exchange for the ~600 cycles nine issues take)
    constexpr int PBYTES = PIECE * 16;
    constexpr int NFULL = NU4 / PIECE;               // pieces every wave issues (one 16-byte unit per lane each)
    constexpr int NTAIL = NU4 - NFULL * PIECE;       // units of the last, partial piece (the first lanes of the workgroup)
    static_assert(IMG % 4 == 0 && NTAIL >= 0 && NTAIL < PIECE && NKS % 2 == 0, "image geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_id & 3;                    // query group of the workgroup
    const int dpart = wave_id >> 2;                  // which DH output tiles this wave accumulates (0 with WPQ = 1)
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware placement (1-D grid): workgroup L runs on XCD L % 8 and workgroups are dispatched in order, so the query blocks of ONE
    // (batch, head) pair are given ids of one residue class -- they then run on one XCD at about the same time and stream the pair's
    // K / V images through THAT XCD's L2 once (76 MB of images per launch against 8 x 4 MB of L2: with the plain (query block, head,
    // batch) grid the 32 query blocks of a pair sat on all eight XCDs and every tile came from beyond L2 several times).
    const int nqb = (p.T + 127) / 128, npair = p.nh * p.B;
    int qblk, pair;
    if ((npair & 7) == 0) {
        const int L = blockIdx.x, xcd = L & 7, idx = L >> 3;
        qblk = idx % nqb;
        pair = (idx / nqb) * 8 + xcd;
    } else {
        qblk = blockIdx.x % nqb;
        pair = blockIdx.x / nqb;
    }
    const int b = pair / p.nh, h = pair - b * p.nh;
    const int i0 = (qblk * 4 + wave) * 32;
    const int dk = p.dk, T = p.T;
    const int nrel = (p.ws >= 0 && p.rel_k) ? 2 * p.ws + 1 : 0;

    unsigned *const ring = reinterpret_cast<unsigned *>(smem);           // [3][IMG]
    float *const QRs = smem + 3 * IMG;                                    // [4][32][ATT_QRS] rel-key logits
    float *const Sws = QRs + 4 * 32 * ATT_QRS;                            // [4][32][ATT_QRS] in-window raw scores
    float *const Xs = Sws + 4 * 32 * ATT_QRS;                             // WPQ = 2: [8 waves][16][64] partial scores of a tile, read by the partner wave
    float *const RVs = smem;                                              // [nrel][dk] relative value embeddings: over the ring, after the loop
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)ring);

    const float *qb = p.q + (long long)b * p.bs + (long long)h * dk * T;
    const float *maskb = p.mask ? p.mask + (long long)b * T : nullptr;
    const float *relk = nrel ? p.rel_k + (long long)(p.nh_rel == 1 ? 0 : h) * nrel * dk : nullptr;
    const float *relv = nrel ? p.rel_v + (long long)(p.nh_rel == 1 ? 0 : h) * nrel * dk : nullptr;

    // (debug, tools/attn_phase_stamps.py --wide: shader-clock stamps of the phases of workgroup (0, 0, 0), wave 0; p.stamps is null in production)
#define PSTAMP(k)                                                                                                    \
    do {                                                                                                             \
        if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wave_id == 0) {                     \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                              \
            if (lane == 0) p.stamps[1000 + (k)] = t_;                                                                \
        }                                                                                                            \
    } while (0)
    PSTAMP(0);
    // ---- query fragments: B operand of S^T = K^T Q, element j of k-step ks = Q[d = 16 ks + 8 half + j][query l31], RNE to bf16 ----
    const int qi = i0 + l31;
    const int qic = min(qi, T - 1);
    auto planes8 = [&](const float (&v)[8]) __attribute__((always_inline)) -> u32x4 {
        u32x4 o;
        o.x = pack_hi(rne_bf16(v[0]), rne_bf16(v[1])); o.y = pack_hi(rne_bf16(v[2]), rne_bf16(v[3]));
        o.z = pack_hi(rne_bf16(v[4]), rne_bf16(v[5])); o.w = pack_hi(rne_bf16(v[6]), rne_bf16(v[7]));
        return o;
    };
    // rel-key logits QR[i][r] = (q_i / sqrt(dk)) . rel_k[r] in fp32: the table goes into LDS once per workgroup, transposed and zero-padded
    // ([d][16], over the ring, which no DMA has touched yet); every wave sums over ITS k-steps -- sixteen loads of q in flight, one broadcast
    // ds_read_b128 per four window positions -- and the two waves of a pair, which hold complementary halves of the head's channels, add their
    // halves through the logits' own LDS rows (0 + 1 in both: the same bits).
    // (Round 6, in two steps: the loop this replaces -- a load of q and nine of rel_k per channel, one channel in flight -- cost ~1 800 cycles
    // per channel, a fifth of this kernel's launch; then every wave walking ALL the k-steps, still 50 000 cycles per workgroup at launch
    // start: tools/attn_phase_stamps.py --wide.)
    float *RKs = smem;
    float qr[ATT_MAXREL];
#pragma unroll
    for (int r = 0; r < ATT_MAXREL; ++r) qr[r] = 0.f;
    if (nrel) {
#pragma unroll      // (consecutive lanes read consecutive channels of one window position: coalesced, all loads of a thread in flight)
        for (int i = 0; i < DKR * ATT_MAXREL / (256 * WPQ); ++i) {
            const int e = tid + i * (256 * WPQ);
            const int r = e / DKR, d = e % DKR;
            const float w = relk[min(r, nrel - 1) * dk + min(d, dk - 1)];      // (unconditional load on a clamped index: the loads of a thread overlap)
            RKs[d * ATT_MAXREL + r] = (d < dk && r < nrel) ? w : 0.f;
        }
    }
    __syncthreads();
    PSTAMP(1);
    constexpr int NKW = NKS / WPQ;                   // k-steps of S^T this wave contracts (its half of the channels with WPQ = 2)
    const int ks0 = dpart * NKW;
    u32x4 qf[NKW];
#pragma unroll
    for (int ks = 0; ks < NKW; ++ks) {
        float qv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 16 * (ks0 + ks) + 8 * half + j;
            const float v = qb[(long long)min(d, dk - 1) * T + qic];
            qv[j] = (d < dk && qi < T) ? v * p.scale : 0.f;
        }
        qf[ks] = planes8(qv);
        // the query fragments live in the ACCUMULATOR file for the whole kernel (an MFMA takes its A / B operands from either file): 64
        // arch VGPRs freed for LDS fragments in flight -- an LDS round trip is ~200 cycles here and a lone wave has only its own loads
        // to cover it with, so the K / V fragments are requested eight / six MFMAs ahead
        if (ATT_PIN) asm volatile("" : "+a"(qf[ks]));
    }
    // (the logits in a pass of their own over the wave's k-steps -- q comes back from L1 / L2 --, sixteen loads in flight: forming them inside the fully
    //  unrolled fragment loop above left the prologue 16 registers short, and a spill is not allowed beside the hand-counted waits of the ring)
    if (nrel) {
#pragma unroll 2
        for (int ks = 0; ks < NKW; ++ks) {
            float qv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 16 * (ks0 + ks) + 8 * half + j;
                const float v = qb[(long long)min(d, dk - 1) * T + qic];
                qv[j] = (d < dk && qi < T) ? v * p.scale : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 *row = reinterpret_cast<const float4 *>(RKs + (16 * (ks0 + ks) + 8 * half + j) * ATT_MAXREL);
#pragma unroll
                for (int c4 = 0; c4 < ATT_MAXREL / 4; ++c4) {      // (all 16 columns of the zero-padded table: straight-line code)
                    const float4 w = row[c4];
                    qr[4 * c4 + 0] += qv[j] * w.x; qr[4 * c4 + 1] += qv[j] * w.y; qr[4 * c4 + 2] += qv[j] * w.z; qr[4 * c4 + 3] += qv[j] * w.w;
                }
            }
        }
    }
    float *QRw = QRs + wave * 32 * ATT_QRS;
    float *Sww = Sws + wave * 32 * ATT_QRS;
    for (int e = lane; e < 32 * ATT_QRS; e += 64) Sww[e] = -INFINITY;
    if (nrel) {
#pragma unroll
        for (int r = 0; r < ATT_MAXREL; ++r) qr[r] += __shfl_xor(qr[r], 32);      // the two lane halves hold complementary d's
        if (WPQ == 1 || dpart == 0) {
            if (half == 0) {
#pragma unroll
                for (int r = 0; r < ATT_MAXREL; ++r) QRw[l31 * ATT_QRS + r] = qr[r];
            }
        }
        if constexpr (WPQ == 2) {
            __syncthreads();
            if (dpart == 1 && half == 0) {
#pragma unroll
                for (int r = 0; r < ATT_MAXREL; ++r) QRw[l31 * ATT_QRS + r] += qr[r];
            }
        }
    }
    __syncthreads();                                     // (the table is read, the logits are complete: the ring may be written)
    PSTAMP(2);

    f32x16 o[DH];
#pragma unroll
    for (int t = 0; t < DH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_half = 0.f;
    const float mi = (maskb && qi < T) ? maskb[qi] : 1.f;

    // ---- the tile ring ----
    const int ntiles = (T + AKT - 1) / AKT;
    const char *const imgb = reinterpret_cast<const char *>(p.kvimg + ((long long)(b * p.nh + h) * ntiles) * IMG);
    const int lane16 = tid * 16;                                    // byte offset of this lane's unit inside a piece of 256 units
    // NFULL pieces of 256 units each (one 16-byte unit per lane: lane i of wave w moves unit 256 p + 64 w + i of the image to the same
    // unit of the slot), issued from ONE statement that saves M0 once: per piece an s_mov of the LDS address, the DMA, two scalar adds
    auto dma_tile = [&](int jt, int slot) __attribute__((always_inline)) {
        const char *src = imgb + (long long)jt * (IMG * 4);
        unsigned dst = ring_lds + slot * (IMG * 4) + wave_id * 1024;
        unsigned keep;
        // (two offset registers used alternately, each advanced only after the OTHER one's load has been issued: the add never follows
        //  the load that reads the register)
        int va = lane16, vb = lane16 + PBYTES;
#define VS_LD(v) "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 " v ", %4\n\ts_add_u32 %1, %1, %5\n\t"
#define VS_PAIR VS_LD("%2") VS_LD("%3") "v_add_u32 %2, %6, %2\n\tv_add_u32 %3, %6, %3\n\t"
#define VS_ODD VS_LD("%2") "s_nop 1\n\tv_add_u32 %2, %5, %2\n\t"
#define VS_DMA_OPS : "=&s"(keep), "+s"(dst), "+v"(va), "+v"(vb) : "s"(src), "n"(PBYTES), "n"(2 * PBYTES) : "memory", "scc"
        static_assert(NFULL == 3 || NFULL == 4 || NFULL == 6 || NFULL == 9, "pieces per tile image (192- / 256-channel heads, one or two waves per query group)");
        if constexpr (NFULL == 9) asm volatile("s_mov_b32 %0, m0\n\t" VS_PAIR VS_PAIR VS_PAIR VS_PAIR VS_ODD "s_mov_b32 m0, %0" VS_DMA_OPS);
        else if constexpr (NFULL == 6) asm volatile("s_mov_b32 %0, m0\n\t" VS_PAIR VS_PAIR VS_PAIR "s_mov_b32 m0, %0" VS_DMA_OPS);
        else if constexpr (NFULL == 4) asm volatile("s_mov_b32 %0, m0\n\t" VS_PAIR VS_PAIR "s_mov_b32 m0, %0" VS_DMA_OPS);
        else asm volatile("s_mov_b32 %0, m0\n\t" VS_PAIR VS_ODD "s_mov_b32 m0, %0" VS_DMA_OPS);
#undef VS_DMA_OPS
#undef VS_ODD
#undef VS_PAIR
#undef VS_LD
        const int voff = va;          // = lane16 + NFULL * PBYTES in every case
        if (NTAIL && tid < NTAIL) glds16(voff, src, dst);      // (the last, partial piece: the first waves only)
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (nothing of the prologue's loads may be counted against the ring)
    dma_tile(0, 0);
    if (ntiles > 1) dma_tile(1, 1);

#define ASTAMP(k)                                                                                                    \
    do {                                                                                                             \
        if (p.stamps && blockIdx.x == 0 && wave_id == 0 && jt < 120) {            \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                              \
            if (lane == 0) p.stamps[jt * 8 + (k)] = t_;                                                              \
        }                                                                                                            \
    } while (0)
    PSTAMP(3);
    for (int jt = 0; jt < ntiles; ++jt) {
        const int j0 = jt * AKT;
        ASTAMP(0);
        // this wave's pieces of tile jt have landed (those of tile jt + 1 -- NFULL, + 1 for the waves of the partial piece -- may stay in flight) ...
        if (jt + 1 < ntiles) {
            if (NTAIL && wave_id * 64 < NTAIL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NFULL + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NFULL) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // ... everybody's have, and everybody is done with tile jt - 1, whose slot takes tile jt + 2
        __syncthreads();
        ASTAMP(1);
        if (jt + 2 < ntiles) dma_tile(jt + 2, (jt + 2) % 3);
        ASTAMP(2);
        const unsigned *Kb = ring + (jt % 3) * IMG, *Vb = Kb + KPL;
        const float *Mb = reinterpret_cast<const float *>(Vb + VPL);

        // ---- S^T tile: rows = keys acc_row(r), columns (lanes) = queries; even / odd k-steps in two accumulators ----
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        {
            auto readK = [&](int ks) __attribute__((always_inline)) {
                return *reinterpret_cast<const u32x4 *>(Kb + ((2 * (ks0 + ks) + half) * AKT + l31) * 4);
            };
            // (fragments eight k-steps ahead of their MFMA: one wave per SIMD has nothing else to cover an LDS round trip with)
            constexpr int KD = ATT_KD;
            u32x4 kf[KD];
#pragma unroll
            for (int i = 0; i < KD; ++i) kf[i] = readK(i);
            if (ATT_SB) __builtin_amdgcn_sched_barrier(0);          // (the scheduler may sink every read to its use to save registers: pin the order)
#pragma unroll
            for (int ks = 0; ks < NKW; ++ks) {
                if (ATT_SB) __builtin_amdgcn_sched_barrier(0);
                const u32x4 kc = kf[ks % KD];
                if (ks + KD < NKW) kf[ks % KD] = readK(ks + KD);
                if (ks & 1) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kc), __builtin_bit_cast(bf16x8, qf[ks]), s1, 0, 0, 0);
                else s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kc), __builtin_bit_cast(bf16x8, qf[ks]), s0, 0, 0, 0);
            }
        }
        if constexpr (WPQ == 2) {
            // the partial scores of this wave's channels -> LDS, the partner's back: s0 becomes the whole score tile (own + partner: commutative, so both waves of
            // the pair hold the same bits), s1 zero.  Slot layout [register][lane]: a wave-instruction touches 256 consecutive bytes.
            float *const mine = Xs + wave_id * (16 * 64), *const theirs = Xs + (wave_id ^ 4) * (16 * 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = s0[r] + s1[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = (s0[r] + s1[r]) + theirs[r * 64 + lane]; s1[r] = 0.f; }
        }
        ASTAMP(3);
        const bool near_diag = nrel && (j0 + AKT - 1 >= i0 - p.ws) && (j0 <= i0 + 31 + p.ws);
        // the common tile: wholly inside the sequence, no padded key, no padded query in this wave, off the relative window -- the scores
        // are the MFMA results as they are (wave-uniform test: one LDS read of the tile's key mask per lane and two ballots)
        const bool plain = !near_diag && (j0 + AKT <= T) && __all(Mb[l31] != 0.f) && __all(mi != 0.f);
        float tmax = -INFINITY;
        float sv_[16];
        if (plain) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sv_[r] = s0[r] + s1[r];
                tmax = fmaxf(tmax, sv_[r]);
            }
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int j = j0 + jj;
            float sv = s0[r] + s1[r];
            if (near_diag) {
                const int rel = j - qi;
                if (rel >= -p.ws && rel <= p.ws) sv += QRw[l31 * ATT_QRS + rel + p.ws];
            }
            if (mi * Mb[jj] == 0.f) sv = -1e4f;          // masked_fill(mask == 0, -1e4)
            if (j >= T) sv = -INFINITY;                  // beyond the sequence: not part of the softmax
            if (near_diag) {
                const int rel = j - qi;
                if (rel >= -p.ws && rel <= p.ws && j < T) Sww[l31 * ATT_QRS + rel + p.ws] = sv;
            }
            sv_[r] = sv;
            tmax = fmaxf(tmax, sv);
        }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
        float psum = 0.f;
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pv[r] = __expf(sv_[r] - m_new);              // exp(-inf) = 0 for excluded keys; the result is rounded to 8 bits anyway: the fast exp
            psum += pv[r];
        }
        u32x4 pf[2];                                      // P^T fragments of the two 16-key k-steps
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = pv[8 * sh + e];
            pf[sh] = planes8(v8);
        }
        l_half = l_half * alpha + psum;
        m_run = m_new;
        // the running maximum of a row settles after a few tiles: skip the rescale of the output accumulators whenever no query of the
        // wave moved its maximum
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int t = 0; t < DH; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
                __builtin_amdgcn_sched_barrier(0);      // (one output tile at a time through the VGPRs: batching all eight costs 128 live registers)
            }
        }

        ASTAMP(4);
        // ---- O^T += V P^T: k-step s4 sums over the keys 16 s4 + 8 (j >> 2) + 4 half + (j & 3), the order of the V rows ----
        {
            const unsigned *const Vw = Vb + dpart * (DH * 32 * AVP);          // this wave's DH output tiles (V rows dpart * DH * 32 ..)
            auto readV = [&](int s4, int t) __attribute__((always_inline)) {
                return *reinterpret_cast<const u32x4 *>(Vw + (t * 32 + l31) * AVP + s4 * 8 + half * 4);
            };
            constexpr int VD = ATT_VD;
            u32x4 vf[VD];
#pragma unroll
            for (int i = 0; i < VD; ++i) vf[i] = readV(i / DH, i % DH);
            if (ATT_SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 2 * DH; ++n) {
                if (ATT_SB) __builtin_amdgcn_sched_barrier(0);
                const u32x4 vc = vf[n % VD];
                if (n + VD < 2 * DH) vf[n % VD] = readV((n + VD) / DH, (n + VD) % DH);
                o[n % DH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vc), __builtin_bit_cast(bf16x8, pf[n / DH]), o[n % DH], 0, 0, 0);
            }
        }
        ASTAMP(5);
    }
#undef ASTAMP
    PSTAMP(4);

    // ---- finish: normalise, add the relative-value term (fp32), store ----
    __syncthreads();                                     // every read of the ring has retired: its first bytes take the relative value table
    for (int e = tid; e < nrel * DKR; e += 256 * WPQ) RVs[e] = (e % DKR < dk) ? relv[(e / DKR) * dk + e % DKR] : 0.f;      // rows of DKR: float4 reads below
    __syncthreads();
    const float l_tot = l_half + __shfl_xor(l_half, 32);
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int t = 0; t < DH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= inv;
    const int t_base = dpart * DH;                       // first output tile of this wave
    // sum_r p[i, i + r - ws] * rel_v[r]: the window index is the (rolled) outer loop so that every access to the output accumulators
    // has a compile-time index
#pragma unroll 1
    for (int rr = 0; rr < nrel; ++rr) {
        const float w = expf(Sww[l31 * ATT_QRS + rr] - m_run) * inv;
        const float *rv = RVs + rr * DKR;
#pragma unroll
        for (int t = 0; t < DH; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 v4 = *reinterpret_cast<const float4 *>(rv + (t_base + t) * 32 + 8 * r4 + 4 * half);
                o[t][4 * r4 + 0] += w * v4.x; o[t][4 * r4 + 1] += w * v4.y; o[t][4 * r4 + 2] += w * v4.z; o[t][4 * r4 + 3] += w * v4.w;
            }
    }
    PSTAMP(5);
    float *ob = p.out + (long long)b * p.out_bs + (long long)h * dk * T;
#pragma unroll
    for (int t = 0; t < DH; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = (t_base + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (d < dk && qi < T) ob[(long long)d * T + qi] = o[t][r];
        }
    }
    PSTAMP(6);
#undef PSTAMP
}

template <int DT, int WPQ>
static int launch_dma_dt(const AttnParams &p, hipStream_t s) {
    constexpr int DKR = DT * 32, IMG = (DKR / 8) * 32 * 4 + DKR * 20 + 32;
    const size_t lds = 4 * ((size_t)3 * IMG + 2 * 4 * 32 * ATT_QRS + (WPQ == 2 ? 8 * 16 * 64 : 0));
    auto kern = relattn_dma_kernel<DT, WPQ>;
    static bool attr_set = false;
    if (!attr_set) {
        VS_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)(ceil_div(p.T, 128) * p.nh * p.B), 1, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256 * WPQ), lds, s, p);
    VS_CHECK_HIP(hipGetLastError());
    set_last_kernel("relattn_dma_kernel<%d, %d>", DT, WPQ);
    return VS_OK;
}

// p.kvimg holds the images attn_pack_kv_kernel<DT, 32> wrote for this launch (32-key tiles: heads of 129 .. 256 channels); no key split
bool attn_dma_supported(const AttnParams &p) {
    const int DT = (int)ceil_div(p.dk, 32);
    return p.kvimg && DT > 4 && DT <= 8 && !(p.part && p.ksplit > 1) && !opt(OPT_NO_ATTN_DMA);
}

int launch_attn_dma(const AttnParams &p, hipStream_t s) {
    const int DT = (int)ceil_div(p.dk, 32);
    if (opt(OPT_ATTN_DMA_ONE_WAVE)) return DT <= 6 ? launch_dma_dt<6, 1>(p, s) : launch_dma_dt<8, 1>(p, s);      // (the round-4 form: A/B switch)
    return DT <= 6 ? launch_dma_dt<6, 2>(p, s) : launch_dma_dt<8, 2>(p, s);
}

}  // namespace vs
