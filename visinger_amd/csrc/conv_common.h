// conv_common.h -- launch parameters and device helpers shared by the conv kernels of libvisinger_hip.so
// (conv_engine.hip: fp32 MFMA + Winograd F(2,3); conv_split.hip: split-bf16 MFMA).
#pragma once
#include "vs_internal.h"

namespace vs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CK = 16;        // input channels staged per LDS chunk
constexpr int MAX_SPAN = 64;  // max (taps-1)*dilation supported by the LDS window
constexpr int MT_ALLOC = 8;   // packed weights are zero-padded to a multiple of this many M tiles

struct OutSpec {
    float *y;
    const float *res;
    const float *acc;
    long long y_bs, res_bs, acc_bs;
    float scale;
    int out_act, out_mask, mode;
    int rows;    // rows addressable through res/acc (bounds of the buffer descriptors)
};

struct ConvParams {
    const float *x;
    long long x_bs;
    const float *wp;      // packed weights
    const float *biasp;   // packed bias over virtual rows (always present, zeros if no bias)
    const float *bias_b;  // optional per-item bias over ORIGINAL rows
    long long bias_b_bs;
    const float *mask;    // [B, Tin]
    float *logdet;
    float *ld_part;       // PAIRED coupling forward: per-(item, channel pair, 32-column tile) partial log-det sums [B][ld_slots], zeroed by the host;
    int ld_nt, ld_slots;  // each wave stores its own slot (no atomics), logdet_reduce_kernel adds them up in a fixed order: bit-reproducible
    OutSpec out[2];
    int split_row;
    int kind, pair_mode, in_act;
    int B, Cin, Tin;
    int M;        // valid virtual rows
    int MT;       // virtual M tiles
    int N;        // virtual columns (time positions computed)
    int Tout;     // true output length (row stride of y)
    int c_out;    // original output rows
    int Hh;       // PAIRED: half rows
    int KT, CP, nchunks;
    int off0, tstep, lo, W;
    int up, upK, uppad, dmin;   // transposed: stride, kernel, padding, min delta
    int row_lo, row_hi;         // only rows in [row_lo, row_hi) are stored
    int fast_epi;               // host-checked preconditions of the LDS-transposed float4 epilogue
    int x_bf16, y_bf16;         // bf16-RESIDENT tensors (plain-bf16 arithmetic only): x / (y, res, acc) hold bf16 elements; strides in elements
    const float *wscale;        // split-f16 arithmetic: {s_w, 1 / s_w}, the power-of-two scale the packed weight planes carry (device memory)
    int dbg;                    // perturbation experiments (VS_WINO_DBG: 1 = no weight-fragment loads, 2 = no staging), 0 in production
    unsigned long long *stamps; // debug: per-workgroup phase time stamps (NULL in production)
};

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.1f * v; }
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
// tanh via one exp: t = e^{-2|x|} in (0,1], tanh|x| = (1-t)/(1+t).  Branch-free; absolute error <= ~1e-7 (the
// subtraction is exact, the error is t's rounding), which is what a tanh-bounded output needs.  The device
// library's tanhf is branchy and would be inlined once per accumulator register.
__device__ __forceinline__ float tanh_fast(float v) {
    const float t = expf(-2.0f * fabsf(v));
    return copysignf((1.0f - t) / (1.0f + t), v);
}


__device__ __forceinline__ void stamp(const ConvParams &p, int slot) {
    if (p.stamps && threadIdx.x == 0) {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        p.stamps[(size_t)lin * 64 + slot] = __builtin_amdgcn_s_memrealtime();
        if (slot == 1 || slot == 2) p.stamps[(size_t)lin * 64 + 3 + slot] = __builtin_amdgcn_s_memtime();   // [4], [5]: shader clock
        if (slot == 0) p.stamps[(size_t)lin * 64 + 7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                                                       ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);
    }
}

// ---- split-bf16 helpers (conv_split.hip, resblock_pair_split.hip)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// gfx950 (MI355X), found in round 6 (tools/ubench/pk_opsel_probe.hip): a packed-fp32 VALU instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32)
// whose op_sel makes the LOW result read the HIGH register of its SECOND source -- `op_sel:[_,1]`, e.g. a pair times a scalar that lives in an odd register, or a
// horizontal add -- returns wrong low results in lanes 48-63 while the other wave of its SIMD issues MFMAs (0.1-0.5 % of the instructions in the probe; never with
// one wave per SIMD, never for the op_sel of src0 / src2, for op_sel_hi, or for v_pk_mov_b32).  hipcc's SLP vectoriser forms such instructions from plain scalar
// code; visinger_amd/csrc/build.py disassembles every object and fails the build on one.  These two keep a multiply / an add out of the vectoriser's reach where it
// would form one (an opaque scalar instruction; the register operands stay visible to hipcc's wait-count and hazard passes).
__device__ __forceinline__ float mul_f32_scalar(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float add_f32_scalar(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned f2u(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float u2f(unsigned v) { return __builtin_bit_cast(float, v); }
// (hi16(b) << 16) | hi16(a)
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ unsigned rne_bf16(float v) {       // bf16 bits in the HIGH half (finite inputs)
    const unsigned u = f2u(v);
    return u + 0x7fffu + ((u >> 16) & 1u);
}
// four consecutive bf16 elements <-> float4 (bf16-resident tensors)
__device__ __forceinline__ float4 bf4_to_f4(uint2 u) {
    return make_float4(u2f(u.x << 16), u2f(u.x & 0xffff0000u), u2f(u.y << 16), u2f(u.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 f4_to_bf4(float4 v) {
    return make_uint2(pack_hi(rne_bf16(v.x), rne_bf16(v.y)), pack_hi(rne_bf16(v.z), rne_bf16(v.w)));
}
// planes of a pair of values -> one packed dword per plane
template <int NPL>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&out)[NPL]) {
    if constexpr (NPL == 1) {
        out[0] = pack_hi(rne_bf16(a), rne_bf16(b));
    } else {
        unsigned ab = f2u(a), bb = f2u(b);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            out[pl] = pack_hi(ab, bb);
            if (pl + 1 < NPL) {
                a -= u2f(ab & 0xffff0000u);
                b -= u2f(bb & 0xffff0000u);
                ab = f2u(a);
                bb = f2u(b);
            }
        }
    }
}

// ---- split-f16 helpers (TERMS = 3, VS_MATH_SPLIT3): x * s = xh + xl with xh = RNE_f16(x * s), xl = RNE_f16(x * s - xh) -- 22 significant
// bits under a power-of-two scale s that keeps the largest magnitude of the tile below 2^15; three cross products hh + hl + lh on
// v_mfma_f32_32x32x16_f16, each exact in the fp32 accumulator; the dropped ll term and the 22-bit representation are <= 2^-22 of a product
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_h(float a, float b, unsigned (&out)[2]) {       // a, b already scaled
    const f32x2 v = {a, b};
    const f16x2 h = __builtin_convertvector(v, f16x2);                       // v_cvt_pk_f16_f32 (round to nearest even)
    const f32x2 r = v - __builtin_convertvector(h, f32x2);                   // exact
    const f16x2 l = __builtin_convertvector(r, f16x2);
    out[0] = __builtin_bit_cast(unsigned, h);
    out[1] = __builtin_bit_cast(unsigned, l);
}
// biased exponent eb of the largest |value| of a tile -> the tile's scale 2^(141 - eb): every magnitude < 2^(eb - 126) lands below 2^15
constexpr int F16_EB_MIN = 24;      // tiles whose largest magnitude is below 2^-102 share the scale of 2^-102 (their values flush towards 0)
__device__ __forceinline__ float f16_scale(int eb) { return u2f((unsigned)(268 - eb) << 23); }
__device__ __forceinline__ float f16_inv_scale(int eb) { return u2f((unsigned)(eb - 14) << 23); }
// Running maximum of FINITE magnitudes in a form that costs two VALU per value: key(v) = (bits << 1) + 2^24 drops the sign, keeps the
// order of finite magnitudes (keys 2^24 .. 2^32 - 1) and wraps +-Inf / NaN to keys below 2^24, i.e. below every finite value: a
// non-finite activation must not set the tile's scale (it would flush every finite value of the tile to zero) -- it becomes Inf / NaN
// planes under the scale of its finite neighbours and poisons exactly the outputs whose receptive field holds it, as in fp32.
__device__ __forceinline__ unsigned f16_maxkey(unsigned key, float v) { return max(key, (f2u(v) << 1) + 0x01000000u); }
__device__ __forceinline__ int f16_key_exponent(unsigned key) { return max((int)(key >> 24) - 1, 0); }      // biased exponent of the largest finite magnitude
// largest value of a non-negative quantity over the 64 lanes of a wave, result wave-uniform: an inclusive max-scan in six DPP steps (shifts
// by 1, 2, 4, 8 inside the rows of 16 lanes with 0 shifted in, then row_bcast:15 / row_bcast:31 across the rows) whose last lane holds the
// maximum -- ~60 cycles.  (The bisection with eight dependent ballots it replaces cost 660 cycles per staged chunk: a third of the fixed
// cost of a chunk on the 32-row tiles, tools/conv_stamps.py with -DVS_SPLIT_PERTURB.)
__device__ __forceinline__ int wave_max_u8(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));      // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));      // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));      // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));      // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));     // row_bcast:15 into rows 1 and 3
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));     // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// 16 bytes per lane from global memory straight into LDS (no register destination: nothing for hipcc to copy before the data lands):
// lane i's bytes go to lds_dst + 16 * i (tools/ubench/glds_layout.hip); counted in vmcnt like any load.  M0 is written in the statement
// that reads it (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(int lane_byte_off, const char *base, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_off), "s"(base), "s"(lds_dst) : "memory");
}

// conv_split.hip
struct vs_split_pack {            // re-pack of the fp32 fragment-order weights into bf16 planes
    const float *wp;              // Wp[m_tile][tap][chunk][quad(2)][64][4] (pack_conv_kernel)
    void *ws;                     // Ws[m_tile][tap][chunk][plane][64 lanes][8 bf16]
    int MT_alloc, KT, nchunks, terms;
    float *wscale;                // terms = 3 (two f16 planes): out, {s_w, 1 / s_w} with s_w = 2^(14 - floor(log2 max|w|))
    const unsigned *maxbits;      // terms = 3: the largest |w| (bits) when the caller already has it, else NULL (found here, through `scratch`)
    unsigned *scratch;            // terms = 3 without maxbits: one device word of the handle for the maximum (no allocation on the caller's stream)
};
int split_planes(int terms);
int pack_split(const vs_split_pack &q, hipStream_t s);
int pack_split_pair(const vs_split_pack &q0, const vs_split_pack &q1, hipStream_t s);   // terms = 3, both maxima ready: one launch
int pack_split_multi(const vs_split_pack *jobs_dev, const unsigned *blk0_dev, int n, unsigned nblocks, hipStream_t s);   // terms = 3, maxima ready, table in device memory: one launch
// cfg: tile shape as chosen by vs_conv_forward for the direct engine (0: 128 rows, 1/3: 64, 2: 32 x 256, 6: 32 x 128; 4/5: paired); span = receptive span
int launch_split(const ConvParams &p, int cfg, int terms, int span, hipStream_t s);   // (p.x_bf16 / p.y_bf16: terms = 1, cfg 0 / 2 / 3 / 6)


// conv_ktap.hip / conv_ktap_bf16.hip (conv_ktap.inc): the 128 x 256 tile with the taps unrolled and the staging of the next chunk in the MFMA shadows, in the
// split-f16 x3 arithmetic (terms 3; bit-identical to conv_split_kernel<1, 8, 4, 1, 3>) and in plain bf16 (terms 1; conv_split_kernel<1, 8, 4, 1, 1> and its
// bf16-resident variants); p as for launch_split(cfg 0).  Preconditions: plain stride-1 conv, C_in % 16 == 0, ktap_instance(terms, KT, io, in_act).
bool ktap_instance(int terms, int cfg, int kt, int io, int in_act);     // cfg: launch_split's tile shape (0, 1 / 3, 6); io: bit 0 -- x, bit 1 -- y / res / acc hold bf16 elements
int launch_ktap(const ConvParams &p, int cfg, hipStream_t s);
int launch_ktap_small(const ConvParams &p, int cfg, hipStream_t s);      // (conv_ktap_small.hip: the 64 x 256 and 32 x 128 tiles of the split-f16 arithmetic)
int launch_ktap_bf16(const ConvParams &p, int cfg, hipStream_t s);
bool ktap_pair_instance(int terms, int kt, int io, int in_act, int mt);      // conv_ktap_pair.hip: VS_CONV1D_PAIRED, the 128 virtual rows x 128 columns tile
int launch_ktap_pair(const ConvParams &p, int terms, hipStream_t s);
bool ktap_tr_instance(const ConvParams &p, int terms, int cfg);               // conv_ktap.hip: VS_CONV_TRANSPOSE1D with two taps per phase (k = 2 * stride), the 128 x 256 tile, split-f16 on fp32 tensors
int launch_ktap_tr(const ConvParams &p, hipStream_t s);

}  // namespace vs
