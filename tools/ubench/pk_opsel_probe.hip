// pk_opsel_probe.hip -- does a packed-fp32 multiply whose op_sel modifiers make a result half read the OTHER half of a source pair give the
// right answer when a second wave on the same SIMD keeps the matrix pipe, the LDS and the vector-memory path busy?
//
// Round 6 found the answer "no" inside conv_split_kernel<..., 1>: hipcc's SLP vectoriser had turned `v[j] *= mk[i]` of the
// masked plain-bf16 staging into `v_pk_mul_f32 d, a, m op_sel:[0,1] op_sel_hi:[1,0]` (both halves of m crossed), and with two workgroups per
// CU the LOW result of that instruction came out unwritten in lanes 48-63, run-to-run differently; the same statement as a plain
// `v_pk_mul_f32 d, a, m` (an in-situ A/B with inline asm, tools/build_variant.py pkplain / pkcross) is clean.  This program isolates the
// instruction forms: a workgroup of eight waves, waves 0-3 make noise (MFMA + ds_read / ds_write + global loads), waves 4-7 (one per SIMD,
// beside a noise wave) run every op_sel form on operands freshly loaded from memory and compare with scalar multiplies.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pk_opsel_probe tools/ubench/pk_opsel_probe.hip && tools/ubench/pk_opsel_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int NFORMS = 16;
static const char *FORM_NAME[NFORMS] = {"v_pk_mul_f32 d, a, m", "v_pk_mul_f32 d, a, m op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 d, a, m op_sel:[1,0] op_sel_hi:[0,1]",
                                        "v_pk_mul_f32 d, a, m op_sel:[1,0]", "v_pk_mul_f32 d, a, m op_sel:[0,1]", "v_pk_mul_f32 d, a, m op_sel_hi:[1,0]",
                                        "v_pk_mul_f32 d, a, m op_sel_hi:[0,1]", "v_pk_add_f32 d, a, m op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32 d, a, m op_sel:[0,1]",
                                        "v_pk_add_f32 d, a, m op_sel:[1,0]", "v_pk_fma_f32 d, a, m, a op_sel:[0,1,0]", "v_pk_fma_f32 d, a, m, a op_sel:[0,0,1]",
                                        "v_pk_fma_f32 d, a, m, a op_sel:[1,0,0]", "v_pk_mov_b32 d, a, m op_sel:[1,0]", "v_pk_mov_b32 d, a, m op_sel:[0,1]",
                                        "v_pk_mov_b32 d, a, m op_sel:[1,1]"};

typedef unsigned long long u64;
#define PKI(FORM, INS)                                                                  \
    {                                                                                   \
        u64 r;                                                                          \
        asm volatile(INS : "=&v"(r) : "v"(a), "v"(m));                                  \
        got[FORM][0] = (unsigned)r;                                                     \
        got[FORM][1] = (unsigned)(r >> 32);                                             \
    }
#define PK(FORM, STR) PKI(FORM, "v_pk_mul_f32 %0, %1, %2 " STR)

__global__ void __launch_bounds__(512) probe(const float *__restrict__ x, const float *__restrict__ mk, unsigned *__restrict__ bad, float *__restrict__ sink,
                                             int iters, int noise) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long base = ((long long)blockIdx.x * 8 + wave) * 64 + lane;
    if (wave < 4) {
        // ---- noise: what the co-resident workgroup of the conv kernel does
        if (!noise) return;
        f32x16 acc = {};
        f16x8 fa, fb;
        for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.01f * (lane + i)); fb[i] = (_Float16)(0.02f * (lane - i)); }
        float g = 0.f;
        for (int it = 0; it < iters; ++it) {
            float v = 1.f;
            if (noise & 4) v = x[(base * 7 + (long long)it * 4096) & 0xFFFFF];
            if (noise & 2) lds[(threadIdx.x * 4 + it) & 8191] = v;
            if (noise & 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, acc, 0, 0, 0);
            }
            if (noise & 2) g += lds[(threadIdx.x * 8 + 3 * it) & 8191];
            if (noise & 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fa, acc, 0, 0, 0);
            if (noise & 8) { g = g * 1.0001f + v; g = g * 0.9999f - v; g = fmaxf(g, v); g += 0.5f * v; }      // plain VALU
        }
        float s = g;
        for (int r = 0; r < 16; ++r) s += acc[r];
        if (s == 12345.678f) sink[0] = s;
        return;
    }
    unsigned cnt[NFORMS][2] = {};
    unsigned quarter[4] = {};
    for (int it = 0; it < iters; ++it) {
        const long long e = ((base + (long long)it * 1048573) & 0x7FFFF) * 2;
        // operands straight from memory, as in the kernel (no VALU writes them before the packed instruction)
        const u64 a = *reinterpret_cast<const u64 *>(x + e);
        const u64 m = *reinterpret_cast<const u64 *>(mk + e);
        const float ax = __builtin_bit_cast(float, (unsigned)a), ay = __builtin_bit_cast(float, (unsigned)(a >> 32));
        const float mx = __builtin_bit_cast(float, (unsigned)m), my = __builtin_bit_cast(float, (unsigned)(m >> 32));
        unsigned got[NFORMS][2];
        PK(0, "")
        PK(1, "op_sel:[0,1] op_sel_hi:[1,0]")
        PK(2, "op_sel:[1,0] op_sel_hi:[0,1]")
        PK(3, "op_sel:[1,0]")
        PK(4, "op_sel:[0,1]")
        PK(5, "op_sel_hi:[1,0]")
        PK(6, "op_sel_hi:[0,1]")
        PKI(7, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]")
        PKI(8, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]")
        PKI(9, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0]")
        PKI(10, "v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0]")
        PKI(11, "v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1]")
        PKI(12, "v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0]")
        PKI(13, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]")
        PKI(14, "v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]")
        PKI(15, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]")
        float ll, lh, hl, hh;      // a.lo * m.lo, a.lo * m.hi, a.hi * m.lo, a.hi * m.hi by scalar multiplies
        asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(ll) : "v"(ax), "v"(mx));
        asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(lh) : "v"(ax), "v"(my));
        asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(hl) : "v"(ay), "v"(mx));
        asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(hh) : "v"(ay), "v"(my));
        float s_lh, s_hh, s_hl, f_lhl, f_hhh, f_llh, f_hll;      // sums and fused multiply-adds by scalar instructions
        asm volatile("v_add_f32 %0, %1, %2" : "=&v"(s_lh) : "v"(ax), "v"(my));
        asm volatile("v_add_f32 %0, %1, %2" : "=&v"(s_hh) : "v"(ay), "v"(my));
        asm volatile("v_add_f32 %0, %1, %2" : "=&v"(s_hl) : "v"(ay), "v"(mx));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(f_lhl) : "v"(ax), "v"(my), "v"(ax));      // a.lo * m.hi + a.lo
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(f_hhh) : "v"(ay), "v"(my), "v"(ay));      // a.hi * m.hi + a.hi
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(f_llh) : "v"(ax), "v"(mx), "v"(ay));      // a.lo * m.lo + a.hi
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(f_hll) : "v"(ay), "v"(mx), "v"(ax));      // a.hi * m.lo + a.lo
        // (v_pk_mov_b32: low result = src0 half op_sel[0], high result = src1 half op_sel[1]; the noise-free run calibrates these expectations)
        const float want[NFORMS][2] = {{ll, hh}, {lh, hl}, {hl, lh}, {hl, hh}, {lh, hh}, {ll, hl}, {ll, lh},
                                       {s_lh, s_hl}, {s_lh, s_hh}, {s_hl, s_hh}, {f_lhl, f_hhh}, {f_llh, f_hhh}, {f_hll, f_hhh},
                                       {ay, mx}, {ax, my}, {ay, my}};
#pragma unroll
        for (int f = 0; f < NFORMS; ++f) {
            const bool b0 = got[f][0] != __builtin_bit_cast(unsigned, want[f][0]);
            const bool b1 = got[f][1] != __builtin_bit_cast(unsigned, want[f][1]);
            cnt[f][0] += b0;
            cnt[f][1] += b1;
            if (b0 || b1) quarter[lane >> 4] += 1;      // (counted per lane: the lane's own quarter)
        }
    }
    for (int f = 0; f < NFORMS; ++f)
        for (int h = 0; h < 2; ++h)
            if (cnt[f][h]) atomicAdd(bad + f * 2 + h, cnt[f][h]);
    if (quarter[lane >> 4]) atomicAdd(bad + 2 * NFORMS + (lane >> 4), quarter[lane >> 4]);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const size_t n = (size_t)1 << 21;
    std::vector<float> hx(n), hm(n);
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((seed >> 8) & 0xFFFF) / 65536.f - 0.5f; };
    for (size_t i = 0; i < n; ++i) { hx[i] = rnd() * 4.f; hm[i] = 1.f + (float)(i & 7); }
    float *x, *mk, *sink;
    unsigned *bad;
    hipMalloc(&x, n * 4); hipMalloc(&mk, n * 4); hipMalloc(&sink, 64); hipMalloc(&bad, 64 * 4);
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(mk, hm.data(), n * 4, hipMemcpyHostToDevice);
    for (int noise : {0, 1, 14, 7}) {      // bit 0: MFMA, bit 1: LDS, bit 2: global loads, bit 3: plain VALU in the partner wave
        for (int blocks : {2048}) {
            hipMemset(bad, 0, 64 * 4);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, x, mk, bad, sink, iters, noise);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            unsigned hb[64];
            hipMemcpy(hb, bad, 64 * 4, hipMemcpyDeviceToHost);
            const double total = (double)blocks * 4 * 64 * iters;
            printf("noise %d, %d workgroups of 8 waves, %d iterations (%.3g packed multiplies per form):\n", noise, blocks, iters, total);
            for (int f = 0; f < NFORMS; ++f)
                printf("   %-52s wrong low results %10u   wrong high results %10u\n", FORM_NAME[f], hb[2 * f], hb[2 * f + 1]);
            printf("   wrong results by quarter of the wave (lanes 0-15 / 16-31 / 32-47 / 48-63): %u / %u / %u / %u\n", hb[2 * NFORMS], hb[2 * NFORMS + 1],
                   hb[2 * NFORMS + 2], hb[2 * NFORMS + 3]);
        }
    }
    return 0;
}
