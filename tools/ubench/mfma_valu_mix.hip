// Micro-benchmark: does VALU work issued BETWEEN the MFMAs of the same wave cost matrix throughput on gfx950?  The conv engine's operand
// pattern (v_mfma_f32_32x32x16_f16, every accumulator a chain of three dependent MFMAs, eight accumulator tiles, two workgroups of four
// waves per CU) with V independent v_fma_f32 per MFMA placed behind it.  Reports the MFMA rate and the shader clock per V.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_mix.hip -o tools/ubench/mfma_valu_mix && tools/ubench/mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V, int KIND>   // KIND 0: v_fma_f32, 1: v_cvt_pk (f16 conversions), 2: ds_write_b64 every MFMA (V ignored)
__global__ void __launch_bounds__(256, 2) k(const u32x4 *in, float *out, unsigned long long *clk, int iters) {
    __shared__ unsigned lds[256 * 8];
    u32x4 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { a[i] = in[(threadIdx.x * 7 + i * 131 + blockIdx.x) & 4095]; b[i] = in[(threadIdx.x * 11 + i * 977 + 5 * blockIdx.x) & 4095]; }
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 1.f + threadIdx.x * 1e-3f + i;
    const float c = 0.999f, d = 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[t & 1]), __builtin_bit_cast(f16x8, b[(t + j) & 1]), acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(c), "v"(d));
                    else if constexpr (KIND == 1) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(c));
                }
                if constexpr (KIND == 2) {
                    if (V > 0 && t == 0) *reinterpret_cast<uint2 *>(&lds[threadIdx.x * 8 + 2 * (j & 3)]) = make_uint2(__builtin_bit_cast(unsigned, x[j]), 1u);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s + (KIND == 2 ? (float)lds[threadIdx.x] : 0.f);
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int V, int KIND>
static void run(const u32x4 *in, float *out, unsigned long long *clk, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<V, KIND>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<V, KIND>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    const double flop = (double)reps * blocks * 4.0 * iters * 24.0 * 32768.0;
    const double mf_per_simd = (double)iters * 24.0 * 2.0;      // two waves per SIMD
    printf("%s x %d per MFMA: %7.1f TFLOP/s of MFMA work, shader clock %.3f GHz, %.1f cycles per MFMA and SIMD\n",
           KIND == 0 ? "v_fma_f32" : (KIND == 1 ? "v_cvt_pk_f16_f32" : "ds_write_b64 (1 per 3 MFMAs)"), V, flop / (ms * 1e-3) / 1e12,
           cyc / real / 10.0, (cyc / blocks) / mf_per_simd);
}

int main() {
    const int blocks = 512, iters = 2000;
    u32x4 *in; float *out; unsigned long long *clk;
    hipMalloc(&in, 4096 * 16); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    std::vector<unsigned short> h(4096 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 22 & 0x3ff);     // f16 in [1, 2)
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0, 0>(in, out, clk, blocks, iters);
    run<1, 0>(in, out, clk, blocks, iters);
    run<2, 0>(in, out, clk, blocks, iters);
    run<4, 0>(in, out, clk, blocks, iters);
    run<6, 0>(in, out, clk, blocks, iters);
    run<8, 0>(in, out, clk, blocks, iters);
    run<2, 1>(in, out, clk, blocks, iters);
    run<4, 1>(in, out, clk, blocks, iters);
    run<1, 2>(in, out, clk, blocks, iters);
    return 0;
}
