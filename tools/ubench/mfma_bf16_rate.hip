// Micro-benchmark: sustained rate of the two bf16 MFMA shapes of gfx950 under the conv engine's operand pattern (operands from
// random data held in registers, every accumulator a chain of six dependent MFMAs as in conv_split_kernel), whole chip, two
// workgroups per CU.  Reports TFLOP/s by wall clock and the shader clock the loop ran at (s_memtime / s_memrealtime).
// Measured (MI355X, this pool): 32x32x16 1.56-1.57 PFLOP/s at 1.66-1.67 GHz, 31.4 cycles per MFMA and SIMD -- the pipe is saturated
// and the clock is what the power limit leaves of 2.4 GHz: the PRACTICAL ceiling of a bf16 MFMA kernel here is 0.63 of the spec
// peak.  (The 16x16x32 arm keeps the six-deep dependent chains, which that shape does not issue back to back: not a fair rate.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16_rate.hip -o tools/ubench/mfma_bf16_rate && tools/ubench/mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ void __launch_bounds__(256, 2) k(const u32x4 *in, float *out, unsigned long long *clk, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 a[3], b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { a[i] = in[(threadIdx.x * 7 + i * 131 + blockIdx.x) & 4095]; b[i] = in[(threadIdx.x * 11 + i * 977 + 5 * blockIdx.x) & 4095]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if constexpr (SHAPE == 0) {
        f32x16 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t % 3]), __builtin_bit_cast(bf16x8, b[(t + j) % 3]), acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[j][r];
    } else {
        f32x4 acc[32];     // the same 128 accumulator registers: 32 tiles of 16x16
#pragma unroll
        for (int j = 0; j < 32; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
#pragma unroll
                for (int t = 0; t < 6; ++t)     // 16x16x32 has half the FLOPs of 32x32x16: 32 tiles x 6 = the same work per iteration
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[t % 3]), __builtin_bit_cast(bf16x8, b[(t + j) % 3]), acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 32; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[j][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    (void)lane;
}

template <int SHAPE>
static void run(const char *name, const u32x4 *in, float *out, unsigned long long *clk, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    const double flop = (double)reps * blocks * 4 /*waves*/ * iters * 48.0 * 32768.0;   // 48 MFMAs of 32x32x16 (or 96 x 2 of 16x16x32... 192 x half)
    printf("%-10s  %8.1f TFLOP/s   shader clock %.3f GHz   %.1f cycles per 32768-FLOP MFMA-equivalent per wave\n", name,
           flop / (ms * 1e-3) / 1e12, cyc / rt * 0.1, cyc / blocks / ((double)iters * 48.0));
}

int main() {
    const int blocks = 512 * 8, iters = 400;
    std::vector<unsigned> hin(4096 * 4);
    srand(1);
    for (auto &v : hin) {       // random finite bf16 pairs
        unsigned lo = (rand() & 0x807f) | (((rand() % 16) + 120) << 7), hi = (rand() & 0x807f) | (((rand() % 16) + 120) << 7);
        v = lo | (hi << 16);
    }
    u32x4 *in; float *out; unsigned long long *clk;
    hipMalloc(&in, hin.size() * 4); hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, (size_t)blocks * 16);
    hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("32x32x16", in, out, clk, blocks, iters);
        run<1>("16x16x32", in, out, clk, blocks, iters);
    }
    return 0;
}
