// Micro-benchmark: v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 under the conv engine's operand pattern (random bf16
// operands in registers, 128 accumulator registers per wave, six cross products per accumulator tile), whole chip, two workgroups
// per CU, with the DISTANCE between two MFMAs on the same accumulator as the parameter (DIST = 1: six back-to-back dependent MFMAs
// per tile as conv_split_kernel issues them; 2 / 4: tiles interleaved so that a tile's next MFMA is the 2nd / 4th after it).
// Reports TFLOP/s by wall clock and the shader clock the loop ran at (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shape_rate.hip -o /tmp/mfma_shape_rate && /tmp/mfma_shape_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int DIST>   // SHAPE 0: 32x32x16, 1: 16x16x32
__global__ void __launch_bounds__(256, 2) k(const u32x4 *in, float *out, unsigned long long *clk, int iters) {
    u32x4 a[6], b[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = in[(threadIdx.x * 7 + i * 131 + blockIdx.x) & 4095];
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = in[(threadIdx.x * 11 + i * 977 + 5 * blockIdx.x) & 4095];
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
            for (int j0 = 0; j0 < 8; j0 += DIST)
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int j = j0; j < j0 + DIST; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t % 3]), __builtin_bit_cast(bf16x8, b[(t + j) % 3]), acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[j][r];
    } else {
        f32x4 acc[32];     // the same 128 accumulator registers: 32 tiles of 16x16 = 2 row tiles x 16 column tiles
#pragma unroll
        for (int j = 0; j < 32; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j0 = 0; j0 < 32; j0 += DIST)
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int j = j0; j < j0 + DIST; ++j)   // tile j: row tile j & 1 (its own A fragments), column tile j >> 1
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(t % 3) + 3 * (j & 1)]), __builtin_bit_cast(bf16x8, b[(t + (j >> 1)) % 3]), acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[j][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int DIST>
static void run(const char *name, const u32x4 *in, float *out, unsigned long long *clk, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, DIST>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SHAPE, DIST>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    const double flop = (double)reps * blocks * 4 /*waves*/ * iters * 48.0 * 32768.0;   // 48 MFMAs of 32x32x16 = 192 of 16x16x32 per iteration
    printf("%-9s dist %d  %8.1f TFLOP/s   shader clock %.3f GHz   %.1f cycles per 32768 FLOP per wave\n", name, DIST,
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
        run<0, 1>("32x32x16", in, out, clk, blocks, iters);
        run<0, 2>("32x32x16", in, out, clk, blocks, iters);
        run<1, 1>("16x16x32", in, out, clk, blocks, iters);
        run<1, 2>("16x16x32", in, out, clk, blocks, iters);
        run<1, 4>("16x16x32", in, out, clk, blocks, iters);
        run<1, 8>("16x16x32", in, out, clk, blocks, iters);
    }
    return 0;
}
