// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 for one wave per SIMD, with and without LDS operand reads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ void __launch_bounds__(256, 2) k(const float* in, float* out, long long* cyc, int iters) {
    __shared__ float lds[16 * 512];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 16 * 512; e += 256) lds[e] = in[e & 1023];
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = in[lane];
    float bf[8], bn[8];
    const float* xs = lds + (lane >> 5) * 512 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 8; ++j) bf[j] = xs[j * 32];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int cp = 0; cp < 8; ++cp) {
            if (VARIANT >= 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bn[j] = xs[((cp + 1) & 7) * 2 * 512 / 2 + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (VARIANT >= 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bf[j] = bn[j];
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// V2: the conv kernel's step structure: 8 different A registers per step, loaded by asm global loads two steps ahead
template <int IMM>
__device__ __forceinline__ void gl(float &dst, const float *ptr) {
    asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void wv(float (&r)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(N) : "memory");
}
template <int VARIANT>
__global__ void __launch_bounds__(256, 2) k2(const float* in, float* out, long long* cyc, int iters) {
    __shared__ float lds[16 * 512];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 16 * 512; e += 256) lds[e] = in[e & 1023];
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a0[8], a1[8], a2[8];
    const float* wp = in + lane;
    auto la = [&](float (&d)[8], int s) __attribute__((always_inline)) {
        const float* p = wp + (s & 3) * 512;
        gl<0>(d[0], p); gl<256>(d[1], p); gl<512>(d[2], p); gl<768>(d[3], p); gl<1024>(d[4], p); gl<1280>(d[5], p); gl<1536>(d[6], p); gl<1792>(d[7], p);
    };
    la(a0, 0); la(a1, 1);
    const float* xs0 = lds + (lane >> 5) * 512 + (lane & 31);
    int s = 0;
    const int nsteps = iters;
    float bf[8];
    auto step = [&](float (&ac)[8], float (&ap)[8]) __attribute__((always_inline)) {
        if (VARIANT & 1) {
            int younger = (s + 1 < nsteps) ? 8 : 0;
            if (s + 2 < nsteps) { la(ap, s + 2); younger += 8; }
            if (younger >= 16) wv<16>(ac); else if (younger >= 8) wv<8>(ac); else wv<0>(ac);
        }
        const float* xs = xs0 + (s & 1) * 32;
        float bn[8];
        if (!(VARIANT & 4) || s == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bf[j] = xs[j * 32];
        }
#pragma unroll
        for (int cp = 0; cp < 8; ++cp) {
            if (cp + 1 < 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bn[j] = xs[(cp + 1) * 1024 + j * 32];
            } else if (VARIANT & 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bn[j] = xs0[((s + 1) & 1) * 32 + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[cp], bf[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) bf[j] = bn[j];
        }
        if ((VARIANT & 2) && (s % 3) == 2) __syncthreads();
        ++s;
    };
    long long t0 = __builtin_amdgcn_s_memtime();
    while (s < nsteps) {
        step(a0, a2);
        if (s < nsteps) step(a1, a0);
        if (s < nsteps) step(a2, a1);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run2(const char* name, int blocks, float* in, float* out, long long* cyc) {
    const int iters = 198;
    hipLaunchKernelGGL(k2<V>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-28s blocks=%4d  cycles/MFMA (wave) = %.1f\n", name, blocks, h[0] / ((double)iters * 64));
}

template <int V>
void run(const char* name, int blocks, float* in, float* out, long long* cyc) {
    const int iters = 200;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mf = (double)iters * 64;
    printf("%-28s blocks=%4d  cycles/MFMA (wave) = %.1f   wall %.1f us  -> %.1f TFLOP/s\n", name, blocks, h[0] / mf, ms * 1e3,
           blocks * 4.0 * mf * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    float *in, *out; long long* cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 2048 * 8);
    hipMemset(in, 0, 4096 * 4);
    run<0>("regs only, 1 wg/CU", 256, in, out, cyc);
    run<0>("regs only, 2 wg/CU", 512, in, out, cyc);
    run<1>("lds prefetch, 1 wg/CU", 256, in, out, cyc);
    run<1>("lds prefetch, 2 wg/CU", 512, in, out, cyc);
    run2<0>("steps only (fixed A), 1 wg", 256, in, out, cyc);
    run2<4>("steps, carried bf, 1 wg", 256, in, out, cyc);
    run2<1>("steps + asm ring, 1 wg", 256, in, out, cyc);
    run2<5>("steps + asm ring + carried bf", 256, in, out, cyc);
    run2<7>("  ... + barrier/3 steps", 256, in, out, cyc);
    run2<5>("steps + asm ring + carried, 2wg", 512, in, out, cyc);
    return 0;
}
