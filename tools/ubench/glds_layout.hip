// Where does an LDS-DMA (global_load_lds_dwordx4, M0 = LDS base) put lane i's 16 bytes?  Expected (cdna_hip_programming.md 5.7): base + 16 * i.
// hipcc --offload-arch=gfx950 -O2 glds_layout.hip -o glds_layout && ./glds_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *x, float *y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 2048; i += 64) smem[i] = -1.f;
    __syncthreads();
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    unsigned keep;
    const char *src = (const char *)x;
    int off = (threadIdx.x * 7 % 64) * 16;          // lane l reads global element group (7 l mod 64)
    unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + 1024);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(src), "s"(dst) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) y[i] = smem[i];
}
int main() {
    float *x, *y, hx[256], hy[2048];
    for (int i = 0; i < 256; ++i) hx[i] = (float)i;
    hipMalloc(&x, sizeof(hx)); hipMalloc(&y, sizeof(hy));
    hipMemcpy(x, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, x, y);
    hipMemcpy(hy, y, sizeof(hy), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) ok &= (hy[256 + 4 * l + e] == (float)((l * 7 % 64) * 4 + e));
    for (int i = 0; i < 256; ++i) ok &= (hy[i] == -1.f) && (hy[512 + i] == -1.f);
    printf("lane-linear 16-byte layout at M0 base: %s\n", ok ? "YES" : "NO");
    if (!ok) { for (int i = 240; i < 300; ++i) printf("%d:%g ", i, hy[i]); printf("\n"); }
    return ok ? 0 : 1;
}
