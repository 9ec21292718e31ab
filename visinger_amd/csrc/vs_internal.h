// Internal helpers shared by the HIP translation units of libvisinger_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/visinger_hip.h"

namespace vs {

void set_error(const char *fmt, ...);
// name of the kernel instance the calling thread launched last ("conv_split_kernel<1, 8, 4, 1, 6>": template arguments as
// rocprofv3 prints them) -- read back through vs_last_kernel_name() by the bench's per-launch attribution
void set_last_kernel(const char *fmt, ...);

// Dispatch switches (A/B and debug): ONE table, initialised from the environment variables of the same names when the library is
// loaded and changed afterwards only through vs_set_option() -- no getenv on any launch path.  (INTEGRATION.md lists them.)
enum Opt {
    OPT_CONV_MATH,        // VS_CONV_MATH: arithmetic of conv handles created from now on (0 / 1 / 6), -1 = library default
    OPT_NO_SMALL_CONV,    // VS_NO_SMALL_CONV: convs with <= 4 output channels on the MFMA tiles instead of the VALU kernel
    OPT_NO_FAST_EPI,      // VS_NO_FAST_EPI: element-wise epilogue everywhere
    OPT_WINO_FORCE,       // VS_WINO_FORCE: fp32 engine: F(2,3) on every eligible conv
    OPT_NO_WINO,          // VS_NO_WINO: fp32 engine: never F(2,3)
    OPT_NO_WINO_K7,       // VS_NO_WINO_K7: fp32 engine: no k-specialised F(2,3) instances (read at vs_conv_create)
    OPT_WINO_DBG,         // VS_WINO_DBG: timing-only perturbations of conv_wino_kernel
    OPT_NO_SMALL_GRID,    // VS_NO_SMALL_GRID: keep the 128-row tile on launches that do not cover the chip
    OPT_SMALL_GRID_T6,    // VS_SMALL_GRID_T6: workgroup count below which 32 x 128 tiles are taken (default 512)
    OPT_CONV_CFG,         // VS_CONV_CFG: force a tile shape of the direct engine (-1 = automatic)
    OPT_SPLIT_DBG,        // VS_SPLIT_DBG: timing-only perturbations of conv_split_kernel (-DVS_SPLIT_PERTURB builds)
    OPT_TRACE,            // VS_TRACE: one line per conv launch on stderr
    OPT_NO_BF16_ATTN,     // VS_NO_BF16_ATTN: VS_MATH_BF16 attention on the exact-fp32 kernel
    OPT_NO_SPLIT_ATTN,    // VS_NO_SPLIT_ATTN: VS_MATH_SPLIT6 attention on the exact-fp32 kernel
    OPT_NO_WGRAD_SPLIT,   // VS_NO_WGRAD_SPLIT: weight gradients on the exact-fp32 kernel only
    OPT_RB_TILE256,       // VS_RB_TILE256: whole-resblock launch at 32 channels on 256-column tiles (default 512: half the weight-fragment traffic and halo per output)
    OPT_NO_ATTN_KVPACK,   // VS_NO_ATTN_KVPACK: the plain-bf16 attention kernel converts K / V tiles in place (no pre-packed images)
    OPT_NO_TR_EPI,        // VS_NO_TR_EPI: transposed convs on the generic instances (element-wise polyphase stores)
    OPT_NO_KTAP,          // VS_NO_KTAP: the wide stride-1 split-f16 convs on the tile kernel conv_split_kernel<1, 8, 4, 1, 3> instead of conv_ktap_kernel (taps unrolled, staging in the MFMA shadows; bit-identical: A/B switch)
    OPT_NO_ATTN_DMA,      // VS_NO_ATTN_DMA: wide-head plain-bf16 attention on relattn_bf16_kernel<.., true> (register-staged tiles) instead of relattn_dma_kernel
    OPT_ATTN_DMA_ONE_WAVE,  // VS_ATTN_DMA_ONE_WAVE: relattn_dma_kernel with one wave per query group (the round-4 form) instead of the wave pair that splits the head's channels
    OPT_ATTN_SPLIT6,      // VS_ATTN_SPLIT6: VS_MATH_SPLIT3 attention on the split-bf16 x6 instance (the form of rounds 3-5) instead of relattn_bf16_kernel<.., 3> (split-f16 x3)
    OPT_NO_T1_CONV,       // VS_NO_T1_CONV: 1 x 1 convs over a single frame per item (conditioning vectors) on the tile kernels instead of conv_t1_kernel
    OPT_COUNT
};
long long opt(Opt o);

#define VS_CHECK_HIP(expr)                                                                             \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            vs::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_));       \
            return VS_EHIP;                                                                            \
        }                                                                                              \
    } while (0)

#define VS_REQUIRE(cond, ...)             \
    do {                                  \
        if (!(cond)) {                    \
            vs::set_error(__VA_ARGS__);   \
            return VS_EINVAL;             \
        }                                 \
    } while (0)

#define VS_TRY(expr)                 \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != VS_OK) return rc_; \
    } while (0)

// Device buffer owned by a handle (grown, never shrunk; hipMalloc happens off the steady-state path).
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int reserve(size_t n) {
        if (n <= bytes) return VS_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
            return VS_ENOMEM;
        }
        bytes = n;
        return VS_OK;
    }
    template <typename T>
    T *as() const { return reinterpret_cast<T *>(p); }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace vs

// The conv handle of the C ABI (vs_conv_t), shared by conv_engine.hip (which owns it) and resblock_pair.hip (which runs two
// of them in one launch).
struct vs_conv {
    int kind, c_in, c_out, k, dil, pad;   // dil = stride for transposed
    unsigned flags;
    int M, MT, MT_alloc, KT, CP, nchunks, off0, tstep, lo, span, dmin, Hh;
    bool weights_set = false;
    vs::DevBuf wp, biasp, scale, weff, beff;   // weff/beff: unpacked effective weights (c_out <= 4 VALU path)
    vs::DevBuf wpw;                            // Winograd-domain fragments (conv_wino_kernel), built from wp by the first launch that uses them
    bool wino_packed = false;                  // wpw holds the transform of the current weight version
    int wino_groups = 0;                       // ceil(k / 3) if the conv is eligible for the F(2,3) path, else 0
    bool wino_k7 = false;                      // k = 7 on an even tile count: the TG = 3 instances (direct-form last tap)
    bool wino_k11 = false;                     // k = 11: the TG = 4 instances (F(2,2) last group, 2-slot ring)
    bool has_bias = false;
    int math = 0;                              // 0: fp32 MFMA / F(2,3); 6: split-bf16 x6 (fp32 class); 3: split-f16 x3 (scaled, fp32 class); 1: bf16 (conv_split.hip)
    vs::DevBuf ws;                             // bf16 / f16 plane fragments of the split engine, when math != 0
    vs::DevBuf wsc;                            // split-f16 arithmetic (math == 3): {s_w, 1 / s_w, max |w| bits of even / odd packs} of the packed planes
    int pack_gen = 0;                          // weight versions packed in that arithmetic (selects the max slot)
    vs::DevBuf ldpart;                         // PAIRED coupling forward: per-tile log-det partials (fixed-order reduction, no atomics)
};


// resblock_pair_split.hip: the fused residual pair on the split-bf16 x6 arithmetic (both handles in VS_MATH_SPLIT6)
int vs_respair_split_launch(const vs_conv *c1, const vs_conv *c2, const vs_conv_io_t *io, int fast_epi, hipStream_t s);
