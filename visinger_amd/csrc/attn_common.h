// attn_common.h -- launch parameters shared by the attention kernels of libvisinger_hip.so
// (transformer_ops.hip: exact-fp32 MFMA; attention_bf16.hip: bf16 MFMA for VS_MATH_BF16).
#pragma once
#include "vs_internal.h"

namespace vs {

constexpr int ATT_MAXREL = 16;   // 2*window+1 <= 16
constexpr int ATT_QRS = 17;      // LDS stride of the per-query relative rows

struct AttnParams {
    const float *q, *k, *v;
    long long bs;                 // batch stride of q/k/v (floats)
    const float *rel_k, *rel_v;   // [nh_rel, 2ws+1, dk] or null
    const float *mask;            // [B, T] or null
    float *out;
    long long out_bs;
    int B, nh, dk, T, ws, nh_rel;
    float scale;
    // key split (attention_bf16.hip, single utterances): the keys of a (batch, head) are cut into `ksplit` ranges, each workgroup writes
    // its un-normalised output rows, row maximum, row sum and in-window scores to `part`, a second kernel combines them
    float *part;                  // [B * nh * ksplit][dk + 2 + nrel][T], or null
    int ksplit;                   // >= 2 with part, else 0 / 1
    // pre-packed keys / values (attention_bf16.hip, plain-bf16 arithmetic, long sequences): the LDS image of every 32- / 64-key tile
    // -- K as [d/8][key][8 bf16], V as [d][permuted keys] + row padding, the tile's key mask -- built ONCE per launch by
    // attn_pack_kv_kernel instead of once per query block (T / 128 times) inside the attention kernel; null: convert in the kernel
    unsigned *kvimg;              // [B][nh][key tiles][attn_kv_image_dwords(dk)] or null
    unsigned long long *stamps;   // debug (vs_debug_set_stamp_buffer): shader-clock stamps of workgroup (0, 0, 0), wave 0; null in production
};

// bytes of the pre-packed K / V images for a launch of the plain-bf16 kernel, 0 where the kernel converts in place (short sequences,
// shapes attention_bf16.hip does not take)
size_t attn_kv_work_bytes(long long B, int nh, int dk, long long T);

// attention_bf16.hip: both GEMMs on the bf16 matrix instruction, fp32 softmax / accumulation; terms = 1: bf16 operands (dk <= 256),
// terms = 6: the exact three-plane split with six cross products (fp32 class, dk <= 128); terms = 3: two f16 planes under power-of-two
// scales, three cross products on the f16 matrix instruction (fp32 class, dk <= 128).  Needs T % 4 == 0 and 16-byte aligned
// q / k / v rows: when attn_bf16_supported() says no, the caller runs the exact-fp32 MFMA kernel.
bool attn_bf16_supported(const AttnParams &p, int terms);
int launch_attn_bf16(const AttnParams &p, int terms, hipStream_t s);
int launch_attn_combine(const AttnParams &p, hipStream_t s);
// attention_dma.hip (round 4): the pre-packed plain-bf16 kernel for heads of 129 .. 256 channels with the tile images brought into an LDS
// ring by LDS-DMA; p.kvimg as written by attn_pack_kv_kernel<DT, 32> (launch_attn_bf16 runs the pack, then asks here)
bool attn_dma_supported(const AttnParams &p);
int launch_attn_dma(const AttnParams &p, hipStream_t s);

}  // namespace vs
