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
};

// attention_bf16.hip: bf16 operands, fp32 softmax / accumulation; needs T % 4 == 0, 16-byte aligned q / k / v rows, dk <= 256.
// Returns VS_EUNSUPPORTED without touching the error string when the shape does not qualify (the caller falls back to fp32).
bool attn_bf16_supported(const AttnParams &p);
int launch_attn_bf16(const AttnParams &p, hipStream_t s);

}  // namespace vs
