// conv_ktap.hip -- conv_ktap_kernel<KT, ACT>: the 128 x 256 tile of the split-f16 x3 conv engine (conv_split_kernel<1, 8, 4, 1, 3>, conv_split.hip)
// with the tap count a template constant and the whole 16-channel chunk -- KT taps x 8 column tiles x 3 cross products -- as ONE straight-line
// block in which every instruction that is not an MFMA sits in the shadow of one.  (Reference work: the stride-1 Conv1d's of the HiFi-GAN
// generator's ResBlock1 / conv_pre, modules/visinger/decoder.py:72-101, 40-43, and the FFN's conv_1, modules/rel_transformer.py:336-345.)
//
// Why (rounds 4-5): a wave of the tile kernel alone on its SIMD keeps the matrix pipe 48 % busy -- per (chunk, tap) step 768 cycles
// of MFMA issue and ~840 cycles of everything else (the staging of the next chunk: load, input transform, tile maximum, scale, split into two f16
// planes, LDS writes; the weight-fragment requests; the wait tree; ~45 SALU and ~14 branches of step bookkeeping), one AFTER the other: the staging sits
// in a conditional block of the first tap, and between the three MFMAs of a column tile there are two LDS reads and nothing else.  An MFMA
// 32x32x16 occupies the pipe for 32 cycles and the wave's issue port for 8: up to five or six other instructions issue for free behind each
// (MI355X_MICROARCH.md, "vector-instruction ISSUE cost").  Here
//   * the taps are unrolled (KT = 3 / 7 / 11 ...): a chunk body has no branch, no run-time tap bookkeeping, LDS offsets are immediates;
//   * the staging of chunk c + 1 is cut into ~55 micro-operations of 3 - 5 instructions (one value's transform, half a pair's split, one row
//     of loads ...) that are dealt out over the 3 * 8 * KT MFMA gaps of chunk c by a compile-time schedule, in program order, pinned by
//     scheduling barriers;
//   * nothing in the loop is predicated: columns outside the tensor or the staged window and chunks past the last one are buffer loads the
//     descriptor's range check answers with zeros (no memory traffic), the running tile maximum ignores zeros, rescales are a multiply by one;
//   * every wait is hipcc's own count on a straight-line scoreboard (no hand-counted vmcnt, no wait tree).
// Arithmetic, operand planes, MFMA order and epilogue are those of conv_split_kernel<1, 8, 4, 1, 3>: outputs are BIT-IDENTICAL to it
// (tests/test_conv_ktap_gpu.py), so every parity statement about that instance carries over.
#include "conv_ktap.inc"

namespace vs {

// ---------------------------------------------------------------------------------------------------------------- host side (split-f16 x3 instances)
// the instances that exist.  cfg: the tile shape vs_conv_forward chose (0: 128 x 256, 1 / 3: 64 x 256, 6: 32 x 128); io: bit 0 -- x, bit 1 -- y / res / acc hold bf16 elements.
//   terms 3 (split-f16 x3, fp32 tensors), 128 x 256: 3 / 7 / 11 taps with every input transform, 9 taps (FFN conv_1) unmasked or masked;
//                                         64 x 256 and 32 x 128 (conv_ktap_small.hip): 1 / 2 / 3 / 5 / 7 / 9 / 11 taps untransformed (the training step's forward and
//                                         grad-input convs, the transformers' projections), 1 / 9 taps masked (FFN, modules/rel_transformer.py:336-345);
//   terms 1 (plain bf16), 128 x 256: fp32 tensors -- 1 / 9 taps (transformer: none / mask), 3 / 11 taps (generator: lrelu, lrelu + mask), 7 taps (all four);
//                                    bf16 in and out -- 3 / 7 / 11 taps behind lrelu (+ mask);
//                         64 x 256 and 32 x 128: fp32 tensors, 1 / 9 taps, none / mask.
bool ktap_instance(int terms, int cfg, int kt, int io, int in_act) {
    const bool plain = (in_act == VS_IN_NONE || in_act == VS_IN_MASK);
    const bool small = (cfg == 1 || cfg == 2 || cfg == 3 || cfg == 6);
    if (!small && cfg != 0) return false;
    if (terms == 3) {
        if (io != 0) return false;
        if (small) return (in_act == VS_IN_NONE && (kt == 1 || kt == 2 || kt == 3 || kt == 5 || kt == 7 || kt == 9 || kt == 11)) || (in_act == VS_IN_MASK && (kt == 1 || kt == 9)) ||
                          (in_act == VS_IN_LRELU && (cfg == 1 || cfg == 3) && (kt == 3 || kt == 7 || kt == 11));
        return kt == 3 || kt == 7 || kt == 11 || (kt == 9 && plain);
    }
    if (terms != 1) return false;
    if (small) return cfg != 2 && io == 0 && plain && (kt == 1 || kt == 9);
    if (io == 3) return (kt == 3 || kt == 7 || kt == 11) && !plain;
    if (io != 0) return false;
    return kt == 7 || ((kt == 1 || kt == 9) && plain) || ((kt == 3 || kt == 11) && !plain);
}

template <int KT>
static int launch_ktap_kt(const ConvParams &p, hipStream_t s) {
    switch (p.in_act) {
        case VS_IN_NONE: return launch_ktap_inst<KT, VS_IN_NONE, 2, 0>(p, s);
        case VS_IN_LRELU: return launch_ktap_inst<KT, VS_IN_LRELU, 2, 0>(p, s);
        case VS_IN_MASK: return launch_ktap_inst<KT, VS_IN_MASK, 2, 0>(p, s);
        default: return launch_ktap_inst<KT, VS_IN_LRELU_MASK, 2, 0>(p, s);
    }
}

// p as for launch_split(cfg 0) on a plain stride-1 conv of the split-f16 arithmetic with C_in a multiple of 16 and ktap_instance(3, ...)
int launch_ktap(const ConvParams &p, int cfg, hipStream_t s) {
    if (cfg != 0) return launch_ktap_small(p, cfg, s);
    if (!ktap_geometry_ok(p) || p.x_bf16 || p.y_bf16) {
        set_error("launch_ktap: not a plain stride-1 conv of whole 16-channel chunks on fp32 tensors");
        return VS_EUNSUPPORTED;
    }
    switch (p.KT) {
        case 3: return launch_ktap_kt<3>(p, s);
        case 7: return launch_ktap_kt<7>(p, s);
        case 9:       // (FFN conv_1, modules/rel_transformer.py:336-345: masked input only)
            if (p.in_act == VS_IN_MASK) return launch_ktap_inst<9, VS_IN_MASK, 2, 0>(p, s);
            if (p.in_act == VS_IN_NONE) return launch_ktap_inst<9, VS_IN_NONE, 2, 0>(p, s);
            break;
        case 11: return launch_ktap_kt<11>(p, s);
        default: break;
    }
    set_error("launch_ktap: no instance for %d taps with input transform %d", p.KT, p.in_act);
    return VS_EUNSUPPORTED;
}

// ---- transposed convs (the generator's upsamplers, modules/visinger/decoder.py:36-48: nn.ConvTranspose1d(k = 2 * stride) behind a leaky-relu): conv_ktap.inc IO bit 2.
// Instance: 128 x 256 tiles, two taps per phase, fp32 tensors, leaky-relu input transform.  Every phase must use exactly two of the packed taps, a 32-row
// tile must lie inside one phase and the four row tiles of a workgroup must exist.
bool ktap_tr_instance(const ConvParams &p, int terms, int cfg) {
    if (terms != 3 || cfg != 0 || p.kind != VS_CONV_TRANSPOSE1D || p.Cin % CK != 0 || p.tstep != -1 || p.KT - 1 > MAX_SPAN || p.x_bf16 || p.y_bf16) return false;
    if (p.in_act != VS_IN_LRELU || p.bias_b || p.split_row || (p.c_out % 32) != 0 || (p.MT % 4) != 0 || p.KT < 2 || p.KT > 3) return false;
    for (int phase = 0; phase < p.up; ++phase) {        // (the tap range of conv_split_body.inc / conv_ktap.inc, per phase)
        const int num_lo = -(phase + p.uppad), num_hi = p.upK - 1 - phase - p.uppad;
        const int dlo = (num_lo >= 0) ? (num_lo + p.up - 1) / p.up : -((-num_lo) / p.up);
        const int dhi = (num_hi >= 0) ? num_hi / p.up : -((-num_hi + p.up - 1) / p.up);
        if (dhi - dlo + 1 != 2 || dlo - p.dmin < 0 || dhi - p.dmin >= p.KT) return false;
    }
    return true;
}

int launch_ktap_tr(const ConvParams &p, hipStream_t s) {
    if (!ktap_tr_instance(p, 3, 0)) {
        set_error("launch_ktap_tr: not a two-taps-per-phase transposed conv of whole 16-channel chunks on fp32 tensors");
        return VS_EUNSUPPORTED;
    }
    return launch_ktap_inst<2, VS_IN_LRELU, 2, 4>(p, s);      // (the instance without an input transform -- the training forward's -- spills 728 bytes per lane: not built)
}

}  // namespace vs
