// conv_ktap_small.hip -- the split-f16 x3 instances of conv_ktap_kernel (conv_ktap.inc; design notes in conv_ktap.hip) on the 64 x 256 and 32 x 128 tiles: the convs of
// T_mel-sized tensors (the prior transformers' projections and FFNs, modules/rel_transformer.py:120-134, 336-345; the flow's 1 x 1 convs) and the short launches of
// the GAN training step (forward and grad-input convs of every module, the discriminators' phase-stacked convs: tasks/visinger.py:53-89).  Their grids must cover the
// chip, so a workgroup's tile is small and a launch lasts as long as ONE workgroup's walk over C_in * k: with conv_split_kernel<1, 1, 1, 4, 3> that walk took ~1 600
// cycles per (chunk, tap) step for three MFMAs per wave (446 launches x 57 us = 25 of the training step's 100 ms, round 4).  Here a chunk is one straight-line block
// of 3 * k MFMAs per wave beside ~110 staging instructions.  Bit-identical to conv_split_kernel<1, 4, 2, 2, 3> / <1, 1, 1, 4, 3>.
#include "conv_ktap.inc"

namespace vs {

template <int WM, int WN, int NT>
static int launch_ktap_small_tile(const ConvParams &p, hipStream_t s) {
    if (p.in_act == VS_IN_LRELU && NT == 4) {          // (the generator's 64-channel resblock convs conv by conv: modules/visinger/decoder.py:91-104)
        if (p.KT == 3) return launch_ktap_inst<3, VS_IN_LRELU, 2, 0, WM, WN, NT>(p, s);
        if (p.KT == 7) return launch_ktap_inst<7, VS_IN_LRELU, 2, 0, WM, WN, NT>(p, s);
        if (p.KT == 11) return launch_ktap_inst<11, VS_IN_LRELU, 2, 0, WM, WN, NT>(p, s);
    }
    if (p.in_act == VS_IN_MASK) {
        if (p.KT == 1) return launch_ktap_inst<1, VS_IN_MASK, 2, 0, WM, WN, NT>(p, s);
        if (p.KT == 9) return launch_ktap_inst<9, VS_IN_MASK, 2, 0, WM, WN, NT>(p, s);
    } else if (p.in_act == VS_IN_NONE) {
        switch (p.KT) {
            case 1: return launch_ktap_inst<1, VS_IN_NONE, 2, 0, WM, WN, NT>(p, s);
            case 2: return launch_ktap_inst<2, VS_IN_NONE, 2, 0, WM, WN, NT>(p, s);
            case 3: return launch_ktap_inst<3, VS_IN_NONE, 2, 0, WM, WN, NT>(p, s);
            case 5: return launch_ktap_inst<5, VS_IN_NONE, 2, 0, WM, WN, NT>(p, s);
            case 7: return launch_ktap_inst<7, VS_IN_NONE, 2, 0, WM, WN, NT>(p, s);
            case 9: return launch_ktap_inst<9, VS_IN_NONE, 2, 0, WM, WN, NT>(p, s);
            case 11: return launch_ktap_inst<11, VS_IN_NONE, 2, 0, WM, WN, NT>(p, s);
            default: break;
        }
    }
    set_error("launch_ktap_small: no instance for %d taps with input transform %d", p.KT, p.in_act);
    return VS_EUNSUPPORTED;
}

// cfg as launch_split's: 1 / 3 -> 64 x 256, 2 -> 32 x 256, 6 -> 32 x 128
int launch_ktap_small(const ConvParams &p, int cfg, hipStream_t s) {
    if (!ktap_geometry_ok(p) || p.x_bf16 || p.y_bf16) {
        set_error("launch_ktap_small: not a plain stride-1 conv of whole 16-channel chunks on fp32 tensors");
        return VS_EUNSUPPORTED;
    }
    if (cfg == 6) return launch_ktap_small_tile<1, 4, 1>(p, s);
    if (cfg == 2) return launch_ktap_small_tile<1, 4, 2>(p, s);
    return launch_ktap_small_tile<2, 2, 4>(p, s);
}

}  // namespace vs
