// conv_ktap_bf16.hip -- the plain-bf16 instances of conv_ktap_kernel (conv_ktap.inc; design notes in conv_ktap.hip): operands rounded to bf16, one
// product per MAC, fp32 accumulation (VS_MATH_BF16, BASELINE.json configs[4]), on fp32 tensors (the hidden-512 transformer convs: 1 x 1 projections,
// FFN k = 9) and on bf16-RESIDENT tensors (the generator's wide stages).  Bit-identical to conv_split_kernel<1, 8, 4, 1, 1> / conv_split_kernel_bf16io<1, 8, 4, 1, 1, IO>.
// One MFMA per (tap, column tile): a chunk body has 8 * KT gaps for ~100 staging instructions -- at k = 1 the staging, not the matrix pipe, sets the pace.
#include "conv_ktap.inc"

namespace vs {

template <int WM, int WN, int NT>
static int launch_ktap_bf16_small(const ConvParams &p, hipStream_t s) {
    const bool m = (p.in_act == VS_IN_MASK);
    if (p.KT == 1) return m ? launch_ktap_inst<1, VS_IN_MASK, 1, 0, WM, WN, NT>(p, s) : launch_ktap_inst<1, VS_IN_NONE, 1, 0, WM, WN, NT>(p, s);
    return m ? launch_ktap_inst<9, VS_IN_MASK, 1, 0, WM, WN, NT>(p, s) : launch_ktap_inst<9, VS_IN_NONE, 1, 0, WM, WN, NT>(p, s);
}

int launch_ktap_bf16(const ConvParams &p, int cfg, hipStream_t s) {
    const int io = (p.x_bf16 ? 1 : 0) | (p.y_bf16 ? 2 : 0);
    if (!ktap_geometry_ok(p) || !ktap_instance(1, cfg, p.KT, io, p.in_act)) {
        set_error("launch_ktap_bf16: not a plain stride-1 conv of whole 16-channel chunks with an instance (taps %d, tensors %d, transform %d)", p.KT, io, p.in_act);
        return VS_EUNSUPPORTED;
    }
    if (cfg == 6) return launch_ktap_bf16_small<1, 4, 1>(p, s);
    if (cfg != 0) return launch_ktap_bf16_small<2, 2, 4>(p, s);
    const bool m = (p.in_act == VS_IN_MASK || p.in_act == VS_IN_LRELU_MASK);
#define KTAP_GO(KT, A0, A1, IO) return m ? launch_ktap_inst<KT, A1, 1, IO>(p, s) : launch_ktap_inst<KT, A0, 1, IO>(p, s)
    const bool plain = (p.in_act == VS_IN_NONE || p.in_act == VS_IN_MASK);
    if (io == 3) {
        switch (p.KT) {
            case 3: KTAP_GO(3, VS_IN_LRELU, VS_IN_LRELU_MASK, 3);
            case 7: KTAP_GO(7, VS_IN_LRELU, VS_IN_LRELU_MASK, 3);
            default: KTAP_GO(11, VS_IN_LRELU, VS_IN_LRELU_MASK, 3);
        }
    }
    switch (p.KT) {
        case 1: KTAP_GO(1, VS_IN_NONE, VS_IN_MASK, 0);
        case 9: KTAP_GO(9, VS_IN_NONE, VS_IN_MASK, 0);
        case 3: KTAP_GO(3, VS_IN_LRELU, VS_IN_LRELU_MASK, 0);
        case 11: KTAP_GO(11, VS_IN_LRELU, VS_IN_LRELU_MASK, 0);
        default:
            if (plain) KTAP_GO(7, VS_IN_NONE, VS_IN_MASK, 0);
            KTAP_GO(7, VS_IN_LRELU, VS_IN_LRELU_MASK, 0);
    }
#undef KTAP_GO
}

}  // namespace vs
