// conv_ktap_pair.hip -- the VS_CONV1D_PAIRED instances of conv_ktap_kernel (conv_ktap.inc; design notes in conv_ktap.hip): the WaveNet's dilated k = 5 in_layers with the
// gate tanh(a) * sigmoid(b) / the coupling update in the epilogue (reference modules/visinger/encoder.py:158-161, 167-195, 206-213; modules/visinger/flow.py:66-85).  A wave owns
// the two 32-row tiles of a row pair (rows c and H + c) over 64 columns: 2 x 2 waves = 128 virtual rows x 128 columns, the tile of conv_split_kernel<2, 2, 2, 2, T>, to which the
// outputs are bit-identical.  Split-f16 x3 (the headline) and plain bf16 on fp32 tensors (BASELINE configs[4]).
#include "conv_ktap.inc"

namespace vs {

bool ktap_pair_instance(int terms, int kt, int io, int in_act, int mt) {
    return (terms == 3 || terms == 1) && kt == 5 && io == 0 && in_act == VS_IN_NONE && mt >= 4;
}

// p as for launch_split(cfg 4) on a VS_CONV1D_PAIRED conv with ktap_pair_instance(...)
int launch_ktap_pair(const ConvParams &p, int terms, hipStream_t s) {
    if (!ktap_geometry_ok(p, VS_CONV1D_PAIRED) || p.x_bf16 || p.y_bf16 || p.KT != 5 || p.in_act != VS_IN_NONE) {
        set_error("launch_ktap_pair: not a paired 5-tap conv of whole 16-channel chunks on fp32 tensors");
        return VS_EUNSUPPORTED;
    }
    if (terms == 3) return launch_ktap_inst<5, VS_IN_NONE, 2, 0, 2, 2, 2, 2>(p, s);
    return launch_ktap_inst<5, VS_IN_NONE, 1, 0, 2, 2, 2, 2>(p, s);
}

}  // namespace vs
