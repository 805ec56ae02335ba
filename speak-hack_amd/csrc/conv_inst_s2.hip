// stride-2 instantiations: 3x3 s2 (bottleneck conv2 of the first block of layer2-4) and the 7x7 s2 stem.
#include "conv_mfma_f32.hpp"

namespace spkconv {

template <class C, int K>
static int by_mode(int mode, const spk_conv2d_desc* d, hipStream_t s) {
    if constexpr (K == 7) {   // the stem reads the raw image: no producer BatchNorm to fold in
        if (mode != MODE_PLAIN) return spk::fail(SPK_EUNSUPPORTED, "conv2d: 7x7 is built for plain input only");
        return run<C, K, K, 2, MODE_PLAIN>(d, s);
    } else {
        return mode == MODE_AFFINE_RELU ? run<C, K, K, 2, MODE_AFFINE_RELU>(d, s) : run<C, K, K, 2, MODE_PLAIN>(d, s);
    }
}

template <int K>
static int by_cfg(int cfg, int mode, const spk_conv2d_desc* d, hipStream_t s) {
    switch (cfg) {
        case 4: return by_mode<Cfg4, K>(mode, d, s);
        case 5: return by_mode<Cfg5, K>(mode, d, s);
        case 6: return by_mode<Cfg6, K>(mode, d, s);
        default: return by_mode<Cfg7, K>(mode, d, s);
    }
}

int run_3x3s2_7x7s2(int kh, int cfg, int mode, const spk_conv2d_desc* d, hipStream_t s) {
    return kh == 7 ? by_cfg<7>(cfg, mode, d, s) : by_cfg<3>(cfg, mode, d, s);
}

}  // namespace spkconv
