// stride-2 instantiations: 3x3 s2 (bottleneck conv2 of the first block of layer2-4), the 7x7 s2 stem, and 4x4 s2 pad 1 --
// the data gradient of nn.ConvTranspose2d(Cin, Cout, 4, stride=2, padding=1), the fused upscale of the legacy GBlock
// (styleganv1.py:231): dx = conv2d(dy, weight, stride 2, pad 1) with the module's [Cin,Cout,4,4] parameter read as a conv
// weight [out = Cin][in = Cout].
#include "conv_mfma_f32.hpp"

namespace spkconv {

template <class C, int K>
static int by_mode(int mode, const spk_conv2d_desc* d, hipStream_t s) {
    if constexpr (K == 7 || K == 4) {   // the stem reads the raw image, the 4x4 a gradient: no producer BatchNorm to fold in
        if (mode != MODE_PLAIN) return spk::fail(SPK_EUNSUPPORTED, "conv2d: 7x7 / 4x4 are built for plain input only");
        return run<C, K, K, 2, MODE_PLAIN>(d, s);
    } else {      // 3x3 stride 2: also in the fixed-geometry build (32-wide output tiles; conv_mfma_f32.hpp, FG)
        constexpr bool FG = C::PIX_T >= 64;
        return mode == MODE_AFFINE_RELU ? run<C, K, K, 2, MODE_AFFINE_RELU, FG>(d, s) : run<C, K, K, 2, MODE_PLAIN, FG>(d, s);
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
    return kh == 7 ? by_cfg<7>(cfg, mode, d, s) : (kh == 4 ? by_cfg<4>(cfg, mode, d, s) : by_cfg<3>(cfg, mode, d, s));
}

}  // namespace spkconv
