// 3x3 stride-1 instantiations, half-depth chunk configs (CI_T = 4): ids 4-7.
#include "conv_mfma_f32.hpp"

namespace spkconv {

// (every mode also exists with the fixed 32-wide tile geometry -- conv_kernel<..., FG = true> -- for the 64-, 128- and 256-pixel
// tiles: the layers from 32^2 up)
template <class C>
static int by_mode(int mode, const spk_conv2d_desc* d, hipStream_t s) {
    constexpr bool FG = C::PIX_T >= 64;
    switch (mode) {
        case MODE_PLAIN: return run<C, 3, 3, 1, MODE_PLAIN, FG>(d, s);
        case MODE_UPSAMPLE: return run<C, 3, 3, 1, MODE_UPSAMPLE, FG>(d, s);
        case MODE_BATCH_SCALE: return run<C, 3, 3, 1, MODE_BATCH_SCALE, FG>(d, s);
        case MODE_UPSAMPLE_BATCH_SCALE: return run<C, 3, 3, 1, MODE_UPSAMPLE_BATCH_SCALE, FG>(d, s);
        default: return run<C, 3, 3, 1, MODE_AFFINE_RELU, FG>(d, s);
    }
}

int run_3x3s1_b(int cfg, int mode, const spk_conv2d_desc* d, hipStream_t s) {
    switch (cfg) {
        case 4: return by_mode<Cfg4>(mode, d, s);
        case 5: return by_mode<Cfg5>(mode, d, s);
        case 6: return by_mode<Cfg6>(mode, d, s);
        default: return by_mode<Cfg7>(mode, d, s);
    }
}

}  // namespace spkconv
