// 3x3 stride-1 instantiations, full-depth chunk configs (CI_T = 8): ids 0-3.
#include "conv_mfma_f32.hpp"

namespace spkconv {

template <class C>
static int by_mode(int mode, const spk_conv2d_desc* d, hipStream_t s) {
    switch (mode) {
        case MODE_PLAIN: return run<C, 3, 3, 1, MODE_PLAIN>(d, s);
        case MODE_UPSAMPLE: return run<C, 3, 3, 1, MODE_UPSAMPLE>(d, s);
        default: return run<C, 3, 3, 1, MODE_AFFINE_RELU>(d, s);
    }
}

int run_3x3s1_a(int cfg, int mode, const spk_conv2d_desc* d, hipStream_t s) {
    switch (cfg) {
        case 0: return by_mode<Cfg0>(mode, d, s);
        case 1: return by_mode<Cfg1>(mode, d, s);
        case 2: return by_mode<Cfg2>(mode, d, s);
        default: return by_mode<Cfg3>(mode, d, s);
    }
}

}  // namespace spkconv
