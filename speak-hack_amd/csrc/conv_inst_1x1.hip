// 1x1 instantiations (stride 1 and 2): deep chunks (CI_T = 32 / 16) since a chunk has only CI_T/2 k-steps.
#include "conv_mfma_f32.hpp"

namespace spkconv {

template <class C, int S>
static int by_mode(int mode, const spk_conv2d_desc* d, hipStream_t s) {
    return mode == MODE_AFFINE_RELU ? run<C, 1, 1, S, MODE_AFFINE_RELU>(d, s) : run<C, 1, 1, S, MODE_PLAIN>(d, s);
}

template <int S>
static int by_cfg(int cfg, int mode, const spk_conv2d_desc* d, hipStream_t s) {
    switch (cfg) {
        case 8: return by_mode<Cfg8, S>(mode, d, s);
        case 9: return by_mode<Cfg9, S>(mode, d, s);
        case 10: return by_mode<Cfg10, S>(mode, d, s);
        default: return by_mode<Cfg11, S>(mode, d, s);
    }
}

int run_1x1(int stride, int cfg, int mode, const spk_conv2d_desc* d, hipStream_t s) {
    return stride == 2 ? by_cfg<2>(cfg, mode, d, s) : by_cfg<1>(cfg, mode, d, s);
}

}  // namespace spkconv
