// 3x3 stride-1 instantiations with the BatchNorm-statistics epilogue on a plain input (MODE_PLAIN_STATS), ids 0-7: kept apart
// from the decoders' hot instantiations, whose code generation the statistics epilogue measurably disturbs
// (conv_mfma_f32.hpp, HAS_STATS).
#include "conv_mfma_f32.hpp"

namespace spkconv {

int run_3x3s1_stats(int cfg, const spk_conv2d_desc* d, hipStream_t s) {
    switch (cfg) {
        case 0: return run<Cfg0, 3, 3, 1, MODE_PLAIN_STATS>(d, s);
        case 1: return run<Cfg1, 3, 3, 1, MODE_PLAIN_STATS>(d, s);
        case 2: return run<Cfg2, 3, 3, 1, MODE_PLAIN_STATS>(d, s);
        case 3: return run<Cfg3, 3, 3, 1, MODE_PLAIN_STATS>(d, s);
        case 4: return run<Cfg4, 3, 3, 1, MODE_PLAIN_STATS>(d, s);
        case 5: return run<Cfg5, 3, 3, 1, MODE_PLAIN_STATS>(d, s);
        case 6: return run<Cfg6, 3, 3, 1, MODE_PLAIN_STATS>(d, s);
        default: return run<Cfg7, 3, 3, 1, MODE_PLAIN_STATS>(d, s);
    }
}

}  // namespace spkconv
