// 2x2 stride-1 instantiations (window rows y, y+1 / columns x, x+1, zeros beyond the input): the four output-parity
// classes of the data gradient of a 3x3 stride-2 pad-1 conv (the bottleneck conv2 of the first block of layer2-4 of the
// trunk; every block's conv2 of StyleDiscriminator, styleganv1.py:644-657), written interleaved into the full-size
// gradient by the kernel's epilogue.  dx[2m+py, 2n+px] needs 1 / 2 / 2 / 4 of the 9 taps; the zero-dilated form ran all
// 9 at four times the pixels.  The same four-class launch, anchored one pixel up / left (pshift = 1), is the forward of
// nn.ConvTranspose2d(Cin, Cout, 4, stride=2, padding=1), the fused upscale of the legacy GBlock (styleganv1.py:231): every
// output pixel (2m+py, 2n+px) takes exactly 2x2 of the 16 taps.
#include "conv_mfma_f32.hpp"

namespace spkconv {

int run_2x2_parity(int cfg, const spk_conv2d_desc* d, int Hd, int Wd, int pshift, hipStream_t s) {
    switch (cfg) {
        case 0: return run<Cfg0, 2, 2, 1, MODE_PLAIN>(d, s, Hd, Wd, pshift);
        case 1: return run<Cfg1, 2, 2, 1, MODE_PLAIN>(d, s, Hd, Wd, pshift);
        case 2: return run<Cfg2, 2, 2, 1, MODE_PLAIN>(d, s, Hd, Wd, pshift);
        default: return run<Cfg3, 2, 2, 1, MODE_PLAIN>(d, s, Hd, Wd, pshift);
    }
}

}  // namespace spkconv
