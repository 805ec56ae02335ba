"""Drop-in for the reference's ``styleganv1.py``: the same public names (so
``from styleganv1 import StyleGenerator, StyleDiscriminator`` in ``model.py`` / ``train.py`` keeps
working), every forward on the MI355X HIP kernels.  Implementation: ``speak-hack_amd/``.

Not rebuilt: the legacy ``G_synthesis`` / ``G_mapping`` / ``GBlock`` / ``LayerEpilogue`` graph
(styleganv1.py:155-446) -- nothing in the reference instantiates it and its constructor needs a CUDA
device; its leaf ops (``Blur2d``, ``Upscale2d``, ``PixelNorm``, ``InstanceNorm``) are here.
"""
import importlib as _importlib

_pkg = _importlib.import_module("speak-hack_amd")
_dec = _importlib.import_module("speak-hack_amd.decoder")
_leg = _importlib.import_module("speak-hack_amd.legacy")
_dis = _importlib.import_module("speak-hack_amd.discriminator")

FC = _dec.FC
ApplyNoise = _dec.ApplyNoise
ApplyStyle = _dec.ApplyStyle
SynthesisBlock = _dec.SynthesisBlock
SynthesisNetwork = _dec.SynthesisNetwork
StyleGenerator = _dec.StyleGenerator
StyleDiscriminator = _dis.StyleDiscriminator
DiscriminatorBlock = _dis.DiscriminatorBlock
Blur2d = _leg.Blur2d
Upscale2d = _leg.Upscale2d
PixelNorm = _leg.PixelNorm
InstanceNorm = _leg.InstanceNorm

__all__ = ["FC", "ApplyNoise", "ApplyStyle", "SynthesisBlock", "SynthesisNetwork", "StyleGenerator",
           "StyleDiscriminator", "DiscriminatorBlock", "Blur2d", "Upscale2d", "PixelNorm", "InstanceNorm"]
