"""Drop-in for the reference's ``stylegan.py`` (imported by nothing in the reference, kept for signature
compatibility -- SURVEY.md 8a A12): the same public names, forward on the MI355X HIP kernels.
Implementation: ``speak-hack_amd/progan.py``."""
import importlib as _importlib

_m = _importlib.import_module("speak-hack_amd.progan")

WSLinear = _m.WSLinear
PixelNorm = _m.PixelNorm
WSConv2d = _m.WSConv2d
MappingNetwork = _m.MappingNetwork
InjectNoise = _m.InjectNoise
AdaIN = _m.AdaIN
GenBlock = _m.GenBlock
ConvBlock = _m.ConvBlock
Generator = _m.Generator
Discriminator = _m.Discriminator
factors = _m.factors

__all__ = ["WSLinear", "PixelNorm", "WSConv2d", "MappingNetwork", "InjectNoise", "AdaIN", "GenBlock", "ConvBlock",
           "Generator", "Discriminator", "factors"]
