"""Drop-in for the reference's ``styleganv1.py``: the same public names (so
``from styleganv1 import StyleGenerator, StyleDiscriminator`` in ``model.py`` / ``train.py`` keeps
working), every forward on the MI355X HIP kernels.  Implementation: ``speak-hack_amd/``.
"""
import importlib as _importlib

_pkg = _importlib.import_module("speak-hack_amd")
_dec = _importlib.import_module("speak-hack_amd.decoder")

FC = _dec.FC
ApplyNoise = _dec.ApplyNoise
ApplyStyle = _dec.ApplyStyle
SynthesisBlock = _dec.SynthesisBlock
SynthesisNetwork = _dec.SynthesisNetwork
StyleGenerator = _dec.StyleGenerator

__all__ = ["FC", "ApplyNoise", "ApplyStyle", "SynthesisBlock", "SynthesisNetwork", "StyleGenerator"]
