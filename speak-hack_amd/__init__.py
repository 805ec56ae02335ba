"""speak-hack_amd -- MI355X-native implementation of the SPEAK generative hot path
(IRFD encoders -> StyleGAN synthesis decoder), behind the reference's own module API.

The directory name carries a hyphen, so import it with
``importlib.import_module("speak-hack_amd")`` or through the top-level drop-in modules
``model`` / ``styleganv1`` / ``stylegan`` (same names as the reference's files).

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI of include/spk.h),
``_lib`` (ctypes binding), ``ops`` (launchers) and the host-side mirrors of the reference modules.
"""
from . import _lib, ops  # noqa: F401
from .decoder import (FC, ApplyNoise, ApplyStyle, StyleGenerator, SynthesisBlock,  # noqa: F401
                      SynthesisNetwork)

__all__ = ["FC", "ApplyNoise", "ApplyStyle", "SynthesisBlock", "SynthesisNetwork", "StyleGenerator", "ops"]
