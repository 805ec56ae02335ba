"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the SPEAK generative hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and there only as the checker / the reported
CPU baseline -- never as the thing measured or shipped.  The product path
(``speak-hack_amd``) never imports this package and raises when its HIP library is missing.

Parity pinning status (see DESIGN.md "Oracle"):

* decoder (``decoder_ref``), legacy ops (``legacy_ops_ref``), ``stylegan.py`` generator
  (``progan_ref``): PINNED -- checked against golden vectors under ``tests/golden/`` that
  ``tools/make_goldens.py`` produced in the build container by importing the reference's own
  ``styleganv1.py`` / ``stylegan.py``.
* ResNet-50 trunk (``resnet_ref``): the arithmetic lives in third-party ``torchvision`` (not
  vendored, no version pinned).  Cross-checked against an independent
  implementation of the same published architecture (``transformers`` ResNetModel, random
  init); the reference itself holds no fixture for it => "parity unpinned" by the reference.
* StyleGAN2 modulated-conv variant (``modconv_ref``): not present in the reference at all
   => "parity unpinned"; restates the published formula.
"""
