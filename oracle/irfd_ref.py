"""TEST INFRASTRUCTURE ONLY -- CPU restatement of ``IRFD.forward`` (model.py:80-126) over a flat
state dict with the reference's prefixes (``Ei.`` ``Ee.`` ``Ep.`` ``Gd.`` ``Cm.``).

``model.py`` itself cannot be imported here (seven absent third-party packages and a network
weight fetch), so this follows its text.  The decoder half is pinned through
``decoder_ref`` (goldens from the reference's own code); the encoder half through ``resnet_ref``
(independent implementation) -- the composition is "parity unpinned" by the reference.
The host-RNG draw ``torch.randint(0, 3, (1,))`` (model.py:98) is an explicit argument.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import decoder_ref as D
from . import resnet_ref as E


def sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def irfd_forward(x_s, x_t, sd, swap_type, noises_s, noises_t, training=False, update_running_stats=False,
                 mix_s=None, mix_t=None):
    """Returns the reference's 10-tuple.  ``mix_*`` = (mix_features, mix_layer) for the train-mode
    style-mixing branch of each decoder call (None = branch not taken / eval)."""
    enc = {n: sub(sd, n + ".") for n in ("Ei", "Ee", "Ep")}
    run = lambda n, x: E.resnet50_trunk(x, enc[n], training=training, update_running_stats=update_running_stats)
    # model.py:84-90 -- call order matters for the running statistics: Ei, Ee, Ep on x_s, then on x_t
    fi_s, fe_s, fp_s = run("Ei", x_s), run("Ee", x_s), run("Ep", x_s)
    fi_t, fe_t, fp_t = run("Ei", x_t), run("Ee", x_t), run("Ep", x_t)
    if swap_type == 0:                      # model.py:98-104
        fi_s, fi_t = fi_t, fi_s
    elif swap_type == 1:
        fe_s, fe_t = fe_t, fe_s
    else:
        fp_s, fp_t = fp_t, fp_s
    cat = lambda *f: torch.cat([t.view(t.size(0), -1) for t in f], dim=1)          # model.py:64-69
    gd = sub(sd, "Gd.")
    kw_s = dict(mix_features=mix_s[0], mix_layer=mix_s[1]) if mix_s else {}
    kw_t = dict(mix_features=mix_t[0], mix_layer=mix_t[1]) if mix_t else {}
    x_s_recon = D.style_generator(cat(fi_s, fe_s, fp_s), gd, noises_s, **kw_s)     # model.py:113-114
    x_t_recon = D.style_generator(cat(fi_t, fe_t, fp_t), gd, noises_t, **kw_t)
    emo = lambda fe: torch.softmax(F.linear(fe.view(fe.size(0), -1), sd["Cm.weight"], sd["Cm.bias"]), dim=1)
    return x_s_recon, x_t_recon, fi_s, fe_s, fp_s, fi_t, fe_t, fp_t, emo(fe_s), emo(fe_t)


def irfd_recipe_state_dict():
    """Recipe weights for the parts ``IRFD.forward`` touches (encoders, Gd, Cm)."""
    from .weights_recipe import recipe_tensor, resnet_trunk_state_dict
    import re
    sd = {}
    for n in ("Ei", "Ee", "Ep"):
        sd.update({f"{n}.{k}": v for k, v in resnet_trunk_state_dict(n + ".").items()})
    sd["Cm.weight"] = recipe_tensor("Cm.weight", (8, 2048))
    sd["Cm.bias"] = recipe_tensor("Cm.bias", (8,))
    return sd
