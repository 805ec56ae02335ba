"""Pins oracle/progan_ref.py (the stylegan.py generator) to the golden vectors the
reference's own stylegan.py produced (tests/golden/progan.npz).  CPU only."""
import torch

from conftest import rel_l2
from oracle import progan_ref as P
from oracle.weights_recipe import recipe_input

torch.set_num_threads(8)
CASES = [(0, 1.0, True, 2), (3, 0.3, True, 2), (3, 1.0, False, 2), (6, 0.3, False, 1)]


def case_inputs(steps, alpha, zero_noise, B):
    tag = f"s{steps}_a{alpha}_z{int(zero_noise)}"
    w = recipe_input(f"progan.{tag}.w", (B, 512))
    noises = None if zero_noise else [recipe_input(f"progan.{tag}.n{i}", s) for i, s in enumerate(P.noise_shapes(B, steps))]
    return tag, w, noises


def test_generator_matches_reference_goldens(golden):
    g = golden("progan.npz")
    sd = P.generator_recipe_state_dict()
    assert len(sd) == 145 and sum(v.numel() for k, v in sd.items() if not k.startswith("rgb_layers.0")) == 24117867
    for steps, alpha, zero_noise, B in CASES:
        tag, w, noises = case_inputs(steps, alpha, zero_noise, B)
        with torch.no_grad():
            y = P.generator(w, alpha, steps, sd, noises)
        assert y.shape == (B, 3, 4 * 2 ** steps, 4 * 2 ** steps)
        ref = g[f"{tag}.y"]
        got = y if y.shape[-1] <= 64 else y[..., ::4, ::4]
        assert rel_l2(got, ref) < 5e-6, tag
        assert abs(float(y.double().norm()) / float(g[f"{tag}.norm"]) - 1) < 1e-5
