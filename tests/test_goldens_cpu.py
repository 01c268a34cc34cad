"""The committed synthetic goldens (tests/golden/synth_*.npz; oracle/make_golden_synth.py) store the fp32 CPU oracle's bytes inside the
dilated mask plus SHA-256 digests of what they were computed FROM.  Inputs and weights are pure functions of seeds
(propainter_amd/synthetic.py): this test regenerates them and checks every digest, so that a change of a recipe cannot silently
detach the fixtures from the tests / bench legs that use them (GPU: tests/test_stress_gpu.py, bench.py `parity_windows_with_reference_frames`,
`configs`, `stress`)."""
import hashlib
import os

import numpy as np
import pytest

from propainter_amd.synthetic import case_inputs, seeded_models

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ["synth_c2_432x240x80.npz", "synth_c3_720x1280x18.npz", "synth_c3_720x1280x80.npz", "synth_stress_240x432x12.npz", "synth_stress_720x1280x6.npz"]
_weights = {}


def _weights_digest(recipe):
    if recipe not in _weights:
        raft, fc, gen = seeded_models("cpu", recipe=recipe)
        h = hashlib.sha256()
        for sd in (raft.fix_raft.state_dict(), fc.state_dict(), gen.state_dict()):
            for k in sorted(sd):
                h.update(sd[k].float().numpy().tobytes())
        _weights[recipe] = h.hexdigest()
    return _weights[recipe]


@pytest.mark.parametrize("fn", FIXTURES)
def test_synthetic_golden_matches_the_regenerated_inputs_and_weights(fn):
    g = np.load(os.path.join(GOLDEN, fn), allow_pickle=False)
    L, H, W, recipe = int(g["L"]), int(g["H"]), int(g["W"]), str(g["recipe"])
    clip, masks = case_inputs(L, H, W, recipe)
    dg = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert clip.shape == (L, H, W, 3) and masks.shape == (L, H, W) and set(np.unique(masks)) <= {0, 255}
    assert dg(clip) == str(g["frames_sha256"]), "the clip recipe changed: regenerate the golden (oracle/make_golden_synth.py)"
    assert dg(masks) == str(g["masks_sha256"])
    assert _weights_digest(recipe) == str(g["weights_sha256"]), "the weight recipe changed: regenerate the golden"
    assert g["comp_hole"].dtype == np.uint8 and g["comp_hole"].shape == (int((masks > 0).sum()), 3)
    assert (int(g["raft_iter"]), int(g["subvideo_length"]), int(g["neighbor_length"]), int(g["ref_stride"])) == (20, 80, 10, 10)
    if recipe == "stress":      # the stress mask's whole point: every attention window holds a masked token, ~1/3 of the area
        assert 0.3 < float((masks[0] > 0).mean()) < 0.5
