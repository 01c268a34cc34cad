"""Hole-only goldens of SYNTHETIC clips through the fp32 CPU oracle (test infrastructure; run in the authoring container only):

    python -m oracle.make_golden_synth c2        # BASELINE config 2: 432x240, 80 frames, tame recipe        (~7 CPU-min on 8 cores)
    python -m oracle.make_golden_synth c3w       # 720x1280, 18 frames: windows WITH reference frames, tame  (~12 CPU-min)
    python -m oracle.make_golden_synth c3        # BASELINE config 3: 720x1280, 80 frames, 16 windows, reference frames +-40: the TIMED clip (~1 CPU-hour)
    python -m oracle.make_golden_synth stress    # 720x1280, 6 frames, STRESS recipe: full-size flow / offset heads, two layers moving in
                                                 # opposite directions at 8-48 px/frame + occluder, border + lattice mask (all windows masked)

The clips, masks and weights are pure functions of seeds (propainter_amd/synthetic.py), so a fixture stores only the oracle's bytes
INSIDE the dilated mask (outside it the composite is the input frame -- asserted here), the driver settings and SHA-256 digests of the
inputs it was computed from; tests/ and bench.py regenerate the inputs, check the digests and compare the HIP path's bytes."""
import hashlib
import os
import sys
import time
import warnings

import numpy as np
import torch

from . import propainter_oracle as O
from propainter_amd.synthetic import seeded_models

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CASES = {
    # name: (file, L, H, W, recipe)
    "c2": ("synth_c2_432x240x80.npz", 80, 240, 432, "tame"),
    "c3w": ("synth_c3_720x1280x18.npz", 18, 720, 1280, "tame"),
    "c3": ("synth_c3_720x1280x80.npz", 80, 720, 1280, "tame"),      # BASELINE config 3 itself: the clip bench.py times on rank 0
    "stress": ("synth_stress_720x1280x6.npz", 6, 720, 1280, "stress"),
    "stress_small": ("synth_stress_240x432x12.npz", 12, 240, 432, "stress"),
}


def inputs(L, H, W, recipe):
    """(frames uint8 [L,H,W,3], dilated masks uint8 [L,H,W] {0,255}) of a case: propainter_amd.synthetic.case_inputs."""
    from propainter_amd.synthetic import case_inputs
    return case_inputs(L, H, W, recipe)


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main(name):
    fn, L, H, W, recipe = CASES[name]
    warnings.filterwarnings("ignore")
    torch.set_num_threads(os.cpu_count())
    clip, masks = inputs(L, H, W, recipe)
    raft, fc, gen = seeded_models("cpu", recipe=recipe)
    sds = {"raft": {k: v.float() for k, v in raft.fix_raft.state_dict().items()},
           "fc": {k: v.float() for k, v in fc.state_dict().items()},
           "gen": {k: v.float() for k, v in gen.state_dict().items()}}
    kw = dict(raft_iter=20, subvideo_length=80, neighbor_length=10, ref_stride=10)
    t0, timers = time.time(), {}
    with torch.no_grad():
        comp = np.stack(O.inpaint_video(sds, clip, masks, masks, timers=timers, **kw))
    hole = masks > 0
    assert np.array_equal(comp[~hole], clip[~hole])
    wd = hashlib.sha256()
    for part in ("raft", "fc", "gen"):
        for k in sorted(sds[part]):
            wd.update(sds[part][k].numpy().tobytes())
    out = os.path.join(GOLDEN, fn)
    np.savez_compressed(out, comp_hole=comp[hole], frames_sha256=digest(clip), masks_sha256=digest(masks), weights_sha256=wd.hexdigest(),
                        L=L, H=H, W=W, recipe=recipe, oracle_seconds=time.time() - t0, **kw)
    print(out, os.path.getsize(out), f"{time.time() - t0:.0f} s", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in timers.items()},
          "hole fraction", float(hole.mean()), flush=True)


if __name__ == "__main__":
    for n in sys.argv[1:] or ["c2"]:
        main(n)
