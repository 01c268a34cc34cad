"""Is the pass run-to-run deterministic at the BENCH sizes?  (round 5: at config 4 -- 720x1280x320, sub-videos of 80 -- two of four streaming
graph instances differed from the whole-pass graph by +-1 byte in ~0.2 % of the bytes of one sub-video's frames, differently on different runs.)
Runs the eager pass N times with the stage tensors and reports the first stage whose bytes differ between runs, then replays the whole-pass
hipGraph N times.    python tools/diag_determinism.py [frames=80] [subvideo=80] [runs=3] [height=720] [width=1280]"""
import os
import sys

import numpy as np
import scipy.ndimage
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd.pipeline import ClipGraph, InferenceConfig, run_clip                # noqa: E402
from propainter_amd.synthetic import seeded_models, synthetic_clip, synthetic_mask    # noqa: E402

L, S, N, H, W = (int(v) for v in (sys.argv[1:6] + ["80", "80", "3", "720", "1280"][len(sys.argv) - 1:]))
dev = torch.device("cuda")
models = seeded_models(dev, raft_precision="f16x3")
cfg = InferenceConfig(subvideo_length=S, fp16=True)
clip = torch.from_numpy(synthetic_clip(L, H, W)).to(dev)
m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.uint8) * 255
masks = torch.from_numpy(np.repeat(m[None], L, 0)).to(dev)


def stages():
    comp, st = run_clip(models, clip, masks, masks, cfg, dev, return_stages=True)
    torch.cuda.synchronize()
    return {"raft_f": st["gt_flows"][0].clone(), "raft_b": st["gt_flows"][1].clone(), "completed_f": st["pred_flows"][0].clone(),
            "completed_b": st["pred_flows"][1].clone(), "updated_frames": st["updated_frames"].clone(), "updated_masks": st["updated_masks"].clone(),
            "composited": comp.clone()}


ref = stages()
for i in range(1, N):
    cur = stages()
    diffs = {k: float((cur[k].float() - ref[k].float()).abs().max()) for k in ref if not torch.equal(cur[k], ref[k])}
    if diffs:
        k0 = next(k for k in ref if k in diffs)
        ne = cur[k0] != ref[k0]
        where = [j for j in range(ne.shape[1 if ne.dim() == 5 else 0]) if bool((ne[:, j] if ne.dim() == 5 else ne[j]).any())][:8]
        print(f"DETERMINISM eager run {i} vs run 0 ({H}x{W}x{L}, sub-videos of {S}): DIFFERENT; first stage {k0}, frames {where}, max |d| per stage {diffs}", flush=True)
    else:
        print(f"DETERMINISM eager run {i} vs run 0 ({H}x{W}x{L}, sub-videos of {S}): identical in every stage", flush=True)
del cur
g = ClipGraph(models, L, H, W, cfg, dev, example=(clip, masks, masks), release_eager_pool=True)
outs = []
for i in range(N + 1):
    outs.append(g.replay().clone())
    torch.cuda.synchronize()
for i in range(1, N + 1):
    ne = outs[i] != outs[0]
    print(f"DETERMINISM graph replay {i} vs replay 0: {'identical' if not bool(ne.any()) else 'DIFFERENT in frames ' + str([j for j in range(L) if bool(ne[j].any())][:8])}"
          f"; vs eager: {'identical' if torch.equal(outs[i], ref['composited']) else 'DIFFERENT'}", flush=True)
