"""Capture-time memory-hazard check (propainter_amd/hazard.py) of the multi-stream submission forms, on the GPU:

    python tools/check_hazards.py clip   [frames=80]  [subvideo=80] [height=720] [width=1280]     whole-pass ClipGraph (lanes inside)
    python tools/check_hazards.py stream [frames=320] [subvideo=80] [height] [width] [stages=0,2]  StreamingClipGraph, ONE stage-pipelined graph
    python tools/check_hazards.py multi  [frames=320] [subvideo=80] [height] [width]               per-(rank, segment) graphs, CONCURRENT launches
    python tools/check_hazards.py eager  [frames]     [subvideo]                                   the eager pass alone (window / RAFT lanes)

Prints one JSON line `HAZARDS {...}`: launches seen, allocator generations, and every pair of accesses to overlapping memory (one a
write) that no stream / event edge orders -- "alias" (a recycled block touched without an edge from its previous life), "race"
(inside one allocation), "race?" (two strided channel windows of one NHWC buffer: overlap as byte ranges only), "uninit" (a read of memory that nothing has written
since the allocator handed the block out: the result then depends on what the previous tenant left).  With PP_HAZARD_STACKS=1
the findings carry the Python call sites.  The pass's bytes are compared with an eager pass WITHOUT the recorder as well."""
import json
import os
import sys
import time

import numpy as np
import scipy.ndimage
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd import hazard                                                        # noqa: E402
from propainter_amd.pipeline import ClipGraph, InferenceConfig, run_clip                 # noqa: E402
from propainter_amd.synthetic import seeded_models, synthetic_clip, synthetic_mask     # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "clip"
defaults = {"clip": ["80", "80"], "eager": ["80", "80"], "stream": ["320", "80"], "multi": ["320", "80"]}[mode] + ["720", "1280"]
argv = sys.argv[2:6]
L, S, H, W = (int(v) for v in (argv + defaults[len(argv):]))
stages = sys.argv[6] if len(sys.argv) > 6 else "0,2"
dev = torch.device("cuda")
models = seeded_models(dev, raft_precision="f16x3")
cfg = InferenceConfig(subvideo_length=S, fp16=True)
clip = torch.from_numpy(synthetic_clip(L, H, W)).to(dev)
m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.uint8) * 255
masks = torch.from_numpy(np.repeat(m[None], L, 0)).to(dev)

ref = run_clip(models, clip, masks, masks, cfg, dev).clone()          # engines, tables; the bytes every form must reproduce
torch.cuda.synchronize()
t0 = time.time()
same = []
with hazard.Recorder(dev, stacks=os.environ.get("PP_HAZARD_STACKS") == "1") as rec:
    if mode == "eager":
        for _ in range(2):
            same.append(bool(torch.equal(run_clip(models, clip, masks, masks, cfg, dev), ref)))
    elif mode == "clip":
        g = ClipGraph(models, L, H, W, cfg, dev, example=(clip, masks, masks), forked_branches=True)      # the multi-stream form is what is checked
        for _ in range(2):
            same.append(bool(torch.equal(g.replay(), ref)))
    else:
        from propainter_amd.sharding import StreamingClipGraph
        if mode == "stream":
            os.environ["PP_SG_STAGES"] = stages
            s = StreamingClipGraph(models, L, H, W, cfg, dev, single_graph=True, validate=0)
        else:
            s = StreamingClipGraph(models, L, H, W, cfg, dev, single_graph=False, share_pool=False)
        s.load(clip, masks, masks)
        s.capture()
        outs, replay_ms = [], []
        for _ in range(int(os.environ.get("PP_HZ_REPLAYS", "3"))):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                torch.cuda.synchronize(); tr0 = time.time()
                out = s.replay(concurrent=True) if mode == "multi" else s.replay()
                torch.cuda.synchronize(); replay_ms.append(round((time.time() - tr0) * 1e3, 1))
            same.append(bool(torch.equal(out, ref)))
            outs.append(out.clone() if len(outs) < 6 else out)
        # a deviation that is the SAME in every replay is a different (deterministic) computation; one that changes is a race
        detail = {"replay_ms_under_the_recorder": replay_ms, "replay_i_equals_replay_0": [bool(torch.equal(o, outs[0])) for o in outs],
                  "frames_differing_from_eager": [[int(j) for j in range(L) if bool((o[j] != ref[j]).any())][:12] for o in outs[:6]],
                  "bytes_differing_from_eager": [int((o != ref).sum()) for o in outs],
                  "max_abs_vs_eager": [int((o.to(torch.int16) - ref.to(torch.int16)).abs().max()) for o in outs]}
    torch.cuda.synchronize()
rep = rec.report()
rep["replay_detail"] = detail if mode in ("stream", "multi") else None
rep.update(mode=mode, frames=L, subvideo=S, height=H, width=W, stages=stages if mode == "stream" else None,
           replays_equal_eager=same, seconds=round(time.time() - t0, 1), stream_names=len(rec.names))
for k in ("alias", "race", "race?", "uninit", "intra"):
    rep[k + "_count"] = len(rep[k])
from propainter_amd import conv as _pconv                                                # noqa: E402
rep["conv_inplace"] = list(_pconv.inplace_findings)      # conv.check_inplace: exact channel-window test of every convolution launch
print("HAZARDS " + json.dumps(rep), flush=True)
