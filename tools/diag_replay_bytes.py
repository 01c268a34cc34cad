"""How often does a replay of the whole-pass hipGraph (pipeline.ClipGraph) leave bytes that differ from the eager pass?  (round 6: one default bench run
ended with `parity_timed_output.max_abs` 21 -- ~500 hole bytes off by more than one in the LAST timed replay -- where ten earlier runs had 1.)
    python tools/diag_replay_bytes.py [replays=40] [window_streams=2] [raft_streams=2] [frames=80] [height=720] [width=1280] [eager] [sub=80]
(sub=20: BASELINE config 5's --subvideo_length)
One line per deviating replay (frames, bytes, max |d|) and a summary `REPLAY_BYTES {...}`.  PP_LIB_PATH selects another build of the library."""
import json
import os
import sys

import numpy as np
import scipy.ndimage
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd.pipeline import ClipGraph, InferenceConfig, run_clip                # noqa: E402
from propainter_amd.synthetic import seeded_models, synthetic_clip, synthetic_mask    # noqa: E402

EAGER = "eager" in sys.argv
STRESS = "stress" in sys.argv           # the stress recipe: full-size heads, opposite-moving layers, every attention window masked
IMPLS = dict(a.split("=") for a in sys.argv[1:] if "=" in a)      # e.g. dcn=1 off=1 bb=1: kernel family of the propagation layers (diagnosis)
argv = [a for a in sys.argv[1:] if a not in ("eager", "stress") and "=" not in a]
N, WS, RS, L, H, W = (int(v) for v in (argv[:6] + ["40", "2", "2", "80", "720", "1280"][len(argv):]))
dev = torch.device("cuda")
models = seeded_models(dev, raft_precision="f16x3", recipe="stress" if STRESS else "tame")
SUB = int(IMPLS.pop("sub", 80))
cfg = InferenceConfig(fp16=True, window_streams=WS, raft_streams=RS, subvideo_length=SUB)
if STRESS:
    from propainter_amd.synthetic import case_inputs
    c_, m_ = case_inputs(L, H, W, "stress")
    clip, masks = torch.from_numpy(c_).to(dev), torch.from_numpy(m_).to(dev)
else:
    clip = torch.from_numpy(synthetic_clip(L, H, W)).to(dev)
    m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.uint8) * 255
    masks = torch.from_numpy(np.repeat(m[None], L, 0)).to(dev)
if IMPLS:
    run_clip(models, clip, masks, masks, cfg, dev)          # builds the engines
    eng = models[2]._engine[1]
    for name in eng.prop:
        for lname, layer in eng.prop[name].items():
            key = "dcn" if lname == "dcn" else ("off" if lname.startswith("off") else "bb")
            if key in IMPLS:
                layer.impl = int(IMPLS[key])
    if "gen3x3" in IMPLS:        # every other multi-tap stride-1 convolution of the generator (encoder tail, fusion, decoder, SoftComp bias conv): the halo family's layers
        from propainter_amd.conv import ConvLayer
        seen, n_set = set(), 0

        def walk(o, depth=0):
            global n_set
            if id(o) in seen or depth > 6:
                return
            seen.add(id(o))
            if isinstance(o, ConvLayer):
                if o.kh * o.kw > 1 and tuple(o.stride) == (1, 1) and not getattr(o, "dcn", False) and o.impl == 0:
                    o.impl = int(IMPLS["gen3x3"])
                    n_set += 1
            elif isinstance(o, dict):
                for v in o.values():
                    walk(v, depth + 1)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    walk(v, depth + 1)
            elif hasattr(o, "__dict__") and type(o).__module__.startswith("propainter_amd"):
                for v in vars(o).values():
                    walk(v, depth + 1)
        walk(eng)
        print(f"REPLAY_NOTE {n_set} generator convolutions set to impl {IMPLS['gen3x3']}", flush=True)
ref = run_clip(models, clip, masks, masks, cfg, dev).clone()
eager_same = bool(torch.equal(run_clip(models, clip, masks, masks, cfg, dev), ref))
torch.cuda.synchronize()
g = None if EAGER else ClipGraph(models, L, H, W, cfg, dev, example=(clip, masks, masks), forked_branches=True)      # the lanes under test stay inside the capture
import time
t0 = time.time()
bad = []
for i in range(N):
    out = run_clip(models, clip, masks, masks, cfg, dev) if EAGER else g.replay()
    torch.cuda.synchronize()
    ne = out != ref
    if bool(ne.any()):
        d = (out.to(torch.int16) - ref.to(torch.int16)).abs()
        rec = {"replay": i, "bytes": int(ne.sum()), "max_abs": int(d.max()), "frames": [j for j in range(L) if bool(ne[j].any())][:10],
               "bytes_off_by_more_than_1": int((d > 1).sum())}
        bad.append(rec)
        print("REPLAY_DIFF " + json.dumps(rec), flush=True)
print("REPLAY_BYTES " + json.dumps({"replays": N, "mode": "eager passes" if EAGER else "graph replays", "ms_per_pass_incl_compare": round((time.time() - t0) / N * 1e3, 1), "deviating": len(bad), "window_streams": WS, "raft_streams": RS, "clip": f"{H}x{W}x{L}" + (" stress recipe" if STRESS else ""), "subvideo_length": SUB,
                                    "second_eager_pass_identical": eager_same, "lib": os.environ.get("PP_LIB_PATH", "in-tree"),
                                    "queues": os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES"), "prop_layer_impls": IMPLS or None,
                                    "env": {k: v for k, v in os.environ.items() if k in ("PP_CHAIN_IN_LANES", "PP_FENCE_EVERY_LAUNCH")}}), flush=True)
