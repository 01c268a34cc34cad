#!/usr/bin/env python
"""fp16-MFMA RAFT vs the fp32 CPU oracle at the headline resolution (720x1280, 20 iterations, one frame pair of the
synthetic clip): end-point error of the engine in fp32 and in its default fp16 mode.  ~1 min on the GPU box."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import propainter_oracle as O                     # the checker (this is a test tool, not the product)
from propainter_amd.synthetic import seeded_models, synthetic_clip

H, W, iters = 720, 1280, 20
clip = synthetic_clip(2, H, W)
fr = torch.from_numpy(clip).permute(0, 3, 1, 2).float().div(255)[None] * 2 - 1
raft, _, _ = seeded_models("cuda")
sd = {k: v.float().cpu() for k, v in raft.fix_raft.state_dict().items()}
torch.set_num_threads(min(32, os.cpu_count() or 1))
t0 = time.perf_counter()
with torch.no_grad():
    ref_f, ref_b = O.raft_bi(sd, fr, iters=iters)
t_cpu = time.perf_counter() - t0
out = {"H": H, "W": W, "iters": iters, "cpu_oracle_seconds": t_cpu, "flow_abs_max": float(ref_f.abs().max())}
for name, dt in (("f32", None), ("f16", torch.float16)):
    raft.compute_dtype = dt
    ff, fb = raft(fr.cuda(), iters=iters)
    torch.cuda.synchronize()
    for tag, a, b in (("fwd", ff, ref_f), ("bwd", fb, ref_b)):
        epe = (a.float().cpu() - b).pow(2).sum(2).sqrt()
        out[f"epe_{name}_{tag}"] = {"mean": float(epe.mean()), "p99": float(epe.flatten().quantile(0.99)), "max": float(epe.max())}
print(json.dumps(out))
