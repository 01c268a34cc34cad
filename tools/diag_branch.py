"""Which op gives different bits when its launches are captured on a FORKED branch of a hipGraph (a side stream that joins the capture at its
root, next to independent work on the capture stream) instead of on the capture stream itself?  Round 5: the single-graph streaming schedule is
wrong exactly when RAFT's fp16 / f16x3 engines run on a forked branch (profiles/r5_streaming_single_graph.txt)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd import hip                                                        # noqa: E402
from propainter_amd.synthetic import seeded_models, synthetic_clip                   # noqa: E402

dev = torch.device("cuda")
H, W, L = 128, 192, 11
raft, fc, gen = seeded_models(dev)
clip = synthetic_clip(L, H, W, seed=12)
fr = (torch.from_numpy(clip).to(dev).permute(0, 3, 1, 2).float().div(255)[None] * 2 - 1).contiguous()
side = torch.cuda.Stream(dev)
big = torch.randn(16 << 20, device=dev)


def check(name, fn, where="branch"):
    ref = [t.clone() for t in fn()]
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        cur = torch.cuda.current_stream(dev)
        if where == "branch":
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                out = fn()
            for _ in range(20):
                big.mul_(1.0001)                 # independent work on the capture stream
            cur.wait_stream(side)
        else:
            out = fn()
    bad = 0
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        bad += int(not all(torch.equal(a, b) for a, b in zip(out, ref)))
    worst = max(float((a.float() - b.float()).abs().max()) for a, b in zip(out, ref))
    print(f"BRANCH_DIAG {name} [{where}]: wrong in {bad} of 4 replays (max |d| {worst:.3e})", flush=True)
    return bad


for prec in ("f16", "f32", "f16x3"):
    raft.precision = prec
    check(f"RAFT {prec} forward", lambda: raft(fr, iters=3))
raft.precision = "f16"
check("RAFT f16 forward", lambda: raft(fr, iters=3), where="capture stream")
eng = raft._get_engine("f16", dev)
x = hip.nchw_to_nhwc(fr[0].contiguous(), out_dtype=torch.float16, cpad=8)
check("f16 fnet encoder (instance norm)", lambda: [eng.encode(eng.fnet, x, True)])
check("f16 cnet encoder", lambda: [eng.encode(eng.cnet, x, False)])
fm, cx = eng.encode(eng.fnet, x, True), eng.encode(eng.cnet, x, False)
torch.cuda.synchronize()
check("f16 refine iters=1", lambda: [eng.refine(fm[:-1], fm[1:], cx[:-1], 1)])
check("f16 refine iters=3", lambda: [eng.refine(fm[:-1], fm[1:], cx[:-1], 3)])
f2l = hip.corr_feature_pyramid(fm[1:])
check("corr_feature_pyramid", lambda: hip.corr_feature_pyramid(fm[1:]))
P, h, w = L - 1, H // 8, W // 8
ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
coords = (torch.stack([xs, ys], -1)[None].expand(P, h, w, 2) + 1.37).contiguous()


def lookup():
    out = torch.empty((P, h, w, 328), dtype=torch.float16, device=dev)
    hip.corr_lookup_otf(fm[:-1], f2l, coords, out)
    return [out]


check("corr_lookup_otf", lookup)
check("torch meshgrid/zeros/clone glue", lambda: [torch.stack(torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32),
                                                                             torch.arange(w, device=dev, dtype=torch.float32), indexing="ij"), -1).clone(),
                                                   torch.zeros((P, h, w, 8), device=dev) + 1])
