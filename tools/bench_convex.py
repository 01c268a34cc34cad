#!/usr/bin/env python
"""Times pp_convex_upsample at the 720p RAFT chunk shape (79 pair-directions x 90 x 160, fp32 mask of 576 logits per coarse pixel).  Tuning tool."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd import hip
hip.lib()
g = torch.Generator().manual_seed(0)
P, h, w = 79, 90, 160
flow = torch.randn(P, h, w, 2, generator=g).cuda()
mask = torch.randn(P, h, w, 576, generator=g).cuda()
out = hip.convex_upsample(flow, mask)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    hip.convex_upsample(flow, mask)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"convex_upsample {P} x {h} x {w}: {ms * 1e3:.1f} us per launch, {(mask.numel() * 4 + out.numel() * 4) / ms / 1e6:.0f} GB/s algorithmic")
