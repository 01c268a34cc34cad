#!/usr/bin/env python
"""Times pp_upsample2x at the generator decoder's shapes (720p: 11 frames x 180 x 320 x 128 and 360 x 640 x 64, fp16).  Tuning tool."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd import hip
hip.lib()
g = torch.Generator().manual_seed(0)
for n, h, w, c in ((11, 180, 320, 128), (11, 360, 640, 64), (2, 37, 41, 24)):
    x = torch.randn(n, h, w, c, generator=g).cuda().half()
    y = hip.upsample2x(x)
    ref = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.upsample2x(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"upsample2x {n} x {h} x {w} x {c}: {ms * 1e3:.1f} us per launch, {(x.numel() + y.numel()) * 2 / ms / 1e6:.0f} GB/s algorithmic, max |d| vs torch {(y.float() - ref).abs().max().item():.2e}")
