#!/usr/bin/env python
"""Times the deformable-convolution kernels (patch-staged impl 90 vs register-staged impl 1) on generator / flow-completion
shapes for several offset distributions.  Tuning tool."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd.conv import ConvLayer

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
NB = int(os.environ.get("NB", "8"))      # batch: the GPU time must dominate the ~150 us of Python per launch (NB=2: the flow-completion step shape)
for name, cin, H, W in (("gen", [128], 180, 320), ("fc", [128, 128], 90, 160)):
    ctot = sum(cin)
    w = torch.randn(128, ctot, 3, 3, generator=g) / math.sqrt(ctot * 9)
    b = torch.zeros(128)
    layer = ConvLayer(w, b, padding=1, src_channels=cin, dcn_groups=16, dtype=torch.float16, device=dev)
    srcs = [torch.randn(NB, H, W, c, generator=g).to(dev, torch.float16) for c in cin]
    for dist in ("zero", "small", "tanh3+flow", "wild"):
        off = torch.zeros(NB, H, W, 288)
        if dist == "small":
            off = torch.randn(NB, H, W, 288, generator=g) * 0.7
        elif dist == "tanh3+flow":
            off = 3 * torch.tanh(torch.randn(NB, H, W, 288, generator=g)) + torch.tensor([2.5, -3.5]).repeat(144)
        elif dist == "wild":
            off = (torch.rand(NB, H, W, 288, generator=g) * 2 - 1) * 12
        om = torch.cat([off, torch.rand(NB, H, W, 144, generator=g)], -1).to(dev, torch.float16).contiguous()
        # 90 = patch-staged kernel, 1 = register-staged gather; 90 + flags = ablations of the patch-staged kernel
        # (1 no offset staging, 2 no patch staging, 4 no sampling, 8 no weights / MFMA: results are garbage, timing only)
        for impl in ((90, 106, 122, 1) if dist in ('small', 'tanh3+flow') else (90, 1)):      # 106 / 122: 128- / 64-pixel tiles forced (90: by launch size)
            layer.impl = impl
            for _ in range(3):
                layer(srcs, dcn_offmask=om)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                layer(srcs, dcn_offmask=om)
            e1.record()
            torch.cuda.synchronize()
            print(f"{name:4s} {H}x{W} offsets {dist:11s} impl {impl:2d}: {e0.elapsed_time(e1) / 20 * 1e3 / NB:8.1f} us per image")
