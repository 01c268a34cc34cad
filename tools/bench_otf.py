#!/usr/bin/env python
"""Volume-free split-plane correlation lookup (pp_corr_lookup_otf_split) micro-benchmark at the 720p RAFT shape (90 x 160 maps).
Usage (GPU box): [PP_OTF_DBG=bits] python tools/bench_otf.py [--pairs 32] [--reps 10] [--flow smooth|zoom|still]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from propainter_amd import hip  # noqa: E402
from tests.cpu_emulation import split_planes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=32)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--h", type=int, default=90)
ap.add_argument("--w", type=int, default=160)
ap.add_argument("--flow", default="smooth")
args = ap.parse_args()
hip.lib()
P, h, w = args.pairs, args.h, args.w
g = torch.Generator().manual_seed(1)
f1 = split_planes(torch.randn(P, h, w, 256, generator=g)).cuda()
f2 = split_planes(torch.randn(P, h, w, 256, generator=g)).cuda()
ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
base = torch.stack([xs, ys], -1)[None].expand(P, h, w, 2)
if args.flow == "smooth":       # camera-like: translation + slow variation (< 1 level-0 cell across a tile)
    c = torch.nn.functional.interpolate(torch.randn(P, 2, h // 30, w // 30, generator=g) * 1.5, size=(h, w), mode="bilinear").permute(0, 2, 3, 1)
    coords = base + torch.tensor([2.3, -1.6]) + c
elif args.flow == "zoom":
    coords = (base - torch.tensor([w / 2, h / 2])) * 1.08 + torch.tensor([w / 2, h / 2])
else:
    coords = base + 0.25
coords = coords.contiguous().cuda()
lv = [f2] + hip.corr_feature_pyramid_split(f2)
out = torch.empty((P, h, w, 8 * hip.OTF_SPLIT_LEVEL_CHANNELS), dtype=torch.float16, device="cuda")
for _ in range(3):
    hip.corr_lookup_otf_split(f1, lv, coords, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    hip.corr_lookup_otf_split(f1, lv, coords, out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.reps
nblk = P * ((h + 7) // 8) * ((w + 7) // 8)
print(f"OTF_SPLIT dbg={os.environ.get('PP_OTF_DBG', '0')} flow={args.flow} P={P}: {ms:.3f} ms per launch = {ms * 1e3 / P:.1f} us per pair, "
      f"{ms * 1e3 * 256 / nblk:.1f} us per block-slot; x158 pairs x20 iterations = {ms / P * 158 * 20:.1f} ms per clip")

if int(os.environ.get("PP_OTF_DBG", "0")) & 16:
    import ctypes
    buf = (ctypes.c_uint64 * 8)()
    hip.lib().pp_debug_otf_prof(buf)
    nb = max(1, buf[7])
    names = ["prologue", "tables+loads+mfma+vstore", "barrier1", "blend", "barrier2", "writeout", "block total"]
    print("OTF_PROF cycles per block (wave 0): " + "  ".join(f"{n} {buf[i] / nb:.0f}" for i, n in enumerate(names)) + f"  | blocks {nb}")
