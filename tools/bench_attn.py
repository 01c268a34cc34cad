#!/usr/bin/env python
"""Sparse window attention micro-benchmark at the 720p generator shape (t = 18, 60x108 token grid, 25 % masked windows).
Usage (GPU box): python tools/bench_attn.py [--impl 0|5] [--reps 10]   (impl 5 = phase-timing build)"""
import argparse, ctypes, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from propainter_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--impl", type=int, default=0)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--t", type=int, default=18)
args = ap.parse_args()
hip.lib()
dev = "cuda"
t, Hp, Wp, c = args.t, 60, 108, 512
g = torch.Generator().manual_seed(3)
qkv = (torch.randn(1, t, Hp, Wp, 3 * c, generator=g) * 0.5).to(dev, torch.float16)
P = (Hp // 4) * (Wp // 4)
pkv = (torch.randn(1, t, P, 2 * c, generator=g) * 0.5).to(dev, torch.float16)
own, rolled = hip.window_tables(Hp, Wp)
own, rolled = torch.from_numpy(own).to(dev), torch.from_numpy(rolled).to(dev)
nW = own.shape[0]
wmask = torch.zeros(1, nW, device=dev)
wy, wx = Hp // 5, Wp // 9
for y in range(wy // 4, wy // 4 + wy // 2):
    for x in range(wx // 4, wx // 4 + wx // 2):
        wmask[0, y * wx + x] = 1.0
tind = torch.arange(0, t, 2, dtype=torch.int32, device=dev)
run = lambda impl: hip.sparse_window_attention(qkv, qkv[..., c:], qkv[..., 2 * c:], pkv, pkv[..., c:], own, rolled, tind, wmask,
                                               qkv_cstride=3 * c, pkv_cstride=2 * c, C_=c, impl=impl)
ref = run(1)       # scalar reference kernel
torch.cuda.synchronize()
out = run(args.impl)
torch.cuda.synchronize()
print("max|d| vs the scalar reference kernel:", (out.float() - ref.float()).abs().max().item(), " masked windows:", int(wmask.sum().item()), "of", nW)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    run(args.impl)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.reps
nm = int(wmask.sum().item())
keys = tind.numel() * (45 + 148 + P)
fl = 4.0 * c * (nm * t * 45 * keys + (nW - nm) * t * 45 * 45)
print(f"impl {args.impl}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
if args.impl == 5:
    buf = (ctypes.c_uint64 * 8)()
    hip.lib().pp_debug_attn_prof(buf)
    w = max(1, buf[6]); tl = buf[7] / w
    names = ["store_tile", "barrier", "load issue", "S mfma", "softmax", "PV mfma"]
    print("per wave per tile (cycles): " + "  ".join(f"{n} {buf[i] / w / tl:.0f}" for i, n in enumerate(names)) + f"  | tiles {tl:.0f}")
