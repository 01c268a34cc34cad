#!/usr/bin/env python
"""Transformer Linear layers of the 720p generator window (M = tokens of 17 frames): the engine's LDS-DMA GEMM against the vendor
library (torch.nn.functional.linear = hipBLASLt / rocBLAS on ROCm), fp16 in / fp32 accumulate / fp16 out.  Tuning tool."""
import math, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd.conv import ConvLayer

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
M = int(os.environ.get("M", "109140"))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, K, N in (("qkv", 512, 1536), ("proj", 512, 512), ("fc1", 512, 1960), ("fc1_pad", 512, 2048), ("sc_embed", 512, 6272), ("k1960", 1960, 512)):
    w = (torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.1
    x = torch.randn(M, K, generator=g).to(dev, torch.float16)
    layer = ConvLayer(w.view(N, K, 1, 1), b, src_channels=[K], dtype=torch.float16, device=dev)
    xs = [x.view(1, 1, M, K)]
    out = layer(xs)
    wd, bd = w.to(dev, torch.float16), b.to(dev, torch.float16)
    ref = F.linear(x, wd, bd)
    err = (out.view(M, -1)[:, :N].float() - ref.float()).abs().max().item()
    t_mine = timeit(lambda: layer(xs, out=out))
    t_lib = timeit(lambda: F.linear(x, wd, bd))
    t_mm = timeit(lambda: torch.mm(x, wd.t()))
    fl = 2.0 * M * K * N
    if N % 64:                                             # output rows padded to a multiple of 128 bytes (same couts, aligned stores)
        Np = (N + 63) // 64 * 64
        outp = torch.zeros(1, 1, M, Np, dtype=torch.float16, device=dev)
        layer(xs, out=outp)
        assert torch.equal(outp[..., :N], out[..., :N])
        t_al = timeit(lambda: layer(xs, out=outp))
        print(f"LINEAR {name:8s} engine with the output row stride {Np}: {t_al * 1e3:7.1f} us = {fl / t_al / 1e9:6.0f} TFLOP/s")
    print(f"LINEAR {name:8s} M {M} K {K} N {N}: engine {t_mine * 1e3:7.1f} us = {fl / t_mine / 1e9:6.0f} TFLOP/s | F.linear {t_lib * 1e3:7.1f} us = {fl / t_lib / 1e9:6.0f} | "
          f"mm (no bias) {t_mm * 1e3:7.1f} us = {fl / t_mm / 1e9:6.0f} | max |engine - lib| {err:.3g}")
