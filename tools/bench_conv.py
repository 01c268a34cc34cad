#!/usr/bin/env python
"""Tile sweep + cross-check of the implicit-GEMM convolution kernels on the layer shapes of the 720p path.

For every shape: run the register-staged kernel (impl=1) as the reference result, then every LDS-DMA tile
configuration; report max |diff| against impl=1 and the achieved TFLOP/s (HIP events, median of `reps`).
Usage (GPU box):  python tools/bench_conv.py [--reps 20] > gpurun_out/conv_sweep.txt
"""
import argparse
import math
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from propainter_amd import hip  # noqa: E402
from propainter_amd.conv import ConvLayer  # noqa: E402

SHAPES = [
    # name, N, H, W, cin list, cout, k, stride, pad, groups
    ("big_gru_zr_1x5_n36", 36, 90, 160, [128, 256], 256, (1, 5), 1, (0, 2), 1),
    ("big_gru_q_5x1_n36", 36, 90, 160, [128, 256], 128, (5, 1), 1, (2, 0), 1),
    ("raft_gru_zr_1x5", 8, 90, 160, [128, 256], 256, (1, 5), 1, (0, 2), 1),
    ("raft_gru_q_5x1", 8, 90, 160, [128, 256], 128, (5, 1), 1, (2, 0), 1),
    ("raft_convc1_1x1", 8, 90, 160, [324], 256, (1, 1), 1, 0, 1),
    ("raft_convc2_3x3", 8, 90, 160, [256], 192, (3, 3), 1, 1, 1),
    ("raft_convm_3x3", 8, 90, 160, [192, 64], 126, (3, 3), 1, 1, 1),
    ("raft_fh1_3x3", 8, 90, 160, [128], 256, (3, 3), 1, 1, 1),
    ("raft_fh2_cout2", 8, 90, 160, [256], 2, (3, 3), 1, 1, 1),
    ("raft_convf1_7x7_c2", 8, 90, 160, [2], 128, (7, 7), 1, 3, 1),
    ("raft_enc_7x7s2", 5, 720, 1280, [3], 64, (7, 7), 2, 3, 1),
    ("raft_enc_64_3x3", 5, 360, 640, [64], 64, (3, 3), 1, 1, 1),
    ("gen_off0_3x3", 1, 180, 320, [128, 128, 5], 128, (3, 3), 1, 1, 1),
    ("gen_off6_cout432", 1, 180, 320, [128], 432, (3, 3), 1, 1, 1),
    ("gen_softsplit_7x7s3", 18, 180, 320, [128], 512, (7, 7), 3, 3, 1),
    ("gen_qkv_linear", 1, 1, 116640, [512], 1536, (1, 1), 1, 0, 1),
    ("gen_fc1_linear", 1, 1, 115560, [512], 1960, (1, 1), 1, 0, 1),
    ("gen_fc2_7x7s3_c40", 18, 180, 320, [40], 512, (7, 7), 3, 3, 1),
    ("gen_sc_embed", 1, 1, 115560, [512], 6272, (1, 1), 1, 0, 1),
    ("gen_enc_grouped8", 18, 180, 320, [32, 48], 256, (3, 3), 1, 1, 8),
    ("gen_enc_128_s2", 18, 360, 640, [64], 128, (3, 3), 2, 1, 1),
    ("gen_dec_64_cout3", 11, 720, 1280, [64], 3, (3, 3), 1, 1, 1),
    ("fc_dec_32_3x3", 16, 360, 640, [32], 32, (3, 3), 1, 1, 1),
    # the batched feature propagation of the generator windows (round 4: 14 windows per launch) and the decoder at 11 local frames
    ("prop_off0_n14", 14, 180, 320, [128, 128, 5], 128, (3, 3), 1, 1, 1),
    ("prop_off2_n14", 14, 180, 320, [128], 128, (3, 3), 1, 1, 1),
    ("prop_off6_n14", 14, 180, 320, [128], 432, (3, 3), 1, 1, 1),
    ("dec_128_64_n11", 11, 360, 640, [128], 64, (3, 3), 1, 1, 1),
    ("dec_64_64_n11", 11, 720, 1280, [64], 64, (3, 3), 1, 1, 1),
]
IMPLS = {"cout>64": [1, 112, 12, 10, 11, 13, 14, 17, 18, 19], "cout>32": [1, 122, 22, 20, 21], "cout>16": [1, 132, 32, 30, 31], "cout<=16": [1, 142, 42, 40, 41]}


def impls_for(cout_g):
    if cout_g > 64:
        return IMPLS["cout>64"]
    if cout_g > 32:
        return IMPLS["cout>32"]
    if cout_g > 16:
        return IMPLS["cout>16"]
    return IMPLS["cout<=16"]


def run(layer, srcs, impl, reps):
    """One warm-up launch, then `reps` back-to-back launches between two HIP events (the queue stays full, so the
    Python launch overhead is hidden and the figure is GPU time per launch)."""
    layer.impl = impl
    out = layer(srcs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        layer(srcs, out=out)
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / reps


def denormal_probe():
    """Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs?  1x1 conv, inputs 2^-20 (subnormal), weights 1."""
    x = torch.full((1, 16, 16, 64), 2.0 ** -20, dtype=torch.float16, device="cuda")
    w = torch.ones(16, 64, 1, 1)
    for impl in (1, 0):
        layer = ConvLayer(w, None, dtype=torch.float16, device="cuda")
        layer.impl = impl
        y = layer([x], out_dtype=torch.float32)
        torch.cuda.synchronize()
        print(f"denormal probe impl={impl}: sum of 64 x 2^-20 = {y[0, 0, 0, 0].item():.6e} (expected {64 * 2.0 ** -20:.6e})")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--impls", default="", help="comma-separated impl ids to run instead of the per-shape defaults")
    args = ap.parse_args()
    hip.lib()
    print(torch.cuda.get_device_name(0))
    denormal_probe()
    g = torch.Generator().manual_seed(5)
    print(f"{'shape':24s} {'impl':>4s} {'ms':>9s} {'TFLOP/s':>9s} {'max|d| vs impl1':>16s}")
    for name, N, H, W, cin, cout, k, stride, pad, groups in SHAPES:
        if args.only and not re.search(args.only, name):
            continue
        kh, kw = k
        wt = torch.randn(cout, sum(cin), kh, kw, generator=g) / math.sqrt(sum(cin) * kh * kw)
        b = torch.randn(cout, generator=g) * 0.1
        layer = ConvLayer(wt, b, stride=stride, padding=pad, groups=groups, src_channels=cin, dtype=torch.float16, device="cuda")
        srcs = []
        for c in cin:
            cp = (c * groups + 7) // 8 * 8 if groups == 1 else c * groups
            t = torch.zeros(N, H, W, cp, dtype=torch.float16, device="cuda")
            t[..., :c * groups] = torch.randn(N, H, W, c * groups, generator=g).to("cuda", torch.float16)
            srcs.append(t)
        OH, OW = layer.out_hw(H, W)
        flops = 2.0 * N * OH * OW * cout * (sum(cin) * kh * kw)
        ref = None
        for impl in ([int(v) for v in args.impls.split(',')] if args.impls else impls_for(cout // groups)):
            try:
                out, ms = run(layer, srcs, impl, args.reps)
            except RuntimeError as e:
                print(f"{name:24s} {impl:4d}  FAILED {e}")
                continue
            if ref is None:
                ref = out.float().clone()
                d = 0.0
            else:
                d = (out.float() - ref).abs().max().item()
            print(f"{name:24s} {impl:4d} {ms:9.3f} {flops / ms / 1e9:9.1f} {d:16.3e}", flush=True)
            if impl in (75, 77, 78):      # phase counters of the PROF build (cycles per wave)
                import ctypes
                buf = (ctypes.c_uint64 * 16)()
                hip.lib().pp_debug_conv_prof(buf)
                w = max(1, buf[6])
                print("    per wave: vmcnt-wait %.0f  barrier-wait %.0f  dma-issue %.0f  compute %.0f  | loop %.0f  epilogue %.0f  (K steps %.0f, waves %d)"
                      % (buf[0] / w, buf[1] / w, buf[2] / w, buf[3] / w, buf[4] / w, buf[5] / w, buf[7] / w, w))
                print("    epilogue: sync %.0f  acc->lds %.0f  bias/act/store %.0f | setup before loop %.0f" % (buf[8] / w, buf[9] / w, buf[10] / w, buf[11] / w))


if __name__ == "__main__":
    main()
