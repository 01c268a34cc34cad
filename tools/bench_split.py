#!/usr/bin/env python
"""Tile sweep of the split-plane ("f16x3") layers of RAFT at the 720p chunk shape (35 pair-directions x 90 x 160), the volume GEMM
and the two ways of building pyramid levels 1..3 (pooling the level-0 volume vs GEMMs with pooled features).  TUNING TOOL, not product.

For every layer: the dispatcher's choice (impl 0) and the alternative tile configurations; HIP events over `reps` back-to-back launches
(the queue stays full); max |diff| against impl 0 (all configurations walk K in the same order per output: bit-identical or ~1 ulp).
Usage (GPU box):  python tools/bench_split.py [--reps 20] [--pairs 35] > gpurun_out/split_sweep.txt
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
from propainter_amd.conv import ConvLayer, batched_gemm_nt_split  # noqa: E402
from tests.cpu_emulation import split_planes  # noqa: E402


def sp(shape, g, scale=1.0):
    """random split-plane tensor [..., 2 * C] on the device"""
    return split_planes(torch.randn(*shape, generator=g) * scale).cuda()


def timeit(fn, reps):
    # (a long warm-up: after the host-side packing of a layer the first ~20 ms of launches run ~10 % slow -- the first configuration of
    #  every layer in profiles/r3u_split_sweep.txt, which is the dispatcher's own choice measured first)
    for _ in range(max(2, reps)):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--pairs", type=int, default=35)
    ap.add_argument("--only", default="")
    ap.add_argument("--h8", type=int, default=90, help="feature-map height (1/8 of the frame): 90 = 720p, 135 = 1080p")
    ap.add_argument("--w8", type=int, default=160)
    args = ap.parse_args()
    hip.lib()
    print(torch.cuda.get_device_name(0), flush=True)
    g = torch.Generator().manual_seed(5)
    P, h, w = args.pairs, args.h8, args.w8
    # halo layers: the 128-pixel kernel (71: 128-cout tiles, 72: 64-cout tiles) against the ping-pong kernel (82 / 83, conv_halo8.h), each twice
    halo_alt = [71, 82, 71, 82]
    halo_alt64 = [72, 83, 72, 83]
    v2_alt = [0, 12, 13, 18, 22, 0, 18]
    # name, N, H, W, cin list, cout, k, stride, pad, impls, extras
    L = [
        ("convc1_1x1_324", P, h, w, [324], 256, (1, 1), 1, 0, v2_alt, dict(act="relu")),
        ("convc2_3x3_256_192", P, h, w, [256], 192, (3, 3), 1, 1, halo_alt64, dict(act="relu")),
        ("convf1_7x1_16_128", P, h, w, [16], 128, (7, 1), 1, (3, 0), v2_alt, dict(act="relu")),
        ("convf2_3x3_128_64", P, h, w, [128], 64, (3, 3), 1, 1, halo_alt64, dict(act="relu")),
        ("convm_3x3_256_126", P, h, w, [192, 64], 126, (3, 3), 1, 1, halo_alt64, dict(act="relu")),
        ("gru_zr_1x5", P, h, w, [128, 128], 256, (1, 5), 1, (0, 2), halo_alt, dict(act="sigmoid", gru="zr")),
        ("gru_q_1x5", P, h, w, [128, 128], 128, (1, 5), 1, (0, 2), halo_alt, dict(act="tanh", gru="h")),
        ("gru_zr_5x1", P, h, w, [128, 128], 256, (5, 1), 1, (2, 0), halo_alt, dict(act="sigmoid", gru="zr")),
        ("gru_q_5x1", P, h, w, [128, 128], 128, (5, 1), 1, (2, 0), halo_alt, dict(act="tanh", gru="h")),
        ("fh1_3x3_128_256", P, h, w, [128], 256, (3, 3), 1, 1, halo_alt, dict(act="relu")),
        ("fh2_3x3_256_2", P, h, w, [256], 2, (3, 3), 1, 1, [0, 110], dict(out_f32=True)),
        ("enc_7x7s2_3_64", 4, 720, 1280, [3], 64, (7, 7), 2, 3, [0, 22, 12], dict(out_f32=True)),
        ("enc_3x3_64_64", 4, 360, 640, [64], 64, (3, 3), 1, 1, halo_alt64, dict(out_f32=True)),
        ("enc_3x3s2_64_96", 4, 360, 640, [64], 96, (3, 3), 2, 1, v2_alt, dict(out_f32=True)),
        ("enc_3x3_96_96", 8, 180, 320, [96], 96, (3, 3), 1, 1, halo_alt64, dict(out_f32=True)),
        ("enc_3x3s2_96_128", 8, 180, 320, [96], 128, (3, 3), 2, 1, v2_alt, dict(out_f32=True)),
        ("enc_3x3_128_128", 16, 90, 160, [128], 128, (3, 3), 1, 1, halo_alt, dict(out_f32=True)),
        ("enc_1x1_128_256", 16, 90, 160, [128], 256, (1, 1), 1, 0, v2_alt, dict()),
    ]
    print(f"{'layer':22s} {'impl':>4s} {'us':>9s} {'TF(fp32-class)':>14s} {'max|d| vs impl 0':>17s}", flush=True)
    for name, N, H, W, cin, cout, k, stride, pad, impls, ex in L:
        if args.only and not re.search(args.only, name):
            continue
        kh, kw = k
        wt = torch.randn(cout, sum(cin), kh, kw, generator=g) / math.sqrt(sum(cin) * kh * kw)
        b = torch.randn(cout, generator=g) * 0.1
        layer = ConvLayer(wt, None if ex.get("gru") else b, stride=stride, padding=pad, src_channels=cin, dtype=torch.float16, device="cuda", split=True)
        srcs = [sp((N, H, W, (c + 7) // 8 * 8), g) for c in cin]
        for s_, c in zip(srcs, cin):       # padded channels hold zeros in both planes
            cp = s_.shape[-1] // 2
            s_[..., c:cp] = 0
            s_[..., cp + c:] = 0
        OH, OW = layer.out_hw(H, W)
        kw_ = dict(act=ex.get("act"))
        if ex.get("out_f32"):
            kw_["out_dtype"] = torch.float32
        if ex.get("gru"):
            kw_["preadd"] = sp((N, OH, OW, cout), g, 0.5)
            hbuf = sp((N, OH, OW, 128), g)
            if ex["gru"] == "zr":
                kw_["fuse"] = dict(kind="gru_zr", h=hbuf, out2=torch.empty((N, OH, OW, 256), dtype=torch.float16, device="cuda"), split=128)
            else:
                kw_["fuse"] = dict(kind="gru_h", h=hbuf, z=sp((N, OH, OW, 128), g, 0.3))
        flops = 2.0 * N * OH * OW * cout * sum(cin) * kh * kw
        ref = None
        for impl in impls:
            layer.impl = impl
            try:
                # (the z | r convolution writes z into a 128-channel split-plane buffer and r * h into out2, as the engine does)
                out0 = torch.empty((N, OH, OW, 256), dtype=torch.float16, device="cuda") if ex.get("gru") == "zr" else None
                out = layer(srcs, out=out0, **kw_)
                us = timeit(lambda: layer(srcs, out=out, **kw_), args.reps) * 1e3
            except RuntimeError as e:
                print(f"{name:22s} {impl:4d}  refused: {str(e)[:90]}", flush=True)
                continue
            if ref is None:
                ref, d = out.float().clone(), 0.0
            else:
                d = (out.float() - ref).abs().max().item()
            print(f"{name:22s} {impl:4d} {us:9.1f} {flops / us / 1e6:14.1f} {d:17.3e}", flush=True)
        del srcs, kw_, ref, out, layer
        torch.cuda.empty_cache()

    # ---- correlation pyramid of a chunk: level 0 GEMM tile choice, then levels 1..3 by pooling vs by pooled-feature GEMMs
    if not args.only or re.search(args.only, "volume"):
        n8 = h * w
        f1, f2 = sp((P, h, w, 256), g, 2.0), sp((P, h, w, 256), g, 2.0)
        vol = None
        for direct in ("0", "1"):       # accumulators staged through LDS (0) / stored directly (1: the shipped default for batched GEMMs)
            os.environ["PP_EPI_DIRECT"] = direct
            for impl in (0, 12, 13, 22):
                try:
                    ms = timeit(lambda: batched_gemm_nt_split(f1.view(P, n8, 512), f2.view(P, n8, 512), out_scale=1.0 / 16.0, impl=impl), max(3, args.reps // 4))
                except RuntimeError as e:
                    print(f"volume_level0 impl {impl}: refused {str(e)[:90]}")
                    continue
                print(f"volume_level0 direct-store {direct} impl {impl:3d}: {ms:8.3f} ms  ({2.0 * P * n8 * n8 * 256 / ms / 1e9:7.1f} TFLOP/s fp32-class, "
                      f"{P * n8 * n8 * 4 / ms / 1e6:7.1f} GB/s written)", flush=True)
        vol = batched_gemm_nt_split(f1.view(P, n8, 512), f2.view(P, n8, 512), out_scale=1.0 / 16.0)

        def by_pooling():
            lv, hh, ww = [vol.view(P * n8, h, w)], h, w
            for _ in range(3):
                lv.append(hip.corr_avgpool(lv[-1], P * n8, hh, ww))
                hh, ww = hh // 2, ww // 2
            return lv

        def by_gemm(impl=0):
            return [batched_gemm_nt_split(f1.view(P, n8, 512), fl.view(P, -1, 512), out_scale=1.0 / 16.0, impl=impl).view(P * n8, fl.shape[1], fl.shape[2])
                    for fl in hip.corr_feature_pyramid_split(f2)]

        for impl in (0, 12, 13, 22):
            try:
                ms = timeit(lambda: by_gemm(impl), max(3, args.reps // 4))
            except RuntimeError as e:
                print(f"levels 1..3 by pooled-feature GEMMs impl {impl}: refused {str(e)[:90]}")
                continue
            print(f"levels 1..3 by pooled-feature GEMMs, impl {impl:3d}: {ms:8.3f} ms", flush=True)
        print(f"levels 1..3 by pooling the volume:                {timeit(by_pooling, max(3, args.reps // 4)):8.3f} ms", flush=True)
        a, b_ = by_pooling(), by_gemm()
        for l in (1, 2, 3):
            d = (a[l] - b_[l - 1]).abs().max().item()
            print(f"  level {l}: max |pooled volume - pooled-feature GEMM| = {d:.3e} (values up to {a[l].abs().max().item():.3e})")

    # ---- bilinear x2 up-sampling (decoders): bytes = input read once + output written once
    if not args.only or re.search(args.only, "upsample"):
        for shape in ((11, 180, 320, 128), (11, 360, 640, 64), (158, 60, 108, 128)):
            x = torch.randn(*shape, generator=g).to("cuda", torch.float16)
            ms = timeit(lambda: hip.upsample2x(x), args.reps)
            print(f"upsample2x {shape}: {ms * 1e3:8.1f} us  {x.numel() * 2 * 5 / ms / 1e6:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
