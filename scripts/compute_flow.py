# -*- coding: utf-8 -*-
"""Pre-compute RAFT flows of a dataset into ``.flo`` files -- the third caller of stage A (reference
``scripts/compute_flow.py:39-108``): for every video folder under ``--root_path`` and every pair of consecutive frames,
forward flow ``<save_path>/<video>/<cur>_<next>_f.flo`` and backward flow ``<save_path>/<video>/<next>_<cur>_b.flo`` at
``--height`` x ``--width`` (frames resized bilinearly, align_corners=False, then scaled to [-1, 1]; RAFT with 20 iterations; files
in the float16 ``PIEH`` format of ``utils/flow_util.py:28-89``).  These are the files ``evaluate_propainter.py --load_flow`` and
``evaluate_flow_completion.py --load_flow`` read back (``core/dataset.py:206-216``).

The reference runs RAFT once per pair and direction; RAFT here is batch-invariant (InstanceNorm per sample, folded BatchNorm), so a
whole clip goes through ``RAFT_bi`` in chunks of ``--chunk`` frames -- same flows, each frame encoded once.

Engine extensions: ``--raft_model_path`` (the reference hard-codes ``weights/raft-things.pth``), ``--precision`` (f32 | f16x3 | f16,
default f32 like the reference), ``--synthetic N`` / ``--frames`` (write N seeded synthetic clips into ``--root_path`` first: there
are no datasets offline).  Runs on the HIP engine only.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def frame_stem(name):
    """The reference strips a 4-character extension (``m_list[i][:-4]``, compute_flow.py:100-101)."""
    return name[:-4]


def clip_flows(fix_raft, frames_u8, size, device, iters=20, chunk=60):
    """frames_u8: uint8 [t, H, W, 3] RGB -> (flows_f, flows_b) float32 numpy [t-1, h, w, 2] at size = (h, w).
    Pre-processing of compute_flow.py:71-89: ToTensor (/255), bilinear resize with align_corners=False, x * 2 - 1."""
    import torch
    import torch.nn.functional as F
    x = torch.from_numpy(np.ascontiguousarray(frames_u8)).to(device).permute(0, 3, 1, 2).float().div(255.0)
    x = F.interpolate(x, size=size, mode='bilinear', align_corners=False) * 2 - 1
    t = x.shape[0]
    ff, fb = [], []
    for f in range(0, t - 1, chunk - 1):           # chunks share their boundary frame: every consecutive pair exactly once
        e = min(t, f + chunk)
        a, b = fix_raft(x[None, f:e], iters=iters)
        ff.append(a[0])
        fb.append(b[0])
    ff, fb = torch.cat(ff), torch.cat(fb)
    assert ff.shape[0] == t - 1
    return ff.permute(0, 2, 3, 1).float().cpu().numpy(), fb.permute(0, 2, 3, 1).float().cpu().numpy()


def write_synthetic_dataset(root, n, frames, size):
    from PIL import Image
    from propainter_amd.synthetic import synthetic_clip
    h, w = size
    for i in range(n):
        d = os.path.join(root, f"synthetic_{i:02d}")
        os.makedirs(d, exist_ok=True)
        for j, fr in enumerate(synthetic_clip(frames, h, w, seed=100 + i)):
            Image.fromarray(fr).save(os.path.join(d, f"{j:05d}.png"))


def main(argv=None, out=print):
    parser = argparse.ArgumentParser()
    parser.add_argument('-i', '--root_path', type=str, default='your_dataset_root/youtube-vos/JPEGImages')
    parser.add_argument('-o', '--save_path', type=str, default='your_dataset_root/youtube-vos/Flows_flo')
    parser.add_argument('--height', type=int, default=240)
    parser.add_argument('--width', type=int, default=432)
    parser.add_argument('--raft_model_path', type=str, default='weights/raft-things.pth')
    parser.add_argument('--precision', choices=['f32', 'f16x3', 'f16'], default='f32')
    parser.add_argument('--chunk', type=int, default=60, help='frames per RAFT_bi call')
    parser.add_argument('--synthetic', type=int, default=0, help='first write this many seeded synthetic clips into --root_path')
    parser.add_argument('--frames', type=int, default=12, help='length of the synthetic clips')
    args = parser.parse_args(argv)

    import torch
    from PIL import Image
    from propainter_amd import flow_io, hip
    from propainter_amd.model.modules.flow_comp_raft import RAFT_bi, assert_finite_flows
    from propainter_amd.synthetic import seeded_models
    if not torch.cuda.is_available():
        raise SystemExit("compute_flow.py runs on the HIP engine only: no GPU visible")
    device = torch.device("cuda:0")
    hip.lib()
    if os.path.isfile(args.raft_model_path):
        fix_raft = RAFT_bi(args.raft_model_path, device, precision=args.precision)
    else:
        out("checkpoint not found (none ship offline): RAFT with the repo's seeded weights")
        fix_raft = seeded_models(device, raft_precision=args.precision)[0]
    size = (args.height, args.width)
    if args.synthetic:
        write_synthetic_dataset(args.root_path, args.synthetic, args.frames, size)
    written = 0
    for f in sorted(os.listdir(args.root_path)):
        out(f'Processing: {f} ...')
        m_list = sorted(os.listdir(os.path.join(args.root_path, f)))
        if len(m_list) < 2:
            continue
        frames = np.stack([np.asarray(Image.open(os.path.join(args.root_path, f, m)).convert('RGB'), dtype=np.uint8) for m in m_list])
        with torch.no_grad():
            flows_f, flows_b = clip_flows(fix_raft, frames, size, device, iters=20, chunk=args.chunk)
        torch.cuda.synchronize()
        assert_finite_flows(fix_raft)
        for i in range(len(m_list) - 1):
            cur, nxt = frame_stem(m_list[i]), frame_stem(m_list[i + 1])
            flow_io.flowwrite(flows_f[i], os.path.join(args.save_path, f, f'{cur}_{nxt}_f.flo'))
            flow_io.flowwrite(flows_b[i], os.path.join(args.save_path, f, f'{nxt}_{cur}_b.flo'))
            written += 2
    out(f'wrote {written} flow files under {args.save_path}')
    return written


if __name__ == '__main__':
    main()
