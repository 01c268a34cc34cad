# -*- coding: utf-8 -*-
"""Flow-completion evaluation -- the dataset caller of stages A + B (reference ``scripts/evaluate_flow_completion.py:54-197``):
for every video, RAFT flows of the UNMASKED frames are the ground truth (computed here in chunks of 60 frames, or read from
``--flow_root`` with ``--load_flow``: the ``.flo`` pairs of ``scripts/compute_flow.py``), the recurrent flow-completion net fills the
masked region (``forward_bidirect_flow`` + ``combine_flow``), and the report is the end-point error between the two

    [  1/ 50] Name: bear                      | EPE: 0.1234 | Time: 0.0123
    Finish evaluation... Average Frame EPE: 0.1234 | | Time: 0.0123

written to ``results_flow/<dataset>/<dataset>_metrics.txt``; ``--save_results`` adds colour-coded flow PNGs
(``forward_png/%05d.png``, ``backward_png/%05d.png``).  Data loading follows ``core/dataset.py:173-232`` (frames bilinear to
width x height, masks nearest -> > 0 -> dilated 4 x with the 3 x 3 cross).

The reference compares against the loader's ``flows_f`` / ``flows_b``, which exist only with ``--load_flow`` (``core/dataset.py:229-232``
returns the string 'None' otherwise); without it this script scores against the RAFT flows it just computed -- the same quantity
when the ``.flo`` files came from ``compute_flow.py``.  ``--load_flow`` is a flag here (the reference's ``type=bool`` turns any
non-empty string into True).

Engine extensions: ``--synthetic N`` / ``--frames`` (seeded synthetic clips; no datasets offline), ``--fp16`` (the completion net in
fp16, RAFT at fp32-class precision: the CLI's precision split).  Runs on the HIP engine only.
"""
import argparse
import os
import sys
from time import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def make_colorwheel():
    """The Middlebury colour wheel (Baker et al., "A Database and Evaluation Methodology for Optical Flow", ICCV 2007): 55 hues,
    RY 15, YG 6, GC 4, CB 11, BM 13, MR 6 -- what ``cvbase.flow2rgb`` (a dependency the reference does not vendor) draws with."""
    segs = [(15, 0, 1, +1), (6, 1, 0, -1), (4, 1, 2, +1), (11, 2, 1, -1), (13, 2, 0, +1), (6, 0, 2, -1)]
    rows = []
    for n, full, ramp, sign in segs:
        c = np.zeros((n, 3))
        c[:, full] = 1.0
        r = np.arange(n) / n
        c[:, ramp] = r if sign > 0 else 1.0 - r
        rows.append(c)
    return np.concatenate(rows, 0)


def flow2rgb(flow, unknown_thr=1e6):
    """(h, w, 2) flow -> (h, w, 3) float RGB in [0, 1]: hue = direction, saturation = magnitude / the frame's largest magnitude;
    NaN / |v| > unknown_thr pixels are black."""
    wheel = make_colorwheel()
    n = wheel.shape[0]
    dx, dy = flow[..., 0].astype(np.float64).copy(), flow[..., 1].astype(np.float64).copy()
    bad = np.isnan(dx) | np.isnan(dy) | (np.abs(dx) > unknown_thr) | (np.abs(dy) > unknown_thr)
    dx[bad] = 0
    dy[bad] = 0
    rad = np.sqrt(dx * dx + dy * dy)
    if rad.max() > np.finfo(float).eps:
        dx, dy, rad = dx / rad.max(), dy / rad.max(), rad / rad.max()
    ang = np.arctan2(-dy, -dx) / np.pi
    pos = (ang + 1) / 2 * (n - 1)
    lo = np.floor(pos).astype(int)
    hi = (lo + 1) % n
    w = (pos - lo)[..., None]
    col = (1 - w) * wheel[lo] + w * wheel[hi]
    small = rad <= 1
    col[small] = 1 - rad[small, None] * (1 - col[small])
    col[~small] *= 0.75
    col[bad] = 0
    return col


def save_flows(output, flows_f, flows_b):
    """flows_*: [t-1, 2, h, w] numpy -> <output>/{forward_png,backward_png}/%05d.png (:33-48)."""
    from PIL import Image
    for sub, fl in (("forward_png", flows_f), ("backward_png", flows_b)):
        os.makedirs(os.path.join(output, sub), exist_ok=True)
        for i in range(fl.shape[0]):
            vis = (flow2rgb(np.transpose(fl[i], (1, 2, 0))) * 255.0).astype(np.uint8)
            Image.fromarray(vis).save(os.path.join(output, sub, f'{i:05d}.png'))


def resize_flows(flows, h, w):
    """[t-1, 2, h0, w0] -> [t-1, 2, h, w], bilinear with the displacement rescaled (utils/flow_util.py:6-11, core/dataset.py:213-214)."""
    import torch
    import torch.nn.functional as F
    flows = torch.as_tensor(flows, dtype=torch.float32)
    h0, w0 = flows.shape[-2:]
    if (h0, w0) == (h, w):
        return flows
    flows = F.interpolate(flows, size=(h, w), mode='bilinear', align_corners=False)
    flows[:, 0] *= w / w0
    flows[:, 1] *= h / h0
    return flows


def raft_flows(fix_raft, frames, raft_iter, short_len=60):
    """frames [1, t, 3, h, w] in [-1, 1] -> (flows_f, flows_b) [1, t-1, 2, h, w]; clips longer than 60 frames in chunks of 60 that
    re-use the previous chunk's last frame (:94-110)."""
    import torch
    t = frames.size(1)
    if t <= short_len:
        return fix_raft(frames, iters=raft_iter)
    ff, fb = [], []
    for f in range(0, t, short_len):
        e = min(t, f + short_len)
        a, b = fix_raft(frames[:, max(f - 1, 0):e], iters=raft_iter)
        ff.append(a)
        fb.append(b)
    return torch.cat(ff, 1), torch.cat(fb, 1)


def evaluate(args, out=print):
    import torch
    from evaluate_propainter import load_video, synthetic_dataset
    from propainter_amd import flow_io, hip
    from propainter_amd.model.modules.flow_comp_raft import RAFT_bi, assert_finite_flows
    from propainter_amd.model.recurrent_flow_completion import RecurrentFlowCompleteNet
    from propainter_amd.synthetic import seeded_models
    if not torch.cuda.is_available():
        raise SystemExit("evaluate_flow_completion.py runs on the HIP engine only: no GPU visible")
    device = torch.device("cuda:0")
    hip.lib()
    size = (args.width, args.height)
    prec = "f16x3" if args.fp16 else "f32"
    if os.path.isfile(args.raft_model_path) and os.path.isfile(args.fc_model_path):
        fix_raft = RAFT_bi(args.raft_model_path, device, precision=prec)
        fix_flow_complete = RecurrentFlowCompleteNet(args.fc_model_path).to(device).eval()
    else:
        out("checkpoints not found (none ship offline): evaluating with the repo's seeded weights -- the numbers measure the engine, not a trained model")
        fix_raft, fix_flow_complete, _ = seeded_models(device, raft_precision=prec)
    if args.fp16:
        fix_flow_complete = fix_flow_complete.half()
    if args.synthetic:
        videos = list(synthetic_dataset(args.synthetic, args.frames, size))
        dataset = "synthetic"
    else:
        assert args.dataset in ('davis', 'youtube-vos'), f"{args.dataset} dataset is not supported"
        dataset = args.dataset
        videos = [(v,) + load_video(args.video_root, args.mask_root, v, size) for v in sorted(os.listdir(args.mask_root))]
    if args.load_flow:
        assert os.path.exists(args.flow_root), args.flow_root
    out('Start evaluation...')
    result_path = os.path.join(args.result_root, f'{dataset}')
    os.makedirs(result_path, exist_ok=True)
    eval_summary = open(os.path.join(result_path, f"{dataset}_metrics.txt"), "w")
    total_frame_epe, time_all = [], []
    avg_time = float('nan')
    for index, (video_name, frames_u8, masks_u8) in enumerate(videos):
        L = len(frames_u8)
        frames = torch.from_numpy(np.ascontiguousarray(frames_u8)).to(device).permute(0, 3, 1, 2).float().div(255.0)[None] * 2.0 - 1.0
        local_masks = torch.from_numpy(np.ascontiguousarray(masks_u8)).to(device).float().div(255.0)[None, :, None]
        with torch.no_grad():
            if args.load_flow:
                lf, lb = flow_io.load_clip_flows(os.path.join(args.flow_root, video_name))
                gt_flows_bi = (resize_flows(lf, args.height, args.width)[None].to(device),
                               resize_flows(lb, args.height, args.width)[None].to(device))
            else:
                gt_flows_bi = raft_flows(fix_raft, frames, args.raft_iter)
                assert_finite_flows(fix_raft)
            gt_flows_bi = tuple(g.float() for g in gt_flows_bi)
            work = tuple(g.half() for g in gt_flows_bi) if args.fp16 else gt_flows_bi
            masks = local_masks.half() if args.fp16 else local_masks
            torch.cuda.synchronize()
            time_start = time()
            pred_flows_bi, _ = fix_flow_complete.forward_bidirect_flow(work, masks)
            pred_flows_bi = fix_flow_complete.combine_flow(work, pred_flows_bi, masks)
            torch.cuda.synchronize()
        time_i = (time() - time_start) / L
        time_all += [time_i] * L
        epe1 = torch.mean(torch.sum((gt_flows_bi[0] - pred_flows_bi[0].float()) ** 2, dim=2).sqrt()).item()
        epe2 = torch.mean(torch.sum((gt_flows_bi[1] - pred_flows_bi[1].float()) ** 2, dim=2).sqrt()).item()
        if not (np.isfinite(epe1) and np.isfinite(epe2)):
            raise FloatingPointError(f"{video_name}: non-finite completed flows")
        total_frame_epe += [epe1] * (L - 1) + [epe2] * (L - 1)
        cur_epe = (epe1 + epe2) / 2
        avg_time = sum(time_all) / len(time_all)
        line = f'[{index + 1:3}/{len(videos)}] Name: {str(video_name):25} | EPE: {cur_epe:.4f} | Time: {avg_time:.4f}'
        out(line)
        eval_summary.write(line + '\n')
        if args.save_results:
            save_flows(os.path.join(result_path, video_name), pred_flows_bi[0][0].float().cpu().numpy(), pred_flows_bi[1][0].float().cpu().numpy())
    avg_frame_epe = sum(total_frame_epe) / len(total_frame_epe)
    line = f'Finish evaluation... Average Frame EPE: {avg_frame_epe:.4f} | | Time: {avg_time:.4f}'
    out(line)
    eval_summary.write(line + '\n')
    eval_summary.close()
    return dict(epe=avg_frame_epe, time=avg_time, path=result_path)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--height', type=int, default=240)
    parser.add_argument('--width', type=int, default=432)
    parser.add_argument('--raft_model_path', default='weights/raft-things.pth', type=str)
    parser.add_argument('--fc_model_path', default='weights/recurrent_flow_completion.pth', type=str)
    parser.add_argument('--dataset', choices=['davis', 'youtube-vos'], type=str)
    parser.add_argument('--video_root', default='dataset_root', type=str)
    parser.add_argument('--mask_root', default='mask_root', type=str)
    parser.add_argument('--flow_root', default='flow_ground_truth_root', type=str)
    parser.add_argument('--load_flow', action='store_true')
    parser.add_argument("--raft_iter", type=int, default=20)
    parser.add_argument('--save_results', action='store_true')
    parser.add_argument('--num_workers', default=4, type=int, help='accepted for compatibility (frames are read in-process)')
    parser.add_argument('--result_root', default='results_flow', type=str)
    parser.add_argument('--fp16', action='store_true')
    parser.add_argument('--synthetic', type=int, default=0, help='evaluate this many seeded synthetic clips instead of a dataset')
    parser.add_argument('--frames', type=int, default=24, help='length of the synthetic clips')
    return parser


if __name__ == '__main__':
    evaluate(build_parser().parse_args())
