#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Evaluation entry point on the MI355X engine: the protocol and the report of the reference's
``scripts/evaluate_propainter.py``.

Same flags (``--dataset --video_root --mask_root --height --width --ref_stride --neighbor_length --raft_iter --task
--*_model_path --save_results``), same protocol (:103-178 of the reference):
  * frames resized to ``--width x --height`` (432 x 240), masks ``> 0`` dilated 4x with a 3x3 cross (core/dataset.py:193-204);
  * RAFT on the whole video (the reference cuts it into 60-frame pieces with one frame of overlap -- the same flows: every pair is
    computed independently), flow completion and image propagation on the WHOLE video (no sub-video chunking), generator windows
    of ``neighbor_length // 2`` stride with ALL ``ref_stride`` frames of the video as references (get_ref_index:29-35);
  * composite: uint8 prediction inside the mask, frames covered by several windows averaged in float32 WITHOUT truncation
    (:170-178; the inference script truncates after every blend);
  * per video ``[  i/  N] Name: ... | PSNR/SSIM: .../... | Avg PSNR/SSIM: .../... | Time: ...`` (Time = seconds per frame, averaged
    over the videos so far: :181-222), the closing ``Finish evaluation... Average Frame PSNR/SSIM/VFID`` line, and the same lines in
    ``results_eval/<dataset>_rs_<rs>_nl_<nl>_video_completion/<dataset>_metrics.txt``.
PSNR and SSIM follow core/metrics.py:20-54 (SSIM = scikit-image's ``compare_ssim(data_range=255, multichannel=True, win_size=65)``:
uniform 65 x 65 windows, sample covariance, border of 32 pixels cropped -- restated here on scipy because scikit-image is not in
the image).  VFID needs the I3D checkpoint ``weights/i3d_rgb_imagenet.pt`` the reference downloads and the I3D network
(core/metrics.py:57-571): neither is part of the inference hot path; the field prints ``nan`` and the report says so.

Engine extensions: ``--fp16`` (stages B-D in fp16, RAFT at fp32-class precision: the CLI's precision split), ``--synthetic N``
(evaluate N seeded synthetic clips instead of a dataset: there are no datasets or checkpoints offline), ``--frames`` (their length).
"""
import argparse
import os
import sys
from time import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def calculate_psnr(img1, img2):
    """core/metrics.py:20-36."""
    assert img1.shape == img2.shape, f'Image shapes are differnet: {img1.shape}, {img2.shape}.'
    mse = np.mean((img1 - img2) ** 2)
    if mse == 0:
        return float('inf')
    return 20. * np.log10(255. / np.sqrt(mse))


def structural_similarity(img1, img2, data_range=255.0, win_size=65):
    """scikit-image's ``compare_ssim(im1, im2, data_range=255, multichannel=True, win_size=65)`` as core/metrics.py:47-51 calls it:
    per channel, uniform win_size x win_size filters (reflect borders), SAMPLE covariance (N / (N - 1)), K1 = 0.01, K2 = 0.03, the
    mean of the SSIM map with a border of (win_size - 1) // 2 pixels cropped; channels averaged."""
    from scipy.ndimage import uniform_filter
    assert img1.shape == img2.shape and img1.ndim == 3
    if min(img1.shape[:2]) < win_size:
        raise ValueError(f"win_size {win_size} exceeds the image extent {img1.shape[:2]}")
    NP = win_size ** 2
    cov_norm = NP / (NP - 1.0)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    pad = (win_size - 1) // 2
    vals = []
    for ch in range(img1.shape[2]):
        X, Y = img1[..., ch].astype(np.float64), img2[..., ch].astype(np.float64)
        ux, uy = uniform_filter(X, size=win_size), uniform_filter(Y, size=win_size)
        uxx, uyy, uxy = uniform_filter(X * X, size=win_size), uniform_filter(Y * Y, size=win_size), uniform_filter(X * Y, size=win_size)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        vals.append(S[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


def calc_psnr_and_ssim(img1, img2):
    """core/metrics.py:39-54."""
    img1, img2 = img1.astype(np.float64), img2.astype(np.float64)
    return calculate_psnr(img1, img2), structural_similarity(img1, img2, data_range=255, win_size=65)


def load_video(video_root, mask_root, name, size):
    """core/dataset.py:173-204 (TestDataset.__getitem__): RGB frames resized to `size` (bilinear), masks nearest -> > 0 -> dilated
    4 x with the 3 x 3 cross.  Returns uint8 frames [L,H,W,3] and masks [L,H,W] {0,255}."""
    import scipy.ndimage
    from PIL import Image
    from propainter_amd import video_io
    frame_list = sorted(os.listdir(os.path.join(video_root, name)))
    frames, masks = [], []
    for idx, fn in enumerate(frame_list):
        # cv2.resize(img, size, INTER_LINEAR) (core/dataset.py:186-188): NO antialiasing when shrinking -- PIL's BILINEAR filters over the
        # whole footprint, which changes both the model inputs and the ground truth of the PSNR / SSIM report
        img = video_io.resize_u8_linear(np.asarray(Image.open(os.path.join(video_root, name, fn)).convert('RGB'), dtype=np.uint8), size)
        frames.append(img)
        m = np.asarray(Image.open(os.path.join(mask_root, name, str(idx).zfill(5) + '.png')).resize(size, Image.NEAREST).convert('L'))
        masks.append(scipy.ndimage.binary_dilation(m > 0, iterations=4).astype(np.uint8) * 255)
    return np.stack(frames), np.stack(masks)


def synthetic_dataset(n, frames, size):
    import scipy.ndimage
    from propainter_amd.synthetic import synthetic_clip, synthetic_mask
    w, h = size
    for i in range(n):
        m = scipy.ndimage.binary_dilation(synthetic_mask(h, w) > 0, iterations=4).astype(np.uint8) * 255
        yield f"synthetic_{i:02d}", synthetic_clip(frames, h, w, seed=100 + i), np.repeat(m[None], frames, 0)


def evaluate(args, out=print):
    import torch
    from propainter_amd import hip
    from propainter_amd.model.modules.flow_comp_raft import RAFT_bi, assert_finite_flows
    from propainter_amd.model.propainter import InpaintGenerator
    from propainter_amd.model.recurrent_flow_completion import RecurrentFlowCompleteNet
    from propainter_amd.pipeline import InferenceConfig, run_clip
    from propainter_amd.synthetic import seeded_models
    if not torch.cuda.is_available():
        raise SystemExit("evaluate_propainter.py runs on the HIP engine only: no GPU visible")
    device = torch.device("cuda:0")
    hip.lib()
    size = (args.width, args.height)
    have = all(os.path.isfile(p) for p in (args.raft_model_path, args.fc_model_path, args.propainter_model_path))
    prec = "f16x3" if args.fp16 else "f32"
    if have:
        fix_raft = RAFT_bi(args.raft_model_path, device, precision=prec)
        fix_flow_complete = RecurrentFlowCompleteNet(args.fc_model_path).to(device).eval()
        model = InpaintGenerator(model_path=args.propainter_model_path).to(device).eval()
        if args.fp16:
            fix_flow_complete, model = fix_flow_complete.half(), model.half()
        models = (fix_raft, fix_flow_complete, model)
    else:
        out("checkpoints not found (none ship offline): evaluating with the repo's seeded weights -- the numbers measure the engine, not a trained model")
        models = seeded_models(device, raft_precision=prec)
    if args.synthetic:
        videos = list(synthetic_dataset(args.synthetic, args.frames, size))
        dataset = "synthetic"
    else:
        assert args.dataset in ('davis', 'youtube-vos'), f"{args.dataset} dataset is not supported"
        dataset = args.dataset
        videos = ((v,) + load_video(args.video_root, args.mask_root, v, size) for v in sorted(os.listdir(args.mask_root)))
        videos = list(videos)
    result_path = os.path.join(args.result_root, f'{dataset}_rs_{args.ref_stride}_nl_{args.neighbor_length}_{args.task}')
    os.makedirs(result_path, exist_ok=True)
    completion = args.task == 'video_completion'
    eval_summary = open(os.path.join(result_path, f"{dataset}_metrics.txt"), "w") if completion else None
    cfg = InferenceConfig(raft_iter=args.raft_iter, subvideo_length=10 ** 6, neighbor_length=args.neighbor_length,
                          ref_stride=args.ref_stride, fp16=bool(args.fp16))
    out('Start evaluation ...')
    time_all, total_psnr, total_ssim = [], [], []
    avg_time = float('nan')
    for index, (video_name, frames_u8, masks) in enumerate(videos):
        out(f'Processing: {video_name}')
        L = len(frames_u8)
        torch.cuda.synchronize()
        t0 = time()
        gt = None
        if args.load_flow:        # (:113-114; the .flo pairs <frame>_<next>_f.flo / <next>_<frame>_b.flo of scripts/compute_flow.py)
            from propainter_amd import flow_io
            gt = flow_io.load_clip_flows(os.path.join(args.flow_root, video_name))
        comp = run_clip(models, frames_u8, masks, masks, cfg, device, float_blend=True, gt_flows=gt)
        torch.cuda.synchronize()
        time_all.append((time() - t0) / L)
        assert_finite_flows(models[0])
        comp = comp.cpu().numpy()
        avg_time = sum(time_all) / len(time_all)
        if completion:
            cur_psnr, cur_ssim = [], []
            for ori, c in zip(frames_u8, comp):
                p, s = calc_psnr_and_ssim(ori, c)
                cur_psnr.append(p)
                cur_ssim.append(s)
            total_psnr += cur_psnr
            total_ssim += cur_ssim
            line = (f'[{index + 1:3}/{len(videos)}] Name: {str(video_name):25} | PSNR/SSIM: {sum(cur_psnr) / L:.4f}/{sum(cur_ssim) / L:.4f} '
                    f'| Avg PSNR/SSIM: {sum(total_psnr) / len(total_psnr):.4f}/{sum(total_ssim) / len(total_ssim):.4f} | Time: {avg_time:.4f}')
            out(line)
            eval_summary.write(line + '\n')
        else:
            out(f'[{index + 1:3}/{len(videos)}] Name: {str(video_name):25} | Time: {avg_time:.4f}')
        if args.save_results:
            from PIL import Image
            d = os.path.join(result_path, video_name)
            os.makedirs(d, exist_ok=True)
            for i, f in enumerate(comp):
                Image.fromarray(f.astype(np.uint8)).save(os.path.join(d, str(i).zfill(5) + '.png'))
    if completion:
        line = ('Finish evaluation... Average Frame PSNR/SSIM/VFID: '
                f'{sum(total_psnr) / len(total_psnr):.2f}/{sum(total_ssim) / len(total_ssim):.4f}/{float("nan"):.3f} | Time: {avg_time:.4f}')
        out(line)
        out('(VFID needs the I3D checkpoint weights/i3d_rgb_imagenet.pt and network of core/metrics.py:57-571: not part of this path)')
        eval_summary.write(line)
        eval_summary.close()
        return dict(psnr=sum(total_psnr) / len(total_psnr), ssim=sum(total_ssim) / len(total_ssim), time=avg_time, path=result_path)
    out(f'Finish evaluation... Time: {avg_time:.4f}')
    return dict(time=avg_time, path=result_path)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--height', type=int, default=240)
    parser.add_argument('--width', type=int, default=432)
    parser.add_argument("--ref_stride", type=int, default=10)
    parser.add_argument("--neighbor_length", type=int, default=20)
    parser.add_argument("--raft_iter", type=int, default=20)
    parser.add_argument('--task', default='video_completion', choices=['object_removal', 'video_completion'])
    parser.add_argument('--raft_model_path', default='weights/raft-things.pth', type=str)
    parser.add_argument('--fc_model_path', default='weights/recurrent_flow_completion.pth', type=str)
    parser.add_argument('--propainter_model_path', default='weights/ProPainter.pth', type=str)
    parser.add_argument('--dataset', choices=['davis', 'youtube-vos'], type=str)
    parser.add_argument('--video_root', default='dataset_root', type=str)
    parser.add_argument('--mask_root', default='mask_root', type=str)
    parser.add_argument('--flow_root', default='flow_ground_truth_root', type=str)
    parser.add_argument('--load_flow', default=False, type=bool)
    parser.add_argument('--save_results', action='store_true')
    parser.add_argument('--num_workers', default=4, type=int)
    # engine extensions
    parser.add_argument('--fp16', action='store_true', help='stages B-D in fp16, RAFT at fp32-class precision (the CLI\'s --fp16 split)')
    parser.add_argument('--synthetic', type=int, default=0, help='evaluate this many seeded synthetic clips instead of a dataset')
    parser.add_argument('--frames', type=int, default=24, help='length of the synthetic clips')
    parser.add_argument('--result_root', default='results_eval', type=str)
    return parser


if __name__ == '__main__':
    evaluate(build_parser().parse_args())
