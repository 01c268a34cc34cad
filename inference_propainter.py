#!/usr/bin/env python
"""ProPainter inference on MI355X -- drop-in for the reference's ``inference_propainter.py`` command line
(flag names, defaults and the ``results/<video_name>/`` layout follow inference_propainter.py:181-217,233,453-472).

    python inference_propainter.py -i inputs/object_removal/bmx-trees -m inputs/object_removal/bmx-trees_mask --fp16
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 inference_propainter.py -i long_clip ...

The device path (RAFT, flow completion, image propagation, sliding-window generator, uint8 composite) runs on the
HIP engine (libpropainter_hip.so); with several processes (one per GPU) a long clip is sharded by sub-video
(propainter_amd/sharding.py).  Extra flags that the reference does not have: ``--weights_dir``, ``--seeded_weights``
(no checkpoints ship with either repository), ``--save_flow`` / ``--load_flow`` (RAFT output as the reference's ``.flo``
files; single-process runs), ``--raft_fp16`` / ``--raft_fp32`` (RAFT arithmetic; the default keeps
RAFT at the reference's precision class under ``--fp16``, see ``raft_precision`` below).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_parser():
    """Flags and defaults of the reference command line (inference_propainter.py:181-217); the help texts are ours."""
    p = argparse.ArgumentParser(description="ProPainter video inpainting / outpainting on MI355X (HIP engine)")
    a = p.add_argument
    a('-i', '--video', type=str, default='inputs/object_removal/bmx-trees', help='input clip: a video file or a folder of frames')
    a('-m', '--mask', type=str, default='inputs/object_removal/bmx-trees_mask', help='hole mask: one image for all frames, or a folder with one mask per frame')
    a('-o', '--output', type=str, default='results', help='where results/<clip name>/ is created')
    a("--resize_ratio", type=float, default=1.0, help='scale factor applied to the processing resolution')
    a('--height', type=int, default=-1, help='processing height in pixels (with --width; -1 keeps the input size)')
    a('--width', type=int, default=-1, help='processing width in pixels (with --height; -1 keeps the input size)')
    a('--mask_dilation', type=int, default=4, help='binary dilation iterations applied to the masks (frames and flows)')
    a("--ref_stride", type=int, default=10, help='distance between the global reference frames of a window')
    a("--neighbor_length", type=int, default=10, help='number of local neighbour frames per window')
    a("--subvideo_length", type=int, default=80, help='sub-video length used to chunk long clips (also the multi-GPU shard unit)')
    a("--raft_iter", type=int, default=20, help='RAFT refinement iterations')
    a('--mode', default='video_inpainting', choices=['video_inpainting', 'video_outpainting'], help="fill masked holes, or extend the field of view")
    a('--scale_h', type=float, default=1.0, help='outpainting: height factor of the new field of view')
    a('--scale_w', type=float, default=1.2, help='outpainting: width factor of the new field of view')
    a('--save_fps', type=int, default=24, help='frame rate of the written videos when the input has none')
    a('--save_frames', action='store_true', help='also write every output frame as PNG')
    a('--fp16', action='store_true', help='half-precision stages (fp16 storage, fp32 accumulation)')
    # ---- not in the reference
    a('--weights_dir', type=str, default='weights', help='folder holding raft-things.pth, recurrent_flow_completion.pth, ProPainter.pth')
    a('--seeded_weights', action='store_true', help='run with the deterministic seeded weights (no checkpoints available offline)')
    a('--save_flow', type=str, default=None, help="write the RAFT flows of the clip as .flo files (the reference's PIEH / float16 "
      'format, utils/flow_util.py) into this folder')
    a('--load_flow', type=str, default=None, help='read the RAFT flows from this folder of .flo files instead of running RAFT '
      '(as written by --save_flow or by the reference scripts/compute_flow.py naming: 00000_f.flo / 00000_b.flo ...)')
    a('--raft_fp16', action='store_true', help='opt in to fp16 RAFT activations/weights (fp32 accumulation, correlation, coordinates and flow): '
      'fastest, ~0.004 px mean end-point error against fp32 at 720p; only with --fp16')
    a('--raft_fp32', action='store_true', help='exact fp32 RAFT products on the fp32 matrix instructions (slowest)')
    return p


def raft_precision(args):
    """The reference keeps RAFT in fp32 even under --fp16 (inference_propainter.py:311,333-337).  So does this command
    line: by default RAFT tensors stay fp32 and the products run as three fp16 MFMAs with fp32 accumulation ("f16x3",
    ~2^-21 per product); --raft_fp32 selects the exact fp32 matrix instructions, --raft_fp16 (with --fp16) the fp16
    engine.  Without --fp16 everything is exact fp32."""
    if args.raft_fp32 or not args.fp16:
        return "f32"
    return "f16" if args.raft_fp16 else "f16x3"


def load_models(args, device):
    import torch
    from propainter_amd.model.modules.flow_comp_raft import RAFT_bi
    from propainter_amd.model.propainter import InpaintGenerator
    from propainter_amd.model.recurrent_flow_completion import RecurrentFlowCompleteNet
    prec = raft_precision(args)
    if args.seeded_weights:
        from propainter_amd.synthetic import seeded_models
        return seeded_models(device, raft_precision=prec)
    paths = {n: os.path.join(args.weights_dir, n) for n in ('raft-things.pth', 'recurrent_flow_completion.pth', 'ProPainter.pth')}
    missing = [p for p in paths.values() if not os.path.exists(p)]
    if missing:
        raise SystemExit(f"missing checkpoints {missing}: place the released .pth files in {args.weights_dir}/ "
                         "(the reference downloads them from its GitHub release) or pass --seeded_weights")
    fix_raft = RAFT_bi(paths['raft-things.pth'], device, precision=prec)
    fix_flow_complete = RecurrentFlowCompleteNet(paths['recurrent_flow_completion.pth'])
    for p in fix_flow_complete.parameters():
        p.requires_grad = False
    fix_flow_complete.to(device).eval()
    model = InpaintGenerator(model_path=paths['ProPainter.pth']).to(device).eval()
    if args.fp16:                                            # (:333-337)
        fix_flow_complete, model = fix_flow_complete.half(), model.half()
    return fix_raft, fix_flow_complete, model


def main(argv=None):
    args = build_parser().parse_args(argv)
    import numpy as np
    import torch
    from propainter_amd import hip, video_io
    from propainter_amd.pipeline import InferenceConfig, run_clip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("inference_propainter.py runs on the HIP engine only: no GPU visible (the CPU path is the oracle, "
                         "oracle/propainter_oracle.py, and is test infrastructure)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    hip.lib()
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    frames, fps, size, video_name = video_io.read_frames(args.video)
    if args.width != -1 and args.height != -1:
        size = (args.width, args.height)
    if args.resize_ratio != 1.0:
        size = (int(args.resize_ratio * size[0]), int(args.resize_ratio * size[1]))
    frames, size, out_size = video_io.resize_frames(frames, size)
    fps = args.save_fps if fps is None else fps
    save_root = os.path.join(args.output, video_name)
    if args.mode == 'video_inpainting':
        flow_masks, masks_dilated = video_io.read_masks(args.mask, len(frames), size, flow_mask_dilates=args.mask_dilation,
                                                        mask_dilates=args.mask_dilation, device=device)
    else:
        frames, flow_masks, masks_dilated, size = video_io.extrapolation(frames, (args.scale_h, args.scale_w))
    frames_u8 = np.stack([np.asarray(f, dtype=np.uint8) for f in frames])
    flow_masks = np.stack(flow_masks)
    masks_dilated = np.stack(masks_dilated)
    L = len(frames_u8)

    models = load_models(args, device)
    cfg = InferenceConfig(raft_iter=args.raft_iter, subvideo_length=args.subvideo_length, neighbor_length=args.neighbor_length,
                          ref_stride=args.ref_stride, fp16=bool(args.fp16))
    if rank == 0:
        print(f'\nProcessing: {video_name} [{L} frames]...  (RAFT precision {models[0].precision}, stages {"fp16" if args.fp16 else "fp32"})')
    t0 = time.perf_counter()
    state = {"sharded": False}
    failed = True
    try:
        _run(args, models, cfg, frames_u8, flow_masks, masks_dilated, L, world, rank, device, t0, save_root, out_size, fps, video_name, state)
        failed = False
    finally:
        if world > 1:
            import torch.distributed as dist
            # Idle ranks (nothing to shard: rank 0 works alone) and finished ranks meet at a barrier before the group is torn down.  It
            # sits in a ``finally`` so that rank 0 failing in an UNSHARDED pass still releases the idle ranks instead of leaving them to
            # the watchdog.  A rank that fails inside a SHARDED pass must not wait: its peers are blocked in point-to-point exchanges
            # with it, not at the barrier -- it leaves at once, which closes their connections and fails them promptly.
            if not (failed and state["sharded"]):
                try:
                    dist.barrier()
                except Exception as e:      # a peer died: nothing left to wait for
                    print(f'[rank {rank}] barrier before shutdown failed: {type(e).__name__}: {e}', file=sys.stderr)
            try:
                dist.destroy_process_group()
            except Exception:
                pass


def _run(args, models, cfg, frames_u8, flow_masks, masks_dilated, L, world, rank, device, t0, save_root, out_size, fps, video_name, state):
    import torch
    from propainter_amd import video_io
    from propainter_amd.model.modules.flow_comp_raft import assert_finite_flows
    from propainter_amd.pipeline import run_clip
    comp = None
    if world > 1:
        from propainter_amd.sharding import can_shard, gather_frames, run_clip_sharded
        if args.save_flow or args.load_flow:
            raise SystemExit('--save_flow / --load_flow describe ONE unsharded clip: run them without torch.distributed.run '
                             '(under sharding every rank computes the RAFT flows of its own sub-videos)')
        if can_shard(L, cfg, world):
            state["sharded"] = True
            lo, part = run_clip_sharded(models, frames_u8, flow_masks, masks_dilated, cfg, device)
            comp = gather_frames(lo, part, L, dst=0)
        elif rank == 0:      # nothing to shard: rank 0 runs the unsharded pass, the others idle
            why = (f'--subvideo_length {cfg.subvideo_length} > 100 (image propagation then runs in chunks of 100: inference_propainter.py:373 of the reference)'
                   if cfg.subvideo_length > 100 else f'{L} frames are a single sub-video at --subvideo_length {cfg.subvideo_length}')
            print(f'not sharding over {world} GPUs: {why}; running on one GPU')
            comp = run_clip(models, frames_u8, flow_masks, masks_dilated, cfg, device)
    else:
        from propainter_amd import flow_io
        gt = None
        if args.load_flow:
            gt = flow_io.load_clip_flows(args.load_flow)
            print(f'RAFT skipped: {gt[0].shape[0]} flow pairs read from {args.load_flow}')
        if args.save_flow:
            comp, stages = run_clip(models, frames_u8, flow_masks, masks_dilated, cfg, device, return_stages=True, gt_flows=gt)
            flow_io.save_clip_flows(stages["gt_flows"][0][0], stages["gt_flows"][1][0], args.save_flow)
        else:
            comp = run_clip(models, frames_u8, flow_masks, masks_dilated, cfg, device, gt_flows=gt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert_finite_flows(models[0])        # split-plane RAFT: a value beyond fp16's range would have produced NaN flows -- fail loudly
    if rank == 0:
        # final resize of the video frames (cv2.resize(f, out_size), :469-470) on the device, before the single device->host copy
        from propainter_amd import hip
        resized = None
        if comp.is_cuda and tuple(out_size) != (comp.shape[2], comp.shape[1]):
            resized = list(hip.resize_bilinear_u8(comp.contiguous(), out_size).cpu().numpy())
        comp = comp.cpu().numpy()
        print(f'{L} frames in {dt:.2f} s ({L / dt:.2f} frames/s on {world} GPU(s))')
        video_io.save_results(save_root, list(comp), video_io.masked_preview(frames_u8, masks_dilated), out_size, fps,
                              args.save_frames, comp_video_frames=resized)
        print(f'\nAll results are saved in {save_root}')


if __name__ == '__main__':
    main()
