#!/usr/bin/env python
"""One LONG clip on one GPU: the whole-pass hipGraph (pipeline.ClipGraph: stages A-D one after the other over the whole clip, the
reference's order) against the streaming schedule (sharding.StreamingClipGraph: the sub-videos as a pipeline, RAFT of sub-video k + 3
next to the generator windows of sub-video k).  BASELINE config 4 by default (720x1280, 320 frames, sub-videos of 80).

    python tools/bench_streaming.py [--frames 320 --height 720 --width 1280 --subvideo_length 80 --steps 2] > gpurun_out/streaming.json

Prints one JSON object: ms per clip and frames/s of both schedules (hipGraph replays, inputs resident, barrier-free single process),
the lockstep replay of the same segment graphs (the A/B that isolates the ORDER), peak memory, and that the streaming output equals
the whole-pass output byte for byte."""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.ndimage
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from propainter_amd import hip  # noqa: E402
from propainter_amd.pipeline import ClipGraph, InferenceConfig  # noqa: E402
from propainter_amd.sharding import StreamingClipGraph  # noqa: E402
from propainter_amd.synthetic import seeded_models, synthetic_clip, synthetic_mask  # noqa: E402


def timed(fn, steps, dev):
    fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * 1e3 / steps, out


def diff_stats(a, b):
    """byte-level difference of two uint8 clips: equal?, fraction of differing bytes, max |d|, frames that differ"""
    ne = a != b
    if not bool(ne.any()):
        return {"equal": True}
    d = (a.to(torch.int16) - b.to(torch.int16)).abs()
    fr = [i for i in range(a.shape[0]) if bool(ne[i].any())]
    return {"equal": False, "bytes_differ_frac": float(ne.float().mean()), "max_abs": int(d.max()), "frames_differing": len(fr), "first_frames": fr[:6]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=320)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--subvideo_length", type=int, default=80)
    ap.add_argument("--neighbor_length", type=int, default=10)
    ap.add_argument("--ref_stride", type=int, default=10)
    ap.add_argument("--raft_iter", type=int, default=20)
    ap.add_argument("--raft-dtype", default="f16x3")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--volume-gb", type=float, default=40.0)
    ap.add_argument("--skip-whole-pass", action="store_true")
    ap.add_argument("--only-pipelined", action="store_true", help="stop after the pipelined single-graph leg")
    ap.add_argument("--debug-stages", action="store_true", help="compare every logical rank's stage outputs inside the pipelined graph with the eager pass (PP_SG_DEBUG)")
    ap.add_argument("--private-pools", action="store_true", help="also run the schedule with one private graph pool per logical rank + its lockstep A/B")
    args = ap.parse_args()
    hip.lib()
    dev = torch.device("cuda", 0)
    L, H, W = args.frames, args.height, args.width
    models = seeded_models(dev, raft_precision=args.raft_dtype)
    cfg = InferenceConfig(raft_iter=args.raft_iter, subvideo_length=args.subvideo_length, neighbor_length=args.neighbor_length,
                          ref_stride=args.ref_stride, fp16=True, window_streams=int(os.environ.get("PP_DIAG_LANES", "2")),
                          batch_propagation=os.environ.get("PP_DIAG_NOBATCH") != "1")
    m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.uint8) * 255
    clip, masks = synthetic_clip(L, H, W, seed=2023), np.repeat(m[None], L, 0)
    rec = {"workload": f"{H}x{W}x{L} frames, subvideo_length {args.subvideo_length}, fp16 stages + RAFT {args.raft_dtype}, one MI355X",
           "steps": args.steps}

    # ---- round 5 default: the wavefront as ONE hipGraph pipelined by stage (the overlapped form)
    if args.debug_stages:
        os.environ["PP_SG_DEBUG"] = "1"
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    sp1 = StreamingClipGraph(models, L, H, W, cfg, dev, volume_gb=args.volume_gb, single_graph=True)
    sp1.load(clip, masks, masks)
    sp1.capture()
    rec["pipelined_capture_s"] = time.perf_counter() - t0
    ms_pipe, out_pipe = timed(sp1.replay, args.steps, dev)
    out_pipe = out_pipe.clone()
    rec["pipelined_single_graph"] = {"ms_per_clip": ms_pipe, "frames_per_s": L / ms_pipe * 1e3}
    if args.debug_stages:
        from propainter_amd.pipeline import run_clip
        from propainter_amd.sharding import ShardPlan
        eager_out, st = run_clip(models, clip, masks, masks, cfg, dev, return_stages=True)
        torch.cuda.synchronize(dev)
        plan = ShardPlan(L, sp1.cfg, sp1.world)
        dbg = []
        mx = lambda a, b: float((a.float() - b.float()).abs().max()) if a.numel() else 0.0
        for r in range(sp1.world):
            lo, hi = plan.flows_own(r)
            flo, fhi = plan.own[r]
            d = sp1._single_debug
            row = {"rank": r}
            row["raft_flows"] = mx(d[(r, 0)]["gt"], torch.stack([st["gt_flows"][0][:, lo:hi], st["gt_flows"][1][:, lo:hi]], 0))
            row["completed_flows"] = mx(d[(r, 1)]["pred_own"], torch.stack([st["pred_flows"][0][:, lo:hi], st["pred_flows"][1][:, lo:hi]], 0))
            row["updated_frames_masks"] = mx(d[(r, 2)]["upd_own"], torch.cat([st["updated_frames"][:, flo:fhi], st["updated_masks"][:, flo:fhi]], 2))
            row["composited_bytes_differ"] = float((out_pipe[flo:fhi] != eager_out[flo:fhi]).float().mean())
            dbg.append(row)
        rec["pipelined_stage_deviation_vs_eager"] = dbg
        again = sp1.replay().clone()
        torch.cuda.synchronize(dev)
        rec["pipelined_replay_to_replay"] = diff_stats(again, out_pipe)
        rec["pipelined_vs_eager"] = diff_stats(out_pipe, eager_out)
        print(json.dumps(dbg), file=sys.stderr, flush=True)
    if args.only_pipelined:
        print(json.dumps(rec))
        return
    rec["pipelined_peak_reserved_GB"] = torch.cuda.max_memory_reserved(dev) / 1e9
    torch.cuda.empty_cache()
    rec["pipelined_reserved_steady_GB"] = torch.cuda.memory_reserved(dev) / 1e9
    print(json.dumps(rec), file=sys.stderr, flush=True)
    del sp1
    torch.cuda.empty_cache()

    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    sc = StreamingClipGraph(models, L, H, W, cfg, dev, volume_gb=args.volume_gb, single_graph=False)   # one graph per (rank, segment), ONE pool, chained launches
    sc.load(clip, masks, masks)
    sc.capture()
    rec["streaming_capture_s"] = time.perf_counter() - t0
    rec["logical_ranks"] = sc.world
    rec["issue_order_head"] = sc.order[:12]
    ms_stream, out_stream = timed(sc.replay, args.steps, dev)
    out_stream = out_stream.clone()
    rec["streaming"] = {"ms_per_clip": ms_stream, "frames_per_s": L / ms_stream * 1e3}
    rec["pipelined_equals_chained"] = bool(torch.equal(out_pipe, out_stream))
    rec["pipelined_vs_chained"] = diff_stats(out_pipe, out_stream)
    rec["streaming_peak_reserved_GB"] = torch.cuda.max_memory_reserved(dev) / 1e9        # incl. the eager warm-up pass of every logical rank
    torch.cuda.empty_cache()
    rec["streaming_reserved_steady_GB"] = torch.cuda.memory_reserved(dev) / 1e9           # what a serving loop holds: the graph pool + static buffers
    print(json.dumps(rec), file=sys.stderr, flush=True)
    del sc
    torch.cuda.empty_cache()

    if args.private_pools:      # the same segment graphs with one private pool per logical rank: the lockstep A/B order needs them
        torch.cuda.reset_peak_memory_stats(dev)
        sp = StreamingClipGraph(models, L, H, W, cfg, dev, volume_gb=args.volume_gb, share_pool=False, single_graph=False)
        sp.load(clip, masks, masks)
        sp.capture()
        ms_priv, out_priv = timed(sp.replay, args.steps, dev)
        ms_lock, out_lock = timed(lambda: sp.replay(lockstep=True), args.steps, dev)
        rec["private_pools"] = {"ms_per_clip": ms_priv, "frames_per_s": L / ms_priv * 1e3, "peak_reserved_GB": torch.cuda.max_memory_reserved(dev) / 1e9,
                                "equals_shared_pool": bool(torch.equal(out_priv, out_stream))}
        rec["same_graphs_lockstep"] = {"ms_per_clip": ms_lock, "frames_per_s": L / ms_lock * 1e3}
        rec["streaming_equals_lockstep"] = bool(torch.equal(out_priv, out_lock))
        print(json.dumps(rec), file=sys.stderr, flush=True)
        del sp, out_priv, out_lock
        torch.cuda.empty_cache()

    if not args.skip_whole_pass:
        torch.cuda.reset_peak_memory_stats(dev)
        t0 = time.perf_counter()
        g = ClipGraph(models, L, H, W, cfg, dev, example=(clip, masks, masks), release_eager_pool=True)
        rec["whole_pass_capture_s"] = time.perf_counter() - t0
        ms_whole, out_whole = timed(g.replay, args.steps, dev)
        rec["whole_pass"] = {"ms_per_clip": ms_whole, "frames_per_s": L / ms_whole * 1e3}
        rec["whole_pass_peak_reserved_GB"] = torch.cuda.max_memory_reserved(dev) / 1e9
        torch.cuda.empty_cache()
        rec["whole_pass_reserved_steady_GB"] = torch.cuda.memory_reserved(dev) / 1e9
        rec["streaming_equals_whole_pass"] = bool(torch.equal(out_stream, out_whole))
        rec["pipelined_equals_whole_pass"] = bool(torch.equal(out_pipe, out_whole))
        rec["pipelined_vs_whole_pass"] = diff_stats(out_pipe, out_whole)
        rec["chained_vs_whole_pass"] = diff_stats(out_stream, out_whole)
        from propainter_amd.pipeline import run_clip
        eager = run_clip(models, clip, masks, masks, cfg, dev)
        torch.cuda.synchronize(dev)
        rec["eager_run_clip_vs_whole_pass"] = diff_stats(eager, out_whole)
        rec["eager_run_clip_vs_pipelined"] = diff_stats(eager, out_pipe)
        rec["eager_run_clip_vs_chained"] = diff_stats(eager, out_stream)
        rec["speedup_streaming_vs_whole_pass"] = ms_whole / ms_stream
        rec["speedup_pipelined_vs_whole_pass"] = ms_whole / ms_pipe
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
