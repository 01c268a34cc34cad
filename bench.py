#!/usr/bin/env python
"""Headline benchmark: inpainted frames/s of the whole ProPainter inference path on MI355X.

A *step* = one pass of the hot path (RAFT -> flow completion -> image propagation -> sliding-window generator ->
uint8 composite/blend -> one device->host copy of the result) over one synthetic clip whose frames and masks are
already resident in HBM when the timed region starts (the reference's own protocol,
scripts/evaluate_propainter.py:100-101,181-184: decode, mask dilation and model load are excluded).

    python bench.py --gpus 1 --steps 2 --warmup 1                      # 720x1280, 80 frames, fp16 (BASELINE C3)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                         # N independent clips, one per GPU (weak scaling)

Rank 0 prints ONE JSON line (schema: see README / the driver contract) with two extra objects:
  "roofline":     the dominant kernel class of the step measured live with HIP events on the launch stream
                  (KernelProfiler in propainter_amd/hip.py) in one extra instrumented step after the timed region;
  "cpu_baseline": the CPU oracle (oracle/propainter_oracle.py — the checker, never the product) timed on this
                  box's host cores on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3}     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
# KernelProfiler class -> kernel family of tools/rocprof_summary.py (profiles/*_hbm_traffic.json, PMC passes)
TRAFFIC_FAMILY = {"conv_gemm_f16": "conv_gemm_f16 (LDS-DMA implicit GEMM)", "conv_gemm_f32": "conv_gemm_f32",
                  "conv_gemm_dcn": "conv_gemm_f16/dcn (register-staged)", "sparse_window_attention": "sparse_window_attention",
                  "fold_tokens": "fold_tokens", "corr_lookup": "corr_lookup"}


def pmc_traffic(kernel_class):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 --pmc summary (FETCH_SIZE / WRITE_SIZE in
    separate passes, gfx950 correction applied by tools/rocprof_summary.py); None when no summary is present."""
    import glob
    fam = TRAFFIC_FAMILY.get(kernel_class)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic*.json")), reverse=True):
        try:
            rec = json.load(open(f))["families"].get(fam)
        except Exception:
            continue
        if rec:
            return rec["hbm_bytes_per_launch"]
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--neighbor_length", type=int, default=10)
    ap.add_argument("--ref_stride", type=int, default=10)
    ap.add_argument("--subvideo_length", type=int, default=80)
    ap.add_argument("--raft_iter", type=int, default=20)
    ap.add_argument("--fp32", action="store_true", help="run stages B-D in fp32 instead of fp16")
    ap.add_argument("--raft-dtype", default="f16", choices=["f32", "f16"],
                    help="RAFT engine dtype: f16 (default) = fp16 activations/weights on MFMA with fp32 accumulation, fp32 "
                         "correlation volume, coordinates and flow (SURVEY.md section 7; measured EPE vs the fp32 reference "
                         "in tests/test_modules_gpu.py); f32 = exact fp32 MFMA like the reference, which keeps RAFT fp32 "
                         "under --fp16 (inference_propainter.py:311)")
    ap.add_argument("--eager", action="store_true",
                    help="issue every launch from Python each step instead of replaying the captured hipGraph of the pass "
                         "(pipeline.ClipGraph); the kernels and their order are identical, only the submission differs")
    ap.add_argument("--detail", action="store_true", help="split the per-kernel table by convolution layer shape (diagnostic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the instrumented (per-kernel HIP events) step")
    ap.add_argument("--cpu-sample-frames", type=int, default=16, help="frames of the 432x240 CPU-oracle sample (about 10-15 s on 16 threads)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def _cpu_baseline_worker(args):
    """Runs in a child process (bounded by a timeout in the parent): the CPU oracle, fp32, on the host cores."""
    import numpy as np
    import scipy.ndimage
    import torch
    from oracle import propainter_oracle as O
    from propainter_amd.synthetic import seeded_models, synthetic_clip, synthetic_mask
    cores = int(os.environ.get("PP_CPU_THREADS", "1"))
    torch.set_num_threads(cores)
    H, W, L = 240, 432, args.cpu_sample_frames
    clip = synthetic_clip(L, H, W)
    m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.uint8) * 255
    masks = np.repeat(m[None], L, 0)
    raft, fc, gen = seeded_models("cpu")
    sds = {"raft": {k: v.float() for k, v in raft.fix_raft.state_dict().items()},
           "fc": {k: v.float() for k, v in fc.state_dict().items()},
           "gen": {k: v.float() for k, v in gen.state_dict().items()}}
    t0 = time.perf_counter()
    with torch.no_grad():
        O.inpaint_video(sds, clip, masks, masks, raft_iter=args.raft_iter, subvideo_length=args.subvideo_length,
                        neighbor_length=args.neighbor_length, ref_stride=args.ref_stride)
    dt = time.perf_counter() - t0
    print(json.dumps({"seconds": dt, "frames": L, "cores": cores, "H": H, "W": W}))


def cpu_baseline(args, timeout=240):
    """CPU oracle (fp32) on a bounded sample: the full path over a short 432x240 clip in a child process with a hard
    timeout, scaled to the bench resolution by the algorithmic FLOPs per frame."""
    import subprocess
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))          # the oracle's small ops stop scaling (and oversubscribe) beyond ~16 threads
    env = dict(os.environ, PP_CPU_THREADS=str(cores), OMP_NUM_THREADS=str(cores), MKL_NUM_THREADS=str(cores),
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--cpu-sample-frames", str(args.cpu_sample_frames),
           "--raft_iter", str(args.raft_iter), "--subvideo_length", str(args.subvideo_length),
           "--neighbor_length", str(args.neighbor_length), "--ref_stride", str(args.ref_stride)]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        rec = json.loads(r.stdout.decode().strip().splitlines()[-1])
    except Exception as e:  # timeout / crash: report it instead of blocking the bench
        return {"value": None, "unit": "frames/s", "cores": cores, "kind": "port",
                "sample": f"CPU oracle sample did not finish within {timeout} s on {cores} threads ({type(e).__name__})"}
    dt, L, H, W = rec["seconds"], rec["frames"], rec["H"], rec["W"]
    fps_sample = L / dt
    # per-frame algorithmic work: 7.49 TFLOP at 720x1280 vs 0.80 TFLOP at 240x432, ~ linear in pixels
    scale = (args.height * args.width) / float(720 * 1280) * (7.49 / 0.80)
    return {"value": fps_sample / scale, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle fp32, full path on a {L}-frame {W}x{H} synthetic clip: {dt:.1f} s = {fps_sample:.4f} frames/s "
                      f"on {cores} threads ({avail} available); scaled to {args.width}x{args.height} by algorithmic "
                      f"FLOPs/frame (/{scale:.2f})",
            "measured_sample_seconds": dt}


def main():
    args = parse()
    if args.cpu_baseline_worker:
        _cpu_baseline_worker(args)
        return
    import numpy as np
    import scipy.ndimage
    import torch
    import torch.distributed as dist
    from propainter_amd import hip
    from propainter_amd.pipeline import InferenceConfig, run_clip, window_schedule
    from propainter_amd.synthetic import seeded_models, synthetic_clip, synthetic_mask

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    hip.lib()

    H, W, L = args.height, args.width, args.frames
    fp16 = not args.fp32
    raft_dt = torch.float16 if args.raft_dtype == "f16" else None
    models = seeded_models(dev, raft_dtype=raft_dt)
    # every rank inpaints its own clip (sub-video sharding of a long video = independent windows per GPU; weak scaling)
    clip = synthetic_clip(L, H, W, seed=2023 + rank)
    m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.uint8) * 255
    frames_dev = torch.from_numpy(clip).to(dev)
    masks_dev = torch.from_numpy(np.repeat(m[None], L, 0)).to(dev)
    host_out = torch.empty((L, H, W, 3), dtype=torch.uint8).pin_memory()
    cfg = InferenceConfig(raft_iter=args.raft_iter, subvideo_length=args.subvideo_length,
                          neighbor_length=args.neighbor_length, ref_stride=args.ref_stride, fp16=fp16)

    def eager_step(stage_hook=None):
        comp = run_clip(models, frames_dev, masks_dev, masks_dev, cfg, dev, stage_hook=stage_hook)
        host_out.copy_(comp, non_blocking=True)

    # ---- setup (untimed, like model load in the reference protocol): one eager pass builds the engines (weight
    # packing, K tables, window tables) and primes the allocator; by default the pass is then captured in a hipGraph
    eager_step()
    eager_step()                      # the allocator settles on the second pass (its blocks are carved during the first)
    torch.cuda.synchronize()
    t_e = time.perf_counter()
    eager_step()
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t_e) * 1e3
    graph = None
    if not args.eager:
        from propainter_amd.pipeline import ClipGraph
        t_c = time.perf_counter()
        try:
            graph = ClipGraph(models, L, H, W, cfg, dev, example=(frames_dev, masks_dev, masks_dev))
            torch.cuda.synchronize()
        except Exception as e:       # capture refused (driver / runtime state): measure the eager submission instead of dying
            sys.stderr.write(f"[bench] hipGraph capture failed on rank {rank} ({type(e).__name__}: {e}); falling back to eager launches\n")
            graph = None
            torch.cuda.synchronize()
        capture_s = time.perf_counter() - t_c

    def step(stage_hook=None):
        if graph is None or stage_hook is not None:
            return eager_step(stage_hook)
        host_out.copy_(graph.replay(), non_blocking=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # per-step markers (diagnostic only, no synchronisation inside the timed region): a HIP event after each step's
    # last launch and the host clock when the step has been fully *submitted*
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_submit = []
    ev[0].record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
        host_submit.append((time.perf_counter() - t0) * 1e3)
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    host_submit = [host_submit[0]] + [host_submit[i] - host_submit[i - 1] for i in range(1, len(host_submit))]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    fps = world * L * args.steps / elapsed

    # ---- one instrumented step: stage split + per-kernel-class HIP-event timing (not part of `value`)
    stages, kernels, roof = None, None, None
    if rank == 0 and not args.no_profile:
        marks = []

        def hook(name):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e, time.perf_counter()))
        with hip.KernelProfiler(detail=args.detail) as kp:
            t1 = time.perf_counter()
            step(hook)
            torch.cuda.synchronize()
            prof_wall = time.perf_counter() - t1
        stages = {marks[i][0]: marks[i - 1][1].elapsed_time(marks[i][1]) for i in range(1, len(marks))}
        stages["host_submit_ms"] = {marks[i][0]: (marks[i][2] - marks[i - 1][2]) * 1e3 for i in range(1, len(marks))}
        kernels = kp.summary()
        for k, v in kernels.items():
            v["avg_us"] = v["ms"] * 1e3 / max(1, v["launches"])
            v["tflops"] = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
            v["gbs"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0
        dom = max(kernels.items(), key=lambda kv: kv[1]["ms"])
        name, v = dom
        if name.startswith("conv_gemm") or name == "sparse_window_attention":
            peak = PEAK_TFLOPS["f32" if name.endswith("f32") else "f16"]
            roof = {"kernel": name, "bound": "mfma", "achieved": v["tflops"], "peak": peak, "unit": "TFLOP/s",
                    "frac": v["tflops"] / peak, "traffic": pmc_traffic(name), "launches": v["launches"], "avg_launch_us": v["avg_us"],
                    "algorithmic_flop_per_launch": v["flops"] / max(1, v["launches"]),
                    "algorithmic_bytes_per_launch": v["bytes"] / max(1, v["launches"]),
                    "share_of_kernel_time": v["ms"] / sum(x["ms"] for x in kernels.values())}
        else:
            roof = {"kernel": name, "bound": "hbm", "achieved": v["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": v["gbs"] / PEAK_HBM_GBS, "traffic": pmc_traffic(name), "launches": v["launches"], "avg_launch_us": v["avg_us"],
                    "algorithmic_bytes_per_launch": v["bytes"] / max(1, v["launches"]),
                    "share_of_kernel_time": v["ms"] / sum(x["ms"] for x in kernels.values())}
        stages["instrumented_step_wall_ms"] = prof_wall * 1e3

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        sched = window_schedule(L, args.neighbor_length, args.ref_stride, args.subvideo_length)
        out = {
            "metric": "inpainted frames/sec (whole path, 80-frame window)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if fp16 else "f32",
            "data": "synthetic (seeded clip + rectangular mask dilated x4, seeded weights of the reference architecture)",
            "config": {"workload": f"{H}x{W} {L}-frame clip, neighbor_length={args.neighbor_length} ref_stride={args.ref_stride} "
                                   f"subvideo_length={args.subvideo_length} raft_iter={args.raft_iter}, one clip per GPU",
                       "height": H, "width": W, "frames": L, "windows": len(sched), "raft_dtype": args.raft_dtype,
                       "stages_dtype": "f16" if fp16 else "f32", "parallelism": f"clip-sharded x{world}"},
            "roofline": roof, "cpu_baseline": cpu, "stages_ms": stages,
            "submission": "eager (Python launches)" if graph is None else "hipGraph replay of the whole pass (pipeline.ClipGraph)",
            "eager_ms_per_step": eager_ms, "graph_capture_s": None if graph is None else capture_s,
            "step_ms": step_ms, "host_submit_ms": host_submit, "kernels": kernels,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
