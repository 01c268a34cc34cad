#!/usr/bin/env python
"""Headline benchmark: inpainted frames/s of the whole ProPainter inference path on MI355X.

A *step* = one pass of the hot path (RAFT -> flow completion -> image propagation -> sliding-window generator ->
uint8 composite/blend -> one device->host copy of the result) over one synthetic clip whose frames and masks are
already resident in HBM when the timed region starts (the reference's own protocol,
scripts/evaluate_propainter.py:100-101,181-184: decode, mask dilation and model load are excluded).

    python bench.py --gpus 1 --steps 2 --warmup 1                      # 720x1280, 80 frames, fp16 (BASELINE C3)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                         # N independent clips, one per GPU (weak scaling)
    python bench.py --gpus N ...                                       # the same: without WORLD_SIZE in the environment bench.py
                                                                       # re-executes itself under torch.distributed.run with N ranks
                                                                       # (the reference's own launch is mp.spawn(nprocs=world_size), train.py:105)
    ... bench.py --gpus N --sharded --frames 320                        # BASELINE C4: ONE clip, sub-video shards over N GPUs
    ... bench.py --gpus N --sharded --height 1080 --width 1920 --frames 160 --subvideo_length 20     # BASELINE C5

Rank 0 prints ONE JSON line (schema: see README / the driver contract) with these extra objects:
  "roofline":        the dominant kernel class of the step measured live with HIP events on the launch stream
                     (KernelProfiler in propainter_amd/hip.py) in one extra instrumented step after the timed region;
  "cpu_baseline":    the CPU oracle (oracle/propainter_oracle.py -- the checker, never the product) timed on this
                     box's host cores on a bounded sample AT THE TIMED RESOLUTION (a 6-frame clip through the whole path),
                     stage by stage; value = the clip's unit counts x the measured per-unit costs, rank 0, N=1;
  "parity":          the HIP path at the TIMED precision configuration and resolution on the very clip the CPU oracle sample
                     inpaints, compared byte for byte with the oracle's frames (PSNR, max |d|, fraction of differing bytes) and
                     |PSNR(HIP, ground truth) - PSNR(oracle, ground truth)| (north_star: within 0.05 dB);
  "parity_timed_output": the bytes the LAST TIMED step left on the host vs the committed fp32-oracle golden of that very 80-frame clip;
  "replay_consistency": --replay-checks more replays of the captured pass, each compared byte for byte with the first (a replay that
                     differs is a defect of the submission, however rare: profiles/r6_replay_bytes.txt);
  "raft_precisions": frames/s and parity of the same pass at the other RAFT precisions ("f16": fp16 activations, NARROWER
                     than the reference's fp32 RAFT; "f32": exact fp32 MFMA) next to the headline's;
  "memory":          peak device memory of the pass (the reference publishes only memory: README.md:192-195).
"""
import argparse
import dataclasses
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3,     # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
               # split-plane ("f16x3") layers: every fp32-class product is three fp16 MFMA products, so the roofline of the
               # ALGORITHMIC (fp32-layer) FLOPs on this hardware is the dense fp16 peak / 3 (identical fraction on executed FLOPs)
               "f16x3": 2500.0 / 3.0}
PEAK_HBM_GBS = 8000.0
# KernelProfiler class -> kernel family of tools/rocprof_summary.py (profiles/*_hbm_traffic.json, PMC passes)
TRAFFIC_FAMILY = {"conv_gemm_f16": "conv_gemm_f16 (LDS-DMA implicit GEMM)", "conv_gemm_f32": "conv_gemm_f32",
                  "conv_gemm_f16x3": "conv_gemm_f16x3 (split-plane LDS-DMA implicit GEMM)",
                  "conv_gemm_dcn": "conv_gemm_dcn (patch-staged)", "corr_lookup_otf": "corr_lookup_otf", "corr_lookup_otf_split": "corr_lookup_otf_split", "sparse_window_attention": "sparse_window_attention",
                  "fold_tokens": "fold_tokens", "corr_lookup": "corr_lookup"}
# minimal algorithmic FLOPs of BASELINE config 3 (720x1280x80, 25 % of the windows masked)
C3_ALGORITHMIC_TFLOP = 599.0


def pmc_traffic(kernel_class, raft_dtype):
    """HBM bytes per launch of a kernel class from a committed rocprofv3 --pmc summary (FETCH_SIZE / WRITE_SIZE in separate
    passes, gfx950 correction applied by tools/rocprof_summary.py).  PMC counters cannot be collected inside this process, so
    the figure is a committed profile's, never this run's: only a profile that records the commit it was measured at and the
    same RAFT precision is used, and both are reported.  Returns (bytes or None, source description or None)."""
    import glob
    from propainter_amd import build as _build
    fam = TRAFFIC_FAMILY.get(kernel_class)
    best = None
    digest = _build.source_digest()
    for f in glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic*.json")):
        try:
            d = json.load(open(f))
            rec = d["families"].get(fam)
        except Exception:
            continue
        # only a profile of THESE kernel sources (SHA-256 over csrc/ + include/ + the build flags, recorded by the profiling run): a
        # figure measured on other kernels is not this run's traffic (round 4 carried a profile that was several kernel changes old)
        if d.get("csrc_digest") != digest:
            continue
        if rec and d.get("commit") and d.get("raft_dtype") == raft_dtype and (best is None or d.get("commit_time", 0) > best[0]):
            best = (d.get("commit_time", 0), rec["hbm_bytes_per_launch"],
                    f"{os.path.relpath(f, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at commit {d['commit']}, RAFT {raft_dtype}, the same "
                    f"kernel sources as this run: csrc digest {digest[:12]}; not collected in this run)")
    return (best[1], best[2]) if best else (None, f"no committed --pmc profile of these kernel sources (csrc digest {digest[:12]}): traffic not reported")


def parse(argv=None):
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
    ap.add_argument("--raft-dtype", default="f16x3", choices=["f32", "f16x3", "f16"],
                    help="RAFT precision of the TIMED pass.  The reference keeps RAFT fp32 under --fp16 "
                         "(inference_propainter.py:311), so the like-for-like default is f16x3 = fp32-class RAFT: split-plane fp16 "
                         "activations / weights (22 significand bits), every product as three fp16 MFMAs (hi*hi + lo*hi + hi*lo, "
                         "~2^-21) with fp32 accumulation.  f32 = exact fp32 MFMA; f16 = fp16 activations / weights (NARROWER than "
                         "the reference: reported only as value_raft_f16 / under raft_precisions)")
    ap.add_argument("--sharded", action="store_true",
                    help="ONE clip of --frames frames sharded by sub-video over the ranks (propainter_amd/sharding.py, RCCL "
                         "point-to-point halo exchange; BASELINE configs 4 / 5) instead of one clip per rank; strong scaling")
    ap.add_argument("--eager", action="store_true",
                    help="issue every launch from Python each step instead of replaying the captured hipGraph of the pass "
                         "(pipeline.ClipGraph); the kernels and their order are identical, only the submission differs")
    ap.add_argument("--detail", action="store_true", help="split the per-kernel table by convolution layer shape (diagnostic)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU-oracle sample (and with it the parity block)")
    ap.add_argument("--no-profile", action="store_true", help="skip the instrumented (per-kernel HIP events) step")
    ap.add_argument("--no-precisions", action="store_true", help="skip the extra timed steps at the other RAFT precisions")
    ap.add_argument("--raft-streams", type=int, default=2,
                    help="RAFT encoders / pair-direction groups on this many HIP streams (InferenceConfig.raft_streams; identical flows)")
    ap.add_argument("--window-streams", type=int, default=2,
                    help="generator windows in flight on separate HIP streams (pipeline.InferenceConfig.window_streams)")
    ap.add_argument("--graph-lanes", action="store_true",
                    help="keep the window / RAFT lanes INSIDE the captured hipGraph (pipeline.ClipGraph(forked_branches=True)); default: the "
                         "captured pass is one chain of launches -- with forked branches config 5 left wrong bytes in 3 of 64 replays "
                         "(profiles/r6_c5_replays.txt); the lanes still serve eager submission")
    ap.add_argument("--single-pass", action="store_true",
                    help="profiling aid: run exactly ONE eager pass of the clip and exit (what the rocprofv3 passes of "
                         "tools/gpu_profile.sh wrap, so that per-kernel counts are per pass)")
    ap.add_argument("--steady-pass", action="store_true",
                    help="profiling aid: one eager pass (engine build), a 1.5 s pause, then ONE more eager pass; "
                         "tools/rocprof_summary.py stats keeps the kernels after the pause = a steady-state pass without setup kernels")
    ap.add_argument("--cpu-sample-frames", type=int, default=6,
                    help="frames of the CPU-oracle sample clip at the bench resolution (6 frames of 720x1280: ~2 min on 16 threads, "
                         "next to the GPU legs)")
    ap.add_argument("--cpu-timeout", type=float, default=480.0, help="hard limit of the CPU-oracle child process, seconds")
    ap.add_argument("--cpu-baseline-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-configs", action="store_true", help="skip the short timing of BASELINE config 2 (432x240x80) reported under `configs`")
    ap.add_argument("--replay-checks", type=int, default=16,
                    help="after the timed loop: this many more replays of the captured pass compared byte for byte with the first (`replay_consistency`; 0 = skip)")
    ap.add_argument("--no-stress", action="store_true", help="skip the stress-recipe leg (`stress`: non-tame weights / motion / border mask)")
    return ap.parse_args(argv)


def launch_plan(args, environ, argv, n_devices):
    """How this invocation runs (pure: tests/test_bench_launch_cpu.py).  ``--gpus N`` IS the number of ranks of the job:
      * WORLD_SIZE set (launched by torch.distributed.run, as the driver does for N > 1): it must equal --gpus -> ("run", world);
      * WORLD_SIZE unset and --gpus 1 -> ("run", 1);
      * WORLD_SIZE unset and --gpus N > 1 -> ("spawn", command): re-execute under ``python -m torch.distributed.run --nnodes=1
        --nproc-per-node N --master-addr 127.0.0.1`` (one process per GPU over RCCL; the reference's multi-process launch is
        ``mp.spawn(main_worker, nprocs=world_size)``, train.py:105) -- ``python bench.py --gpus 8`` used to bench ONE GPU silently.
    Raises SystemExit with the reason when the request cannot be met (more ranks than devices, WORLD_SIZE != --gpus)."""
    if args.gpus < 1:
        raise SystemExit(f"--gpus {args.gpus}: need at least one rank")
    if n_devices is not None and args.gpus > n_devices:
        raise SystemExit(f"--gpus {args.gpus}: this node exposes {n_devices} device(s) (one rank per GPU)")
    ws = environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ws}: launch with --nproc-per-node {args.gpus} (or drop WORLD_SIZE and let "
                             f"bench.py spawn its own ranks)")
        return "run", int(ws)
    if args.gpus == 1:
        return "run", 1
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(argv[0])] + list(argv[1:])
    return "spawn", cmd


class DeviceRuntime:
    """What bench.py needs from the device side.  The product runtime is this class (HIP device, RCCL, hipGraphs); the CPU test harness
    (tests/bench_cpu_harness.py) substitutes stand-in models on the CPU over gloo so that the launch / rendezvous / timing / reporting
    path of an N-rank job is exercised end to end without a GPU -- never used by a real run."""
    backend = "nccl"
    graphs = True           # hipGraph capture of the pass
    extras = True           # roofline / cpu_baseline / parity / precisions / configs / stress legs

    def n_devices(self):
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0

    def device(self, local):
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
        torch.cuda.set_device(local)
        return torch.device("cuda", local)

    def init_library(self):
        from propainter_amd import hip
        hip.lib()

    def models(self, dev, raft_dtype):
        from propainter_amd.synthetic import seeded_models
        return seeded_models(dev, raft_precision=raft_dtype)

    def pin(self, t):
        return t.pin_memory()

    def sync(self):
        import torch
        torch.cuda.synchronize()

    def event(self):
        import torch
        return torch.cuda.Event(enable_timing=True)

    def reset_peak(self, dev):
        import torch
        torch.cuda.reset_peak_memory_stats(dev)

    def peak_allocated(self, dev):
        import torch
        return torch.cuda.max_memory_allocated(dev)

    def peak_reserved(self, dev):
        import torch
        return torch.cuda.max_memory_reserved(dev)

    def reserved(self, dev):
        import torch
        return torch.cuda.memory_reserved(dev)

    def free_bytes(self, dev):
        import torch
        return torch.cuda.mem_get_info(dev)[0]

    def total_memory(self, dev):
        import torch
        return torch.cuda.get_device_properties(dev).total_memory

    def empty_cache(self):
        import torch
        torch.cuda.empty_cache()


def sample_clip(args):
    """The bounded sample both the CPU oracle and the parity check run on: a short synthetic clip AT THE TIMED RESOLUTION."""
    import numpy as np
    import scipy.ndimage
    from propainter_amd.synthetic import synthetic_clip, synthetic_mask
    L = args.cpu_sample_frames
    clip = synthetic_clip(L, args.height, args.width)
    m = scipy.ndimage.binary_dilation(synthetic_mask(args.height, args.width), iterations=4).astype(np.uint8) * 255
    return clip, np.repeat(m[None], L, 0)


def _cpu_baseline_worker(args):
    """Runs in a child process (bounded by a timeout in the parent): the CPU oracle, fp32, on the host cores.  Writes the
    composited frames to the .npy path given on the command line (the parity check of the parent reads them)."""
    import numpy as np
    import torch
    from oracle import propainter_oracle as O
    from propainter_amd.synthetic import seeded_models
    cores = int(os.environ.get("PP_CPU_THREADS", "1"))
    torch.set_num_threads(cores)
    clip, masks = sample_clip(args)
    raft, fc, gen = seeded_models("cpu")
    sds = {"raft": {k: v.float() for k, v in raft.fix_raft.state_dict().items()},
           "fc": {k: v.float() for k, v in fc.state_dict().items()},
           "gen": {k: v.float() for k, v in gen.state_dict().items()}}
    t0 = time.perf_counter()
    timers = {}
    with torch.no_grad():
        frames = O.inpaint_video(sds, clip, masks, masks, raft_iter=args.raft_iter, subvideo_length=args.subvideo_length,
                                 neighbor_length=args.neighbor_length, ref_stride=args.ref_stride, timers=timers)
    dt = time.perf_counter() - t0
    np.save(args.cpu_baseline_worker, np.stack(frames))
    print(json.dumps({"seconds": dt, "frames": len(frames), "cores": cores, "H": args.height, "W": args.width, "timers": timers}))


class CpuBaseline:
    """CPU oracle (fp32) on a bounded sample at the bench resolution: the full path over a short clip in a child process
    with a hard timeout, started next to the GPU work (it uses host cores only) and collected at the end.  The per-unit
    costs it measures (RAFT pair-direction, flow-completion flow, propagation step, generator window-frame) times the unit
    counts of the timed clip give the baseline."""

    def __init__(self, args, timeout=None):
        timeout = timeout or args.cpu_timeout
        import subprocess
        self.args, self.timeout = args, timeout
        try:
            self.avail = len(os.sched_getaffinity(0))
        except AttributeError:
            self.avail = os.cpu_count() or 1
        self.cores = max(1, min(self.avail - 1, 16))    # the oracle's small ops stop scaling (and oversubscribe) beyond ~16 threads
        env = dict(os.environ, PP_CPU_THREADS=str(self.cores), OMP_NUM_THREADS=str(self.cores), MKL_NUM_THREADS=str(self.cores),
                   HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        fd, self.out_path = tempfile.mkstemp(suffix=".npy", prefix="pp_oracle_")
        os.close(fd)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", self.out_path,
               "--height", str(args.height), "--width", str(args.width),
               "--cpu-sample-frames", str(args.cpu_sample_frames), "--raft_iter", str(args.raft_iter),
               "--subvideo_length", str(args.subvideo_length), "--neighbor_length", str(args.neighbor_length),
               "--ref_stride", str(args.ref_stride)]
        self.t_start = time.perf_counter()
        self.proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)

    def collect(self):
        """-> (cpu_baseline dict, oracle frames uint8 [L,H,W,3] or None)"""
        import numpy as np
        import subprocess
        args, cores = self.args, self.cores
        try:
            left = max(1.0, self.timeout - (time.perf_counter() - self.t_start))
            so, _ = self.proc.communicate(timeout=left)
            rec = json.loads(so.decode().strip().splitlines()[-1])
            frames = np.load(self.out_path)
        except Exception as e:  # timeout / crash: report it instead of blocking the bench
            try:
                self.proc.kill()
            except Exception:
                pass
            return ({"value": None, "unit": "frames/s", "cores": cores, "kind": "port",
                     "sample": f"CPU oracle sample did not finish within {self.timeout} s on {cores} threads ({type(e).__name__})"}, None)
        finally:
            try:
                os.unlink(self.out_path)
            except OSError:
                pass
        dt, Ls, H, W, tm = rec["seconds"], rec["frames"], rec["H"], rec["W"], rec["timers"]
        # unit counts of the TIMED clip: RAFT pair-directions, completed flows (both directions), propagation
        # steps, generator window-frames (sum of t over the windows)
        from propainter_amd.pipeline import window_schedule
        L = args.frames
        sched = window_schedule(L, args.neighbor_length, args.ref_stride, args.subvideo_length)
        units = {"raft_pair_directions": 2 * (L - 1), "fc_flows": 2 * (L - 1), "prop_steps": 2 * (L - 1),
                 "gen_window_frames": sum(len(nb) + len(ref) for nb, ref in sched)}
        per_unit = {"raft_pair_direction_s": tm["raft_s"] / tm["raft_pair_directions"], "fc_flow_s": tm["fc_s"] / tm["fc_flows"],
                    "prop_step_s": tm["prop_s"] / tm["prop_steps"], "gen_window_frame_s": tm["gen_s"] / tm["gen_window_frames"]}
        est = (units["raft_pair_directions"] * per_unit["raft_pair_direction_s"] + units["fc_flows"] * per_unit["fc_flow_s"] +
               units["prop_steps"] * per_unit["prop_step_s"] + units["gen_window_frames"] * per_unit["gen_window_frame_s"])
        return ({"value": L / est, "unit": "frames/s", "cores": cores, "kind": "port",
                 "sample": f"CPU oracle fp32 (oracle/propainter_oracle.py), whole path on a {Ls}-frame {W}x{H} synthetic clip, measured at "
                           f"this size on {cores} threads ({self.avail} available) UNDER CONTENTION -- the GPU legs of this bench (their host threads, graph "
                           f"captures, H2D copies) ran on the same cores at the same time, and the 6-frame generator windows under-state what an "
                           f"18-frame window costs the CPU: {dt:.1f} s "
                           f"(RAFT {tm['raft_s']:.1f} s / {tm['raft_pair_directions']} pair-directions, flow completion {tm['fc_s']:.1f} s / "
                           f"{tm['fc_flows']} flows, image propagation {tm['prop_s']:.1f} s / {tm['prop_steps']} steps, generator "
                           f"{tm['gen_s']:.1f} s / {tm['gen_windows']} windows of {tm['gen_window_frames'] // max(1, tm['gen_windows'])} frames); "
                           f"value = {L} frames / (unit counts of the {L}-frame clip x these per-unit costs) = {est:.0f} s -- EXTRAPOLATED "
                           f"in clip length only; the generator's per-frame cost grows with the window length (attention), so longer "
                           f"windows (17 frames on average here) would cost the CPU more, not less",
                 "contention": "measured while the GPU legs of the same bench run used the host (a reported baseline, not a target)",
                 "measured_sample_seconds": dt, "measured_sample_fps": Ls / dt, "per_unit_seconds": per_unit, "unit_counts": units,
                 "estimated_clip_seconds": est}, frames)


def parity_of(got_u8, ref_u8, masks_u8, gt_u8=None):
    """Byte-level comparison of composited frames with the oracle's (uint8 [L,H,W,3]); the PSNR is also given over the
    hole only (outside the dilated mask both are the input frame, which inflates a whole-frame PSNR).  gt_u8: the unmasked
    clip -- PSNR(HIP, GT) and PSNR(oracle, GT) with core/metrics.py:20-36's formula and their difference (north_star: the
    output PSNR within 0.05 dB of the reference's)."""
    import numpy as np
    from oracle import propainter_oracle as O
    d = np.abs(got_u8.astype(np.int16) - ref_u8.astype(np.int16))
    hole = np.broadcast_to((masks_u8 > 0)[..., None], got_u8.shape)
    mse_h = float((d[hole].astype(np.float64) ** 2).mean()) if hole.any() else 0.0
    vs_gt = {}
    if gt_u8 is not None:
        pg = float(np.mean([O.psnr(got_u8[i], gt_u8[i]) for i in range(len(gt_u8))]))      # per frame, averaged (scripts/evaluate_propainter.py:199-205)
        pr = float(np.mean([O.psnr(ref_u8[i], gt_u8[i]) for i in range(len(gt_u8))]))
        vs_gt = {"psnr_vs_ground_truth_db": round(pg, 4), "oracle_psnr_vs_ground_truth_db": round(pr, 4),
                 "psnr_delta_vs_oracle_db": round(abs(pg - pr), 4)}
    return {**vs_gt, "psnr_db": round(O.psnr(got_u8, ref_u8), 2),
            "psnr_hole_db": round(float("inf") if mse_h == 0 else 20.0 * np.log10(255.0 / np.sqrt(mse_h)), 2),
            "max_abs": int(d.max()), "bytes_differ_frac": float((d > 0).mean()),
            "bytes_differ_frac_hole": float((d[hole] > 0).mean()) if hole.any() else 0.0,
            "bytes_off_by_more_than_1_frac_hole": float((d[hole] > 1).mean()) if hole.any() else 0.0}


TIMED_GOLDENS = {
    # (H, W, L, raft_iter, subvideo_length, neighbor_length, ref_stride) of the timed clip -> committed oracle golden (oracle/make_golden_synth.py)
    (720, 1280, 80, 20, 80, 10, 10): "synth_c3_720x1280x80.npz",
    (240, 432, 80, 20, 80, 10, 10): "synth_c2_432x240x80.npz",
}


def timed_golden_name(args):
    """The committed golden of the clip rank 0 TIMES (seed 2023, tame recipe), or None when this configuration has none."""
    return TIMED_GOLDENS.get((args.height, args.width, args.frames, args.raft_iter, args.subvideo_length, args.neighbor_length, args.ref_stride))


def timed_output_parity(timed_out, fn, submission):
    """The last timed step's host bytes vs the committed fp32 CPU-oracle golden of the same clip: inputs regenerated from the seeds and
    checked against the fixture's digests, the oracle's bytes inside the dilated mask (outside it both are the input frame)."""
    import hashlib
    import numpy as np
    from propainter_amd.synthetic import case_inputs
    path = os.path.join(ROOT, "tests", "golden", fn)
    if not os.path.exists(path):
        return {"error": f"{fn} not found"}
    g = np.load(path)
    gclip, gmasks = case_inputs(int(g["L"]), int(g["H"]), int(g["W"]), str(g["recipe"]))
    dg = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    if dg(gclip) != str(g["frames_sha256"]) or dg(gmasks) != str(g["masks_sha256"]):
        return {"error": f"{fn}: regenerated inputs do not match the fixture's digests"}
    if timed_out.shape != gclip.shape:
        return {"error": f"timed output {timed_out.shape} vs golden clip {gclip.shape}"}
    hole = gmasks > 0
    ref = gclip.copy()
    ref[hole] = g["comp_hole"]
    rec = parity_of(timed_out, ref, gmasks, gt_u8=gclip)
    rec["what"] = (f"host_out after the LAST TIMED step ({submission}) vs the committed fp32 CPU-oracle golden {fn} of the same "
                   f"{int(g['L'])}-frame {int(g['W'])}x{int(g['H'])} clip (16 windows, reference frames up to +-40 frames away)")
    rec["frames_differing_outside_hole"] = int(sum(bool((timed_out[i][~hole[i]] != gclip[i][~hole[i]]).any()) for i in range(len(gclip))))
    return rec


def main(argv=None, runtime=None):
    """argv / runtime: only the CPU test harness passes them (tests/bench_cpu_harness.py); a real run uses sys.argv and DeviceRuntime."""
    argv = list(sys.argv) if argv is None else list(argv)
    args = parse(argv[1:])
    if args.cpu_baseline_worker:
        _cpu_baseline_worker(args)
        return
    rt = runtime or DeviceRuntime()
    if rt.n_devices() == 0:
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    how, plan = launch_plan(args, os.environ, argv, rt.n_devices())
    if how == "spawn":
        # --gpus N without a launcher: become the launcher (one rank per GPU; rank 0 of the child job prints the JSON line)
        sys.stderr.write(f"[bench] --gpus {args.gpus}: launching {args.gpus} ranks: {' '.join(plan)}\n")
        sys.stderr.flush()
        os.execv(plan[0], plan)
    world = plan
    import datetime
    import numpy as np
    import scipy.ndimage
    import torch
    import torch.distributed as dist
    from propainter_amd import hip
    from propainter_amd.pipeline import InferenceConfig, run_clip, window_schedule
    from propainter_amd.synthetic import synthetic_clip, synthetic_mask

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = rt.device(local)
    cpu_job = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and rt.extras:
        cpu_job = CpuBaseline(args)           # host cores only; runs while the GPU legs below execute
    seen_world = 1
    if world > 1:
        # a rank that dies takes the job down within the timeout instead of leaving its peers in a collective for ever
        kw = {"device_id": dev} if rt.backend == "nccl" else {}
        dist.init_process_group(rt.backend, timeout=datetime.timedelta(seconds=int(os.environ.get("PP_BENCH_DIST_TIMEOUT_S", "600"))), **kw)
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)                  # the world size the collective library itself sees (RCCL over xGMI on the GPU box)
        seen_world = int(one.item())
        if seen_world != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus}, but the process group has {dist.get_world_size()} ranks and an all-reduce of ones gives {seen_world}")
    rt.init_library()

    H, W, L = args.height, args.width, args.frames
    fp16 = not args.fp32
    models = rt.models(dev, args.raft_dtype)
    cfg = InferenceConfig(raft_iter=args.raft_iter, subvideo_length=args.subvideo_length,
                          neighbor_length=args.neighbor_length, ref_stride=args.ref_stride, fp16=fp16,
                          window_streams=args.window_streams, raft_streams=args.raft_streams)
    m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.uint8) * 255
    sharded = bool(args.sharded)
    exchange_stats = {}
    if sharded:
        from propainter_amd.sharding import ShardPlan, can_shard, run_clip_sharded
        # ONE clip for the whole job (same seed on every rank: the raw input is host-resident everywhere, as in the CLI)
        clip = synthetic_clip(L, H, W, seed=2023)
        masks_np = np.repeat(m[None], L, 0)
        use_ranks = world > 1 and can_shard(L, cfg, world)
        if world > 1 and not use_ranks:
            raise SystemExit(f"--sharded: a {L}-frame clip does not split into sub-videos of {args.subvideo_length} over {world} ranks")
        clip_pin, masks_pin = rt.pin(torch.from_numpy(clip)), rt.pin(torch.from_numpy(masks_np))
        own = ShardPlan(L, cfg, world).own[rank] if use_ranks else (0, L)
        host_out = rt.pin(torch.empty((max(1, own[1] - own[0]), H, W, 3), dtype=torch.uint8))
        frames_dev = masks_dev = None
        if not use_ranks:
            frames_dev, masks_dev = clip_pin.to(dev), masks_pin.to(dev)
    else:
        # every rank inpaints its own clip (independent windows per GPU; weak scaling)
        clip = synthetic_clip(L, H, W, seed=2023 + rank)
        frames_dev = torch.from_numpy(clip).to(dev)
        masks_dev = torch.from_numpy(np.repeat(m[None], L, 0)).to(dev)
        host_out = rt.pin(torch.empty((L, H, W, 3), dtype=torch.uint8))
        use_ranks = False

    def eager_step(stage_hook=None, cfg=cfg):
        if use_ranks:       # sub-video shards: every rank uploads the slice of the raw input it needs inside the step
            lo, comp = run_clip_sharded(models, clip_pin, masks_pin, masks_pin, cfg, dev, stats=exchange_stats)
            if comp.shape[0]:
                host_out[:comp.shape[0]].copy_(comp, non_blocking=True)
            return
        comp = run_clip(models, frames_dev, masks_dev, masks_dev, cfg, dev, stage_hook=stage_hook)
        host_out.copy_(comp, non_blocking=True)

    if args.single_pass or args.steady_pass:
        eager_step()
        rt.sync()
        if args.steady_pass:       # a second, steady-state pass behind a 1.5 s pause: tools/rocprof_summary.py keeps the kernels after the pause
            time.sleep(1.5)        # (the first pass builds the engines: ~1 200 weight-packing copies and elementwise kernels that are not per-pass work)
            eager_step()
            rt.sync()
        print(json.dumps({"single_pass": True, "height": H, "width": W, "frames": L, "raft_dtype": args.raft_dtype}))
        return
    # ---- setup (untimed, like model load in the reference protocol): one eager pass builds the engines (weight
    # packing, K tables, window tables) and primes the allocator; by default the pass is then captured in a hipGraph
    eager_step()
    eager_step()                      # the allocator settles on the second pass (its blocks are carved during the first)
    rt.sync()
    rt.reset_peak(dev)
    t_e = time.perf_counter()
    eager_step()
    rt.sync()
    eager_ms = (time.perf_counter() - t_e) * 1e3
    peak_eager = rt.peak_allocated(dev)
    peak_eager_reserved = rt.peak_reserved(dev)      # what the caching allocator holds from the driver (all stream pools)
    exchange_stats.clear()
    graph = None
    capture_s = None
    sgraph = None
    def vote(ok):
        """all ranks or none (MIN over the ranks of a 0 / 1 flag)"""
        t = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    if not args.eager and use_ranks and rt.graphs:
        # sub-video shards: the compute segments between the four halo exchanges as hipGraphs, the exchanges stay eager RCCL
        # point-to-point ops (sharding.ShardedClipGraph); falls back to eager submission if the capture is refused.
        # The capture executes its exchanges for real, in lock step with the peers: a rank that fails ALONE (out of memory) would
        # leave them in batch_isend_irecv for ever.  So (1) the ranks vote on a memory check BEFORE anyone starts, and (2) the
        # capture itself votes after every segment (ShardedClipGraph.capture(vote=...)): a failing rank votes 0, every rank
        # abandons the capture at the same exchange, and the bench falls back to eager launches on all ranks.
        from propainter_amd.sharding import CaptureAborted, ShardedClipGraph, dist_exchanger
        t_c = time.perf_counter()
        if rt.reserved(dev) > rt.total_memory(dev) // 3:
            rt.empty_cache()                  # long clips / 1080p: the eager pools back to the driver before the graphs build their own
        need = int(1.15 * peak_eager_reserved)           # the graphs' private pool grows to about what the eager pass reserved
        have = rt.free_bytes(dev) + rt.reserved(dev)
        if not vote(have >= need):
            if have < need:
                sys.stderr.write(f"[bench] rank {rank}: {have / 1e9:.1f} GB available for the shard graphs, ~{need / 1e9:.1f} GB needed\n")
            sys.stderr.write(f"[bench] rank {rank}: the ranks voted against the sharded hipGraph capture (memory); eager launches\n")
        else:
            # the static inputs are allocated before the lock-step capture starts: a rank that cannot (out of memory) must not die alone
            # and leave its peers in the first vote's all-reduce until the process-group timeout -- its failure is a 0 vote like any other
            try:
                sgraph = ShardedClipGraph(models, L, H, W, cfg, dev, rank, world)
                sgraph.load(clip_pin, masks_pin, masks_pin)
                built, why = True, None
            except Exception as e:      # noqa: BLE001 -- whatever it is, the ranks must learn of it together
                built, why, sgraph = False, f"{type(e).__name__}: {e}", None
            if not vote(built):
                sys.stderr.write(f"[bench] rank {rank}: the ranks voted against the sharded hipGraph capture "
                                 f"({why or 'a peer could not build its static inputs'}); eager launches\n")
                sgraph = None
                rt.sync()
                rt.empty_cache()
            else:
                try:
                    sgraph.capture(dist_exchanger(dev, None, None), vote=vote)
                except CaptureAborted as e:
                    sys.stderr.write(f"[bench] sharded hipGraph capture abandoned on rank {rank} ({e}); eager launches\n")
                    sgraph = None
                    rt.sync()
                    rt.empty_cache()
        capture_s = time.perf_counter() - t_c
    if not args.eager and not use_ranks and rt.graphs:      # one clip per rank: the whole pass as ONE hipGraph
        from propainter_amd.pipeline import ClipGraph
        t_c = time.perf_counter()
        try:
            # the graph gets a private pool as large as the eager pass's: when the eager pools already hold more than a third of the
            # device (long clips / 1080p), hand them back to the driver first
            big = rt.reserved(dev) > rt.total_memory(dev) // 3
            graph = ClipGraph(models, L, H, W, cfg, dev, example=(frames_dev, masks_dev, masks_dev), release_eager_pool=big, forked_branches=args.graph_lanes)
            rt.sync()
        except Exception as e:       # capture refused (driver / runtime state): measure the eager submission instead of dying
            sys.stderr.write(f"[bench] hipGraph capture failed on rank {rank} ({type(e).__name__}: {e}); falling back to eager launches\n")
            graph = None
            rt.sync()
        capture_s = time.perf_counter() - t_c

    sgraph_x = None
    if sgraph is not None:
        from propainter_amd.sharding import dist_exchanger
        sgraph_x = dist_exchanger(dev, None, exchange_stats)

    def step(stage_hook=None):
        if sgraph is not None and stage_hook is None:
            sgraph.load(clip_pin, masks_pin, masks_pin)       # every rank uploads the slice of the raw input it needs inside the step
            lo_, comp = sgraph.replay(sgraph_x)
            if comp.shape[0]:
                host_out[:comp.shape[0]].copy_(comp, non_blocking=True)
            return
        if graph is None or stage_hook is not None:
            return eager_step(stage_hook)
        host_out.copy_(graph.replay(), non_blocking=True)

    def fence():
        rt.sync()
        if world > 1:
            dist.barrier()
        rt.sync()

    for _ in range(args.warmup):
        step()
    fence()
    exchange_stats.clear()
    # per-step markers (diagnostic only, no synchronisation inside the timed region): a HIP event after each step's
    # last launch and the host clock when the step has been fully *submitted*
    ev = [rt.event() for _ in range(args.steps + 1)]
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
    # the bytes the LAST TIMED step left in host_out, kept aside before any later leg reuses the buffer: compared below with the
    # committed fp32-oracle golden of this very clip (tests/golden/synth_c3_720x1280x80.npz; inference_propainter.py:407-452)
    timed_out = None
    if rank == 0 and not sharded and args.steps > 0 and timed_golden_name(args) is not None:
        timed_out = host_out.numpy().copy()
    # every replay the same bytes?  (round 6: ~5 % of the replays of rounds 2-5's graph left a few hundred wrong bytes in one frame --
    # profiles/r6_replay_bytes.txt; one comparison per run cannot see that.)  Outside the timed region, on the device.
    replay_consistency = None
    if rank == 0 and world == 1 and graph is not None and rt.extras and args.replay_checks > 0 and dev.type == "cuda":
        differing, worst = graph.self_check(args.replay_checks)          # pipeline.ClipGraph.self_check
        rt.sync()
        replay_consistency = {"replays_compared_with_the_first": args.replay_checks, "differing": differing, "max_abs": worst,
                              "last_equals_last_timed_step": bool(timed_out is not None and np.array_equal(graph.out.cpu().numpy(), timed_out))}
    host_submit = [host_submit[0]] + [host_submit[i] - host_submit[i - 1] for i in range(1, len(host_submit))]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    fps = (L if sharded else world * L) * args.steps / elapsed
    peak_total = rt.peak_allocated(dev)
    exch = {k: dict(v, ms_per_step=v["ms"] / args.steps, sent_bytes_per_step=v["sent_bytes"] // args.steps,
                    recv_bytes_per_step=v["recv_bytes"] // args.steps) for k, v in exchange_stats.items()} if use_ranks else None
    exch_plan = None
    if sharded and rank == 0:
        # what the halo exchanges of THIS clip move on the BASELINE rank counts (4 / 8 GPUs), from the shard plan's index ranges -- the
        # bytes a real multi-GPU run was checked against over gloo (tests/test_sharding_cpu.py); reported also when this run has one GPU
        from propainter_amd.sharding import can_shard, plan_exchange_bytes
        exch_plan = {}
        for nw in sorted({4, 8, world} - {1}):
            if can_shard(L, cfg, nw):
                per_rank = plan_exchange_bytes(L, cfg, nw, args.height, args.width, 4 if args.fp32 else 2)
                busiest = max(range(nw), key=lambda r_: sum(v[0] for v in per_rank[r_].values()))
                exch_plan[f"{nw}_ranks"] = {"busiest_rank": busiest,
                                            "sent_MB": {k: round(v[0] / 1e6, 1) for k, v in per_rank[busiest].items()},
                                            "sent_total_MB": round(sum(v[0] for v in per_rank[busiest].values()) / 1e6, 1),
                                            "ms_if_split_over_two_153GBps_xgmi_links": round(sum(v[0] for v in per_rank[busiest].values()) / 153e9 * 1e3 / 2, 2)}

    # ---- one instrumented step: stage split + per-kernel-class HIP-event timing (not part of `value`)
    stages, kernels, roof = None, None, None
    if rank == 0 and not args.no_profile and not use_ranks and rt.extras:
        marks = []

        stage_peaks = {}

        def hook(name):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e, time.perf_counter()))
            # peak of live device memory inside the stage that just ended (caching-allocator statistics: no synchronisation)
            stage_peaks[name] = torch.cuda.max_memory_allocated(dev) / 1e9
            torch.cuda.reset_peak_memory_stats(dev)
        with hip.KernelProfiler(detail=args.detail) as kp:
            t1 = time.perf_counter()
            # windows serialised on one stream here: a launch's event pair then brackets that launch alone (with
            # concurrent windows the durations of overlapping launches would be counted twice)
            eager_step(hook, dataclasses.replace(cfg, window_streams=1, raft_streams=1))
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
        traffic, traffic_src = pmc_traffic(name, args.raft_dtype)
        common = {"kernel": name, "traffic": traffic, "traffic_source": traffic_src,
                  "launches": v["launches"], "avg_launch_us": v["avg_us"],
                  "timed_with": "HIP events around every launch of one eager pass, generator windows serialised (window_streams=1)",
                  "algorithmic_bytes_per_launch": v["bytes"] / max(1, v["launches"]),
                  "share_of_kernel_time": v["ms"] / sum(x["ms"] for x in kernels.values())}
        if name.startswith("conv_gemm") or name == "sparse_window_attention":
            peak = PEAK_TFLOPS["f32" if name.endswith("f32") else "f16x3" if name.endswith("f16x3") else "f16"]
            roof = dict(common, bound="mfma", achieved=v["tflops"], peak=peak, unit="TFLOP/s", frac=v["tflops"] / peak,
                        algorithmic_flop_per_launch=v["flops"] / max(1, v["launches"]))
            if name.endswith("f16x3"):
                roof["note"] = ("split-plane (fp32-class) convolutions of RAFT: `achieved` counts the ALGORITHMIC FLOPs of the fp32 layers; "
                                "each product executes as three fp16 MFMA products, so `peak` = dense fp16 MFMA peak / 3 "
                                f"(executed: {3 * v['tflops']:.0f} TFLOP/s of {PEAK_TFLOPS['f16']:.0f})")
            # every matrix-core class of the pass next to the dominant one
            roof["other_mfma_classes"] = {k: {"achieved": kv["tflops"], "frac": kv["tflops"] / PEAK_TFLOPS["f32" if k.endswith("f32") else "f16x3" if k.endswith("f16x3") else "f16"],
                                              "launches": kv["launches"], "ms": kv["ms"]}
                                          for k, kv in kernels.items() if k != name and (k.startswith("conv_gemm") or k == "sparse_window_attention") and kv["flops"] > 0}
        else:
            roof = dict(common, bound="hbm", achieved=v["gbs"], peak=PEAK_HBM_GBS, unit="GB/s", frac=v["gbs"] / PEAK_HBM_GBS)
        stages["instrumented_step_wall_ms"] = prof_wall * 1e3
        stages["peak_allocated_GB_by_stage"] = {k: v for k, v in stage_peaks.items() if k != "start"}

    # ---- the pass at the other RAFT precisions, submitted the same way as the headline (own hipGraph, mean of 2 replays
    # after one untimed replay); the reference's own RAFT arithmetic is fp32
    raft_precisions = None
    raft = models[0]
    submission = ("hipGraph replays of the compute segments between the halo exchanges (sharding.ShardedClipGraph)" if sgraph is not None else
                  "eager (Python launches)" if graph is None else
                  "hipGraph replay of the whole pass (pipeline.ClipGraph" + (", window / RAFT lanes as forked branches)" if args.graph_lanes else
                                                                             ", one chain of launches: no forked branches)"))
    lanes_in_effect = (1, 1) if (graph is not None or sgraph is not None) and not args.graph_lanes else (args.window_streams, args.raft_streams)
    if rank == 0 and world == 1 and not args.no_precisions and not sharded and rt.extras:
        raft_precisions = {args.raft_dtype: {"value": fps, "ms_per_step": ms_per_step, "timed": "headline (see value)"}}
        had_graph = graph is not None
        graph = None                      # releases the headline graph's private pool before the other engines are built
        torch.cuda.empty_cache()
        for prec in ("f16x3", "f16", "f32"):
            if prec == args.raft_dtype:
                continue
            raft.precision = prec
            g = None
            try:
                eager_step()              # builds this precision's engine, settles the allocator
                torch.cuda.synchronize()
                how = "mean of 2 eager steps"
                if had_graph:
                    try:
                        from propainter_amd.pipeline import ClipGraph
                        torch.cuda.empty_cache()          # the eager pass's cached blocks back to the driver: the capture builds its own pool
                        g = ClipGraph(models, L, H, W, cfg, dev, example=(frames_dev, masks_dev, masks_dev), forked_branches=args.graph_lanes)
                        how = "mean of 2 hipGraph replays"
                    except Exception as e:
                        sys.stderr.write(f"[bench] capture at RAFT {prec} failed ({type(e).__name__}: {e}); eager\n")
                        g = None
                one = (lambda: host_out.copy_(g.replay(), non_blocking=True)) if g is not None else eager_step
                one()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                one()
                one()
                torch.cuda.synchronize()
                dt1 = (time.perf_counter() - t1) / 2
                raft_precisions[prec] = {"value": L / dt1, "ms_per_step": dt1 * 1e3, "timed": how}
            except Exception as e:
                raft_precisions[prec] = {"value": None, "error": f"{type(e).__name__}: {e}"}
            finally:
                raft.precision = args.raft_dtype
                g = None
                torch.cuda.empty_cache()

    # ---- round 5: the data-dependent slow paths of the headline pass, BASELINE config 2, windows with reference frames, and the
    # STRESS recipe (non-tame weights / motion / masks) -- each against a committed oracle golden (tests/golden/synth_*.npz:
    # oracle/make_golden_synth.py; inputs and weights are regenerated from seeds and checked against the fixture's digests)
    fallback, configs, stress, parity_refs = None, None, None, None
    if rank == 0 and world == 1 and not sharded and rt.extras:
        from propainter_amd.pipeline import ClipGraph
        graph = None
        rt.empty_cache()

        def golden_parity(fn, mdl, cfg_g):
            """HIP path at the timed precision split on a committed synthetic golden: bytes inside the dilated mask vs the fp32 CPU oracle's"""
            import hashlib
            path = os.path.join(ROOT, "tests", "golden", fn)
            if not os.path.exists(path):
                return {"error": f"{fn} not found"}
            g = np.load(path)
            from propainter_amd.synthetic import case_inputs
            gclip, gmasks = case_inputs(int(g["L"]), int(g["H"]), int(g["W"]), str(g["recipe"]))
            dg = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
            if dg(gclip) != str(g["frames_sha256"]) or dg(gmasks) != str(g["masks_sha256"]):
                return {"error": f"{fn}: regenerated inputs do not match the fixture's digests"}
            got = run_clip(mdl, gclip, gmasks, gmasks, cfg_g, dev).cpu().numpy()
            hole = gmasks > 0
            ref = gclip.copy()
            ref[hole] = g["comp_hole"]
            rec = parity_of(got, ref, gmasks, gt_u8=gclip)
            rec["clip"] = f"{int(g['L'])}-frame {int(g['W'])}x{int(g['H'])} {str(g['recipe'])} clip, committed oracle golden {fn}"
            return rec

        try:
            # (a) fallback rates of the headline clip (tame recipe): one eager pass with the counting entry points
            with hip.FallbackStats(dev) as fs:
                eager_step()
                fallback = dict(fs.read(), clip="the timed clip (tame recipe)")
            # (b) windows WITH reference frames at the timed resolution (the live cpu_baseline sample is 6 frames: no references)
            parity_refs = golden_parity("synth_c3_720x1280x18.npz", models, cfg)
            # (c) BASELINE config 2: 432x240x80, same precision split, own hipGraph, 2 replays + parity against the 80-frame golden
            if not args.no_configs:
                c2clip, c2m = synthetic_clip(80, 240, 432), scipy.ndimage.binary_dilation(synthetic_mask(240, 432), iterations=4).astype(np.uint8) * 255
                c2f, c2mk = torch.from_numpy(c2clip).to(dev), torch.from_numpy(np.repeat(c2m[None], 80, 0)).to(dev)
                cfg2 = dataclasses.replace(cfg, subvideo_length=80, neighbor_length=10, ref_stride=10)
                run_clip(models, c2f, c2mk, c2mk, cfg2, dev)
                g2 = ClipGraph(models, 80, 240, 432, cfg2, dev, example=(c2f, c2mk, c2mk), forked_branches=args.graph_lanes)
                g2.replay()
                rt.sync()
                t1 = time.perf_counter()
                g2.replay()
                g2.replay()
                rt.sync()
                dt2 = (time.perf_counter() - t1) / 2
                configs = {"c2_432x240x80": {"value": 80 / dt2, "unit": "frames/s", "ms_per_step": dt2 * 1e3, "timed": "mean of 2 hipGraph replays",
                                             "frac_of_f16_peak_on_algorithmic_flops": 63.9 / dt2 / PEAK_TFLOPS["f16"],
                                             "parity": golden_parity("synth_c2_432x240x80.npz", models, cfg2)}}
                g2 = None
                rt.empty_cache()
            # (d) the stress recipe: frames/s of the SAME configuration on non-tame data, its fallback rates, parity on a 6-frame stress clip
            if not args.no_stress:
                from propainter_amd.synthetic import stress_clip, stress_mask
                from propainter_amd.synthetic import seeded_models as _seeded
                smodels = _seeded(dev, raft_precision=args.raft_dtype, recipe="stress")
                sclip_ = stress_clip(L, H, W)
                sm_ = scipy.ndimage.binary_dilation(stress_mask(H, W), iterations=4).astype(np.uint8) * 255
                sf, smk = torch.from_numpy(sclip_).to(dev), torch.from_numpy(np.repeat(sm_[None], L, 0)).to(dev)
                run_clip(smodels, sf, smk, smk, cfg, dev)
                with hip.FallbackStats(dev) as fs:
                    run_clip(smodels, sf, smk, smk, cfg, dev)
                    sfall = fs.read()
                finite = True
                try:
                    from propainter_amd.model.modules.flow_comp_raft import assert_finite_flows
                    assert_finite_flows(smodels[0])
                except FloatingPointError as e:
                    finite = f"{type(e).__name__}: {e}"
                # where the stress pass spends its time, next to the headline's `kernels` table (one instrumented eager pass, windows serialised)
                with hip.KernelProfiler() as kps:
                    run_clip(smodels, sf, smk, smk, dataclasses.replace(cfg, window_streams=1, raft_streams=1), dev)
                    rt.sync()
                skern = {k: round(v["ms"], 1) for k, v in sorted(kps.summary().items(), key=lambda kv: -kv[1]["ms"])[:8]}
                rt.empty_cache()
                gs = ClipGraph(smodels, L, H, W, cfg, dev, example=(sf, smk, smk), forked_branches=args.graph_lanes)
                gs.replay()
                rt.sync()
                t1 = time.perf_counter()
                gs.replay()
                gs.replay()
                rt.sync()
                dts = (time.perf_counter() - t1) / 2
                gs = None
                rt.empty_cache()
                stress = {"value": L / dts, "unit": "frames/s", "ms_per_step": dts * 1e3, "vs_headline": (L / dts) / fps,
                          "timed": "mean of 2 hipGraph replays of the whole pass, same configuration / precision split as the headline",
                          "recipe": "RECIPES_STRESS (flow head x1.0, offset heads x1.0), stress_clip (two layers in opposite directions at 8-48 px/frame "
                                    "+ occluder), stress_mask (outpainting border + one hole per attention window: every window masked)",
                          "mask_area_frac": float((sm_ > 0).mean()), "flows_finite": finite, "fallback": sfall,
                          "kernels_ms": skern, "headline_kernels_ms": ({k: round(v["ms"], 1) for k, v in kernels.items() if k in skern} if kernels else None),
                          "parity": golden_parity("synth_stress_720x1280x6.npz", smodels, cfg)}
                smodels = None
                rt.empty_cache()
        except Exception as e:      # noqa: BLE001 -- the extra legs must never take the headline line down
            import traceback
            sys.stderr.write("[bench] round-5 extra legs failed:\n" + traceback.format_exc())
            stress = stress or {"error": f"{type(e).__name__}: {e}"}

    # ---- parity of the timed configuration (and of the other RAFT precisions) against the CPU oracle's frames
    cpu, parity = None, None
    if cpu_job is not None:
        sclip, smasks = sample_clip(args)
        got = {}
        for prec in ([args.raft_dtype] + ([p for p in ("f16x3", "f16", "f32") if p != args.raft_dtype] if raft_precisions else [])):
            raft.precision = prec
            try:
                got[prec] = run_clip(models, sclip, smasks, smasks, cfg, dev).cpu().numpy()
            except Exception as e:
                got[prec] = f"{type(e).__name__}: {e}"
            finally:
                raft.precision = args.raft_dtype
        torch.cuda.synchronize()
        cpu, ref_frames = cpu_job.collect()
        if ref_frames is not None:
            desc = {"clip": f"{len(sclip)}-frame {W}x{H} synthetic clip of the cpu_baseline sample (the TIMED resolution), seeded weights",
                    "reference": "CPU oracle fp32 (oracle/propainter_oracle.py)", "stages_dtype": "f16" if fp16 else "f32",
                    "ground_truth": "the unmasked synthetic frames (PSNR per frame, averaged: core/metrics.py:20-36, scripts/evaluate_propainter.py)"}
            for prec, g in got.items():
                rec = parity_of(g, ref_frames, smasks, gt_u8=sclip) if not isinstance(g, str) else {"error": g}
                if prec == args.raft_dtype:
                    parity = dict(desc, raft_dtype=prec, dtype="f16" if fp16 else "f32", **rec)
                if raft_precisions and prec in raft_precisions:
                    raft_precisions[prec]["parity"] = rec

    parity_timed = None
    if timed_out is not None and args.raft_dtype in ("f16x3", "f32"):
        try:
            parity_timed = timed_output_parity(timed_out, timed_golden_name(args), submission)
        except Exception as e:      # noqa: BLE001 -- a checker leg never takes the headline line down
            parity_timed = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        sched = window_schedule(L, args.neighbor_length, args.ref_stride, args.subvideo_length)
        par = (f"sub-video shards of one clip x{world}" if sharded else f"clip-sharded x{world}")
        work = (f"{H}x{W} {L}-frame clip, neighbor_length={args.neighbor_length} ref_stride={args.ref_stride} "
                f"subvideo_length={args.subvideo_length} raft_iter={args.raft_iter}, "
                + ("ONE clip sharded by sub-video over the ranks" if sharded else "one clip per GPU"))
        if roof is not None and roof["bound"] == "mfma" and (H, W, L) == (720, 1280, 80) and not sharded:
            # whole-pass figure on SURVEY 8(d)'s minimal algorithmic FLOPs (independent of how the engine executes them)
            roof["whole_pass_algorithmic_tflops"] = C3_ALGORITHMIC_TFLOP / (ms_per_step * 1e-3)
            roof["whole_pass_frac_of_f16_peak"] = roof["whole_pass_algorithmic_tflops"] / PEAK_TFLOPS["f16"]
        out = {
            "metric": "inpainted frames/sec (whole path, 80-frame window)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "collective_world_size": seen_world, "collective_backend": (rt.backend if world > 1 else None),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            # arithmetic of the timed pass: stages B-D / RAFT (the reference's --fp16 run: fp16 stages, fp32 RAFT -- inference_propainter.py:311,333-337)
            "dtype": ("f16" if fp16 else "f32") + " stages + " + {"f16x3": "f16x3 RAFT (fp32-class: 3 fp16 MFMA products per product, fp32 accumulate)",
                                                                  "f32": "f32 RAFT", "f16": "f16 RAFT (narrower than the reference's fp32 RAFT)"}[args.raft_dtype],
            "value_raft_f16": (raft_precisions or {}).get("f16", {}).get("value") if args.raft_dtype != "f16" else fps,
            "data": "synthetic (seeded clip + rectangular mask dilated x4, seeded weights of the reference architecture)",
            "config": {"workload": work, "height": H, "width": W, "frames": L, "windows": len(sched),
                       "raft_dtype": args.raft_dtype, "stages_dtype": "f16" if fp16 else "f32", "parallelism": par,
                       "window_streams": lanes_in_effect[0], "raft_streams": lanes_in_effect[1],
                       "eager_window_streams": args.window_streams, "eager_raft_streams": args.raft_streams},
            "roofline": roof, "cpu_baseline": cpu, "parity_timed_output": parity_timed, "replay_consistency": replay_consistency, "parity": parity,
            "parity_windows_with_reference_frames": parity_refs,
            "fallback": fallback, "configs": configs, "stress": stress, "raft_precisions": raft_precisions,
            "memory": {"peak_allocated_GB_eager_pass": peak_eager / 1e9, "peak_reserved_GB_eager_pass": peak_eager_reserved / 1e9,
                       "peak_allocated_GB_process": peak_total / 1e9, "peak_reserved_GB_process": rt.peak_reserved(dev) / 1e9,
                       "note": "allocated = live tensors (torch.cuda.max_memory_allocated), reserved = what the caching allocator holds "
                               "from the driver over all stream pools (max_memory_reserved: the figure a device-memory monitor shows); the "
                               "eager pass is what a one-shot CLI run needs, the process figures add the hipGraphs' private pools; "
                               "reference README.md:192 quotes 25 GB fp16 at 720x1280x80 (it runs RAFT in 4-frame clips; this engine "
                               "batches all pair-directions; since round 4 the default f16x3 RAFT keeps no correlation volume -- only the "
                               "exact-f32 mode materialises fp32 volumes, within a 40 GB budget)"},
            "exchange": exch, "exchange_plan": exch_plan, "stages_ms": stages,
            "submission": submission,
            "eager_ms_per_step": eager_ms, "graph_capture_s": capture_s,
            "step_ms": step_ms, "host_submit_ms": host_submit, "kernels": kernels,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
