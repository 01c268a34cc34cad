"""Round-5 diagnostics of the streaming schedule's stale-read failure (profiles/r4_streaming_race.txt): first-pass statistics after
loading a new clip, per schedule form, in ONE process so that the forms see the same box.

    python tools/diag_stream2.py N mode [mode ...]
modes:  concurrent       the ranks' segment graphs overlap on their streams (round 4's failing form)
        concurrent_d2d   the same, but the inputs reach the static buffers through a device-side copy kernel from a freshly
                         allocated upload tensor (tests "the graphs do not see the copy engine's writes")
        concurrent_fence the same as `concurrent`, with an event recorded on the upload stream after load() and waited for on every
                         rank stream (the verdict's recipe; replay() already does st.wait_stream(cur))
        chained          wavefront issue order, every launch waits for the previous one (round 4's default)
        single           the whole wavefront captured as ONE hipGraph (StreamingClipGraph(single_graph=True))
Environment variants of the HIP runtime (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, GPU_MAX_HW_QUEUES=8, ...) are set by the caller.
"""
import os
import sys
import time

import numpy as np
import scipy.ndimage
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd.pipeline import InferenceConfig, run_clip          # noqa: E402
from propainter_amd.sharding import StreamingClipGraph                 # noqa: E402
from propainter_amd.synthetic import seeded_models, synthetic_clip, synthetic_mask   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
modes = sys.argv[2:] or ["concurrent", "single"]
models = seeded_models("cuda")
L, H, W = 34, 128, 192
m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.uint8) * 255
masks = np.repeat(m[None], L, 0)
clips = [synthetic_clip(L, H, W, seed=12 + i) for i in range(3)]
dev = torch.device("cuda")
cfg = InferenceConfig(raft_iter=3, subvideo_length=10, neighbor_length=4, ref_stride=3, fp16=True, batch_propagation=False,
                      window_streams=int(os.environ.get("PP_DIAG_LANES", "2")))
models[0].precision = os.environ.get("PP_DIAG_RAFT", "f16x3")
refs = [run_clip(models, c, masks, masks, cfg, dev).clone() for c in clips]
env = {k: v for k, v in os.environ.items() if k.startswith(("DEBUG_", "GPU_MAX", "HIP_FORCE", "GPU_FLUSH", "ROC_"))}
print(f"STREAM_DIAG2 env {env}", flush=True)

built = {}


def graph_for(single, private=False):
    key = (single, private)
    if key not in built:
        sc = StreamingClipGraph(models, L, H, W, cfg, dev, share_pool=False, single_graph=single)
        if private:
            # every logical rank gets its OWN module objects (own engines, packed weights, tables): if the overlap failure needs state shared
            # through the modules, it disappears here
            from propainter_amd.sharding import run_logical_shards
            for g in sc.graphs:
                mr = seeded_models("cuda")
                mr[0].precision = "f16x3"
                run_logical_shards(mr, clips[0], masks, masks, sc.cfg, dev, sc.world)      # eager warm-up: engines, tables
                g.models = mr
            torch.cuda.synchronize()
        sc.load(clips[0], masks, masks)
        sc.capture()
        built[key] = sc
    return built[key]


def load_d2d(sc, frames, fm, md):
    """upload into FRESH device tensors, then device-side copy kernels into the static buffers the graphs read"""
    for g in sc.graphs:
        for dst, src in ((g.frames, frames), (g.flow_masks, fm), (g.masks_dilated, md)):
            if dst.shape[0]:
                tmp = torch.from_numpy(np.ascontiguousarray(src[g.r0:g.r1])).to(dev)
                dst.copy_(tmp)


for mode in modes:
    sc = graph_for(mode.startswith("single"), private=mode.endswith("private_models"))
    bad, t_ms, bad2 = [], [], 0
    for it in range(N):
        k = (it + 1) % 3
        if mode == "concurrent_d2d":
            load_d2d(sc, clips[k], masks, masks)
        else:
            sc.load(clips[k], masks, masks)
        if mode == "concurrent_fence":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            for st in sc.streams:
                st.wait_event(ev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = sc.replay(concurrent=mode.startswith("concurrent")) if not mode.startswith("single") else sc.replay()
        torch.cuda.synchronize()
        t_ms.append((time.perf_counter() - t0) * 1e3)
        ne = out != refs[k]
        if ne.any():
            bad.append((it, [i for i in range(L) if ne[i].any()][:3]))
            out2 = sc.replay(concurrent=mode.startswith("concurrent")) if not mode.startswith("single") else sc.replay()
            torch.cuda.synchronize()
            bad2 += int((out2 != refs[k]).any())
    if mode == "single" and os.environ.get("PP_SG_DEBUG") == "1":
        # which stage output deviates first?  eager stage tensors of the last clip against the graph's named locals
        from propainter_amd.sharding import ShardPlan
        _, st = run_clip(models, clips[k], masks, masks, cfg, dev, return_stages=True)
        plan = ShardPlan(L, sc.cfg, sc.world)
        for r in range(sc.world):
            lo, hi = plan.flows_own(r)
            d0 = sc._single_debug[(r, 0)]
            print(f"STREAM_DIAG2   rank {r} segment-0 locals: {sorted(d0)}", flush=True)
            if "ff" not in d0:
                gtv = d0["gt"].float()
                d0 = dict(d0, ff=gtv[0], fb=gtv[1])
            for name, ref in (("ff", st["gt_flows"][0][:, lo:hi]), ("fb", st["gt_flows"][1][:, lo:hi])):
                got = d0[name].float()
                print(f"STREAM_DIAG2   rank {r} RAFT {name}: max |d| vs eager {float((got - ref.float()).abs().max()):.3e} (shape {tuple(got.shape)})", flush=True)
            r0, r1 = plan.need_raw(r)
            fr_ref = torch.from_numpy(clips[k][r0:r1]).to(dev)
            print(f"STREAM_DIAG2   rank {r} fr_u8 equal {bool(torch.equal(d0['fr_u8'], fr_ref))}", flush=True)
            if (r, 1) in sc._single_debug and "pred_own" in sc._single_debug[(r, 1)]:
                po = sc._single_debug[(r, 1)]["pred_own"].float()
                lo2, hi2 = plan.flows_own(r)
                ref = torch.stack([st["pred_flows"][0][:, lo2:hi2], st["pred_flows"][1][:, lo2:hi2]], 0).float()
                print(f"STREAM_DIAG2   rank {r} completed flows: max |d| vs eager {float((po - ref).abs().max()):.3e}", flush=True)
    print(f"STREAM_DIAG2 {mode}: {len(bad)} of {N} first passes wrong {bad[:4]} (a second pass over the same inputs still wrong: {bad2} of {len(bad)}); "
          f"median pass {sorted(t_ms)[len(t_ms) // 2]:.1f} ms", flush=True)
