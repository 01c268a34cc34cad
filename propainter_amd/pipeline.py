"""Clip-level orchestration of the inference path (what ``inference_propainter.py:298-452`` of the reference
does between "tensors on the device" and "composited uint8 frames"): RAFT in short clips, flow completion and
image propagation in overlapped sub-videos, sliding-window generator calls and the ordered 0.5/0.5 blend.

The chunk boundaries are part of the result (chunked != unchunked), so they follow the reference exactly; they are
also the shard boundaries of the multi-GPU path (``propainter_amd/sharding.py``).  Unlike the reference, the
uint8 composite/blend stays on the GPU (one device->host copy per clip instead of one per window).
"""
from dataclasses import dataclass

import numpy as np
import torch

from . import hip


@dataclass
class InferenceConfig:
    """Mirrors the CLI flags of inference_propainter.py:181-217 that shape the computation."""
    raft_iter: int = 20
    subvideo_length: int = 80
    neighbor_length: int = 10
    ref_stride: int = 10
    fp16: bool = False
    batch_propagation: bool = True       # feature propagation of equal-length generator windows as one batch (InpaintGenerator.propagate_windows)
    # (both lane settings act on EAGER submission; a captured pass -- ClipGraph and the sharded graphs -- is one chain of launches unless
    #  forked_branches=True: a forked hipGraph replayed wrongly a few per cent of the time at config 5, profiles/r6_c5_replays.txt)
    window_streams: int = 2      # engine extension: generator windows in flight on separate HIP streams (bit-identical results;
                                 # measured 1167.7 -> 1102.9 ms per 720p clip with 2, 1114.6 with 3: profiles/r2_window_streams.txt)
    raft_streams: int = 2        # engine extension: RAFT's two encoders, and its pair-directions in this many groups, on separate
                                 # HIP streams (every pair is computed independently: identical flows; measured on one box, f16x3 pass:
                                 # 1 -> 1557 ms / 120 GB peak, 2 -> 1554 / 95, 3 -> 1541 / 87, 4 -> 1545 / 83: profiles/r3k_streams.txt;
                                 # round 4, volume-free correlation: 2 -> 1452.5, 3 -> 1467 / 1471, 4 -> 1454.6: profiles/r4_stream_sweep.txt)


def get_ref_index(mid_neighbor_id, neighbor_ids, length, ref_stride=10, ref_num=-1):
    """Reference frames of a window (inference_propainter.py:159-173)."""
    if ref_num == -1:
        return [i for i in range(0, length, ref_stride) if i not in neighbor_ids]
    ref_index = []
    lo = max(0, mid_neighbor_id - ref_stride * (ref_num // 2))
    hi = min(length, mid_neighbor_id + ref_stride * (ref_num // 2))
    for i in range(lo, hi, ref_stride):
        if i in neighbor_ids:
            continue
        if len(ref_index) > ref_num:      # sic: up to ref_num + 1 references
            break
        ref_index.append(i)
    return ref_index


def raft_clip_length(width):
    """inference_propainter.py:302-309."""
    if width <= 640:
        return 12
    if width <= 720:
        return 8
    if width <= 1280:
        return 4
    return 2


def subvideo_chunks(total, length, pad):
    """[(s, e, pad_s, pad_e)] of the overlapped chunks used for flow completion (pad 5, :341-364) and image
    propagation (pad 10, :373-398): chunk f covers [f, f+length) extended by `pad` on both sides."""
    out = []
    for f in range(0, total, length):
        s = max(0, f - pad)
        e = min(total, f + length + pad)
        out.append((s, e, f - s, e - min(total, f + length)))
    return out


def window_schedule(video_length, neighbor_length, ref_stride, subvideo_length):
    """[(neighbor_ids, ref_ids)] for every generator call (:410-426)."""
    ns = neighbor_length // 2
    ref_num = subvideo_length // ref_stride if video_length > subvideo_length else -1
    sched = []
    for f in range(0, video_length, ns):
        nb = list(range(max(0, f - ns), min(video_length, f + ns + 1)))
        sched.append((nb, get_ref_index(f, nb, video_length, ref_stride, ref_num)))
    return sched


def compute_flows(fix_raft, frames, raft_iter, streams=1):
    """Stage A (:302-330): RAFT (fp32 input like the reference) in clips of raft_clip_length with 1 frame overlap."""
    L = frames.size(1)
    sl = raft_clip_length(frames.size(-1))
    # (engine extension; other RAFT callables keep the reference signature)
    kw = {"streams": streams} if streams > 1 and getattr(fix_raft, "supports_streams", False) else {}
    if L <= sl or getattr(fix_raft, "batch_invariant", False):
        # the reference clips only to bound memory; an engine whose pairs are batch-independent takes the whole clip
        return fix_raft(frames, iters=raft_iter, **kw)
    ff, fb = [], []
    for f in range(0, L, sl):
        e = min(L, f + sl)
        a, b = fix_raft(frames[:, (f if f == 0 else f - 1):e], iters=raft_iter, **kw)
        ff.append(a)
        fb.append(b)
    return torch.cat(ff, 1), torch.cat(fb, 1)


def complete_flows(fix_flow_complete, gt_flows_bi, flow_masks, subvideo_length):
    """Stage B (:341-368)."""
    fl = gt_flows_bi[0].size(1)
    if fl <= subvideo_length:
        pred, _ = fix_flow_complete.forward_bidirect_flow(gt_flows_bi, flow_masks)
        return fix_flow_complete.combine_flow(gt_flows_bi, pred, flow_masks)
    pf, pb = [], []
    for s, e, ps, pe in subvideo_chunks(fl, subvideo_length, 5):
        sub = (gt_flows_bi[0][:, s:e], gt_flows_bi[1][:, s:e])
        pred, _ = fix_flow_complete.forward_bidirect_flow(sub, flow_masks[:, s:e + 1])
        pred = fix_flow_complete.combine_flow(sub, pred, flow_masks[:, s:e + 1])
        pf.append(pred[0][:, ps:e - s - pe])
        pb.append(pred[1][:, ps:e - s - pe])
    return torch.cat(pf, 1), torch.cat(pb, 1)


def propagate_images(model, frames, masks_dilated, pred_flows_bi, subvideo_length):
    """Stage C (:372-404). Returns (updated_frames [1,L,3,H,W], updated_masks [1,L,1,H,W])."""
    L = frames.size(1)
    masked = frames * (1 - masks_dilated)
    sv = min(100, subvideo_length)
    if L <= sv:
        prop, upd_m = model.img_propagation(masked, pred_flows_bi, masks_dilated, 'nearest')
        return frames * (1 - masks_dilated) + prop * masks_dilated, upd_m
    uf, um = [], []
    for s, e, ps, pe in subvideo_chunks(L, sv, 10):
        sub_flows = (pred_flows_bi[0][:, s:e - 1], pred_flows_bi[1][:, s:e - 1])
        prop, upd_m = model.img_propagation(masked[:, s:e], sub_flows, masks_dilated[:, s:e], 'nearest')
        upd_f = frames[:, s:e] * (1 - masks_dilated[:, s:e]) + prop * masks_dilated[:, s:e]
        uf.append(upd_f[:, ps:e - s - pe])
        um.append(upd_m[:, ps:e - s - pe])
    return torch.cat(uf, 1), torch.cat(um, 1)


class Compositor:
    """Ordered uint8 composite + 0.5/0.5 blend of overlapping windows (:435-450), on the device.
    ``(pred+1)/2*255`` truncated to uint8, pasted inside the dilated mask over the original frame; a frame already
    produced by an earlier window becomes ``uint8(0.5*old + 0.5*new)`` — order dependent, so windows must be
    fed in increasing f."""

    def __init__(self, frames_u8, masks_dilated, float_blend=False):
        """float_blend: the EVALUATION script's composite (scripts/evaluate_propainter.py:170-178 of the reference): a frame covered by
        several windows is averaged in float32 WITHOUT truncating back to uint8 after every blend; ``comp`` is then float32."""
        self.ori = frames_u8                                          # uint8 [L,H,W,3] on device
        self.bin = masks_dilated[0].permute(0, 2, 3, 1).to(torch.uint8)  # [L,H,W,1] {0,1}
        self.float_blend = bool(float_blend)
        self.comp = torch.zeros(frames_u8.shape, dtype=torch.float32 if float_blend else torch.uint8, device=frames_u8.device)
        self.done = [False] * frames_u8.shape[0]

    def add(self, neighbor_ids, pred_img):
        """pred_img [l_t,3,H,W] in [-1,1], fp32 or fp16.  The scaling runs in the prediction's own dtype, one rounding
        per operation, exactly like the reference (``(pred_img + 1) / 2`` on the device, ``* 255`` on a float16 / float32
        numpy array, truncation to uint8: inference_propainter.py:437-438,443) -- so the bytes equal the reference's
        for identical predictions in either precision (tests/test_host_logic_cpu.py::test_compositor_*)."""
        if self.float_blend:
            img = (((pred_img + 1) / 2).permute(0, 2, 3, 1) * 255).to(torch.uint8)
            for i, idx in enumerate(neighbor_ids):
                m = self.bin[idx]
                cur = (img[i] * m + self.ori[idx] * (1 - m)).float()
                self.comp[idx] = self.comp[idx] * 0.5 + cur * 0.5 if self.done[idx] else cur
                self.done[idx] = True
            return
        if pred_img.is_cuda:      # one launch per window (pp_composite_window: same roundings, same bytes) instead of ~8 per frame
            pred = pred_img.contiguous()
            for s0 in range(0, len(neighbor_ids), 32):      # the kernel takes <= 32 frames per launch (--neighbor_length >= 32: groups)
                ids = list(neighbor_ids[s0:s0 + 32])
                hip.composite_window(pred[s0:s0 + 32], self.bin, self.ori, self.comp, ids, [self.done[idx] for idx in ids])
            for idx in neighbor_ids:
                self.done[idx] = True
            return
        img = (pred_img + 1) / 2
        img = img.permute(0, 2, 3, 1) * 255
        img = img.to(torch.uint8)
        for i, idx in enumerate(neighbor_ids):
            m = self.bin[idx]
            cur = img[i] * m + self.ori[idx] * (1 - m)
            if self.done[idx]:
                cur = (self.comp[idx].float() * 0.5 + cur.float() * 0.5).to(torch.uint8)
            self.comp[idx] = cur
            self.done[idx] = True


import collections

_index_cache = collections.OrderedDict()      # LRU: (ids, device) -> int64 index tensor
_INDEX_CACHE_MAX = 4096
_index_recorder = None                        # list collecting every index tensor handed out while a ClipGraph is being built


def _window_streams(device, n, key=None):
    """n side streams of `device` (created once, outside any graph capture); ``key``: see hip.side_streams."""
    return hip.side_streams(device, n, key)


def _dev_index(ids, device):
    """int64 index tensor of a Python id list on `device`, cached by value (the window schedule of a clip shape is
    fixed, so steady-state passes create no index tensors; required for hipGraph capture, which forbids the pageable
    host->device copy hidden in ``x[:, list]``)."""
    key = (tuple(ids), str(device))
    t = _index_cache.get(key)
    if t is None:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("window index tensors must be created by an eager pass before graph capture")
        t = torch.tensor(list(ids), dtype=torch.long, device=device)
        _index_cache[key] = t
        # a long-running server sees many clip lengths: least-recently-used entries leave the cache one at a time.  A captured
        # hipGraph has the device pointers of the tensors it used baked in, so every ClipGraph keeps its own references
        # (`_index_recorder`): eviction only drops the cache's reference, the memory a graph replays from is never freed
        while len(_index_cache) > _INDEX_CACHE_MAX:
            _index_cache.popitem(last=False)
    else:
        _index_cache.move_to_end(key)
    if _index_recorder is not None:
        _index_recorder.append(t)
    return t


@torch.no_grad()
def run_clip(models, frames_u8, flow_masks_u8, masks_dilated_u8, cfg: InferenceConfig, device, return_stages=False,
             stage_hook=None, gt_flows=None, float_blend=False):
    """Whole path for one clip.  frames_u8 [L,H,W,3] uint8, masks [L,H,W] uint8 {0,255} (numpy or tensors).
    models = (RAFT_bi, RecurrentFlowCompleteNet, InpaintGenerator).  Returns uint8 tensor [L,H,W,3] on `device`.
    ``stage_hook(name)`` (optional) is called at every stage boundary (bench.py records HIP events there).
    ``gt_flows`` (optional): precomputed RAFT flows ``(flows_f, flows_b)`` each [L-1,2,H,W] (torch / numpy, e.g. read back
    from the reference's ``.flo`` cache by ``flow_io.load_clip_flows``) -- stage A is skipped."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is not None and device.index != torch.cuda.current_device():
        with torch.cuda.device(device):      # launches bind to the current device: make the clip's device current
            return run_clip(models, frames_u8, flow_masks_u8, masks_dilated_u8, cfg, device, return_stages, stage_hook, gt_flows, float_blend)
    mark = stage_hook or (lambda name: None)
    fix_raft, fix_flow_complete, model = models
    to_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    fr_u8 = to_t(frames_u8).to(device)
    frames = fr_u8.permute(0, 3, 1, 2).float().div(255)[None] * 2 - 1          # [1,L,3,H,W] in [-1,1] (:264)
    flow_masks = to_t(flow_masks_u8).to(device).float().div(255)[None, :, None]
    masks_dilated = to_t(masks_dilated_u8).to(device).float().div(255)[None, :, None]
    L = frames.size(1)
    mark('start')
    if gt_flows is None:
        gt_flows_bi = compute_flows(fix_raft, frames, cfg.raft_iter, streams=cfg.raft_streams)
    else:
        gt_flows_bi = tuple(to_t(f).to(device).float()[None] for f in gt_flows)
        if any(f.shape != (1, L - 1, 2, frames.size(3), frames.size(4)) for f in gt_flows_bi):
            raise ValueError(f"precomputed flows must be [L-1, 2, H, W] = [{L - 1}, 2, {frames.size(3)}, {frames.size(4)}]; "
                             f"got {[tuple(f.shape[1:]) for f in gt_flows_bi]}")
    mark('raft')
    if cfg.fp16:                                                               # (:333-337)
        frames, flow_masks, masks_dilated = frames.half(), flow_masks.half(), masks_dilated.half()
        gt_flows_bi = (gt_flows_bi[0].half(), gt_flows_bi[1].half())
    pred_flows_bi = complete_flows(fix_flow_complete, gt_flows_bi, flow_masks, cfg.subvideo_length)
    mark('flow_completion')
    updated_frames, updated_masks = propagate_images(model, frames, masks_dilated, pred_flows_bi, cfg.subvideo_length)
    mark('image_propagation')
    comp = Compositor(fr_u8, masks_dilated, float_blend=float_blend)      # (float_blend: the evaluation script's float32 averaging)
    # engine extension: everything of the generator windows that depends on a frame / flow pair only (encoder features, 1/4-resolution
    # flows and masks, propagation side inputs) once per clip; a window then reads slices of it -- same results
    clip_cache = model.prepare_clip(updated_frames, pred_flows_bi, masks_dilated, updated_masks) if hasattr(model, "prepare_clip") else None
    enc_all = (model.encode_frames(updated_frames, masks_dilated, updated_masks)
               if clip_cache is None and hasattr(model, "encode_frames") else None)
    empty_ref = torch.zeros((0,), dtype=torch.long, device=device)

    def window(nb, ref):
        if clip_cache is not None:                # (nb is a contiguous range of clip frames)
            return model.forward_window(clip_cache, nb[0], len(nb), _dev_index(ref, device) if ref else empty_ref)
        ids = _dev_index(nb + ref, device)        # cached device index: no host->device copy per window (graph-safe)
        kw = {} if enc_all is None else {"enc_feat": enc_all.index_select(0, ids)}
        fl = slice(nb[0], nb[-1])                 # flows of the local pairs (nb is a contiguous range)
        return model(updated_frames.index_select(1, ids), (pred_flows_bi[0][:, fl], pred_flows_bi[1][:, fl]),
                     masks_dilated.index_select(1, ids), updated_masks.index_select(1, ids), len(nb), **kw)

    sched = window_schedule(L, cfg.neighbor_length, cfg.ref_stride, cfg.subvideo_length)
    if clip_cache is not None and cfg.batch_propagation and hasattr(model, "propagate_windows"):
        # engine extension: the windows' feature propagation up front, windows of equal length batched -- one chain of launches over
        # ~14 frames each instead of 14 chains over one frame each (single-generation grids); same results
        model.propagate_windows(clip_cache, [(nb[0], len(nb)) for nb, _ in sched])
    # (rolling batched propagation: a group of windows is propagated on THIS stream right before its first window is issued, and its
    #  fused tensor is dropped once its last window has been composited -- InpaintGenerator.ensure_propagated / release_window)
    rolling = clip_cache is not None and "prop_plan" in clip_cache
    ensure = (lambda nb: model.ensure_propagated(clip_cache, nb[0], len(nb))) if rolling else (lambda nb: None)
    release = (lambda nb: model.release_window(clip_cache, nb[0], len(nb))) if rolling else (lambda nb: None)
    lanes = _window_streams(device, cfg.window_streams) if device.type == "cuda" else []
    if len(lanes) < 2:
        for nb, ref in sched:
            ensure(nb)
            comp.add(nb, window(nb, ref)[0])
            release(nb)
    else:
        # The windows are independent until the ordered blend: consecutive windows run on separate HIP streams (forked from
        # and joined to the current stream, so a hipGraph capture records parallel branches), which lets one window's
        # HBM-bound kernels (LayerNorm, fold, pooling, warps) and partial-wave launches overlap another window's MFMA-bound
        # ones.  The same kernels run on the same data: results are bit-identical to the serial order; the composites are
        # blended afterwards in increasing window position.
        cur = torch.cuda.current_stream(device)
        for i in range(0, len(sched), len(lanes)):
            group = []
            for nb, ref in sched[i:i + len(lanes)]:
                ensure(nb)                        # on the pass's stream: the lanes fork behind it
            for s_, (nb, ref) in zip(lanes, sched[i:i + len(lanes)]):
                s_.wait_stream(cur)
                with torch.cuda.stream(s_):
                    group.append((nb, window(nb, ref), s_))
            for nb, pred, s_ in group:
                cur.wait_stream(s_)
                pred.record_stream(cur)
                comp.add(nb, pred[0])
            for nb, _, _ in group:
                release(nb)                       # every lane of the group has joined: a finished propagation group can go
    mark('generator')
    if return_stages:
        return comp.comp, dict(gt_flows=gt_flows_bi, pred_flows=pred_flows_bi, updated_frames=updated_frames,
                               updated_masks=updated_masks)
    return comp.comp


class ClipGraph:
    """The whole clip pass (stages A-D + composite) of one clip shape captured in ONE hipGraph.

    The eager path issues ~12 000 launches per 720p clip from Python, and the recurrent stages (flow-completion and
    feature propagation: hundreds of dependent steps of 20-200 us kernels) are bound by that host work, not by the GPU.
    A clip pass has no data-dependent host control flow (the masked-window split is read on the device, the window
    schedule is a function of the clip length), so for a serving loop over clips of one shape the pass is captured once
    (``torch.cuda.graph`` records the raw ``hipLaunchKernelGGL`` calls of libpropainter_hip, which are issued on torch's
    current -- capturing -- stream) and replayed: the kernels and their order are identical to the eager pass, the
    activations live in the graph's private pool (a few tens of GB of the 288 GB HBM), inputs are copied into static
    buffers.  Results are bit-identical to ``run_clip`` (same kernels, same order; tested on the GPU).

        g = ClipGraph(models, L, H, W, cfg, device)       # one eager pass (tables, engines) + capture
        out_u8 = g(frames_u8, flow_masks_u8, masks_dilated_u8)   # uint8 [L,H,W,3] on device (static buffer)
    """

    def __init__(self, models, L, H, W, cfg: InferenceConfig, device, example=None, release_eager_pool=False, forked_branches=False):
        # forked_branches=False (default): the captured pass is ONE chain of launches -- no window lanes, no RAFT lanes.  Results are the
        # same bytes either way (the lanes only reorder independent work), but on ROCm 7.2 a hipGraph with forked branches does not
        # replay this pass reliably at every size: BASELINE config 5 (1080x1920x160, subvideo_length 20) with the 2 + 2 lanes left wrong
        # bytes (max |d| 25) in 3 of 64 replays, the same graph without forked branches in 0 of 84; the 720x1280x80 pass with lanes
        # 0 of ~250 (profiles/r6_c5_replays.txt).  The eager pass keeps cfg's lanes (30 eager passes with lanes: 0 deviations).
        # forked_branches=True keeps cfg.window_streams / cfg.raft_streams inside the capture (-2 ... -3 % per pass): run self_check().
        import dataclasses
        if not forked_branches:
            cfg = dataclasses.replace(cfg, window_streams=1, raft_streams=1)
        self.cfg = cfg
        self.shape = (L, H, W)
        self.frames = torch.zeros((L, H, W, 3), dtype=torch.uint8, device=device)
        self.flow_masks = torch.zeros((L, H, W), dtype=torch.uint8, device=device)
        self.masks_dilated = torch.zeros((L, H, W), dtype=torch.uint8, device=device)
        if example is not None:
            self._load(*example)
        run = lambda: run_clip(models, self.frames, self.flow_masks, self.masks_dilated, cfg, device)
        global _index_recorder
        self._pinned = []                        # every cached index tensor the pass reads: the graph owns a reference (see _dev_index)
        prev, _index_recorder = _index_recorder, self._pinned
        try:
            self._build(run, device, release_eager_pool)
        finally:
            _index_recorder = prev

    def _build(self, run, device, release_eager_pool):
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):            # eager pass: builds engines, window tables, index tensors
            run()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        if release_eager_pool:                   # hand the eager pass's cached blocks back before the graph pool grows
            torch.cuda.empty_cache()             # (default: keep them -- 288 GB holds both pools and later eager passes stay warm)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: helper threads of the process (e.g. the RCCL watchdog polling its events in a multi-GPU job) must
        # not invalidate the capture; this thread itself issues nothing but kernel launches and pool allocations
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = run()

    def _load(self, frames_u8, flow_masks_u8, masks_dilated_u8):
        to_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
        self.frames.copy_(to_t(frames_u8), non_blocking=True)
        self.flow_masks.copy_(to_t(flow_masks_u8), non_blocking=True)
        self.masks_dilated.copy_(to_t(masks_dilated_u8), non_blocking=True)

    def replay(self):
        """Re-runs the pass on whatever the static input buffers hold."""
        self.graph.replay()
        return self.out

    def self_check(self, replays=16):
        """Replays the pass ``replays`` + 1 times on the loaded inputs and compares every replay byte for byte with the first:
        -> (replays that differ, max |d|).  A captured pass is deterministic (the reference's loops are: inference_propainter.py:342-452);
        a replay that differs is a defect of the submission however rare -- rounds 2-5 shipped a graph of which ~5 % of the replays had a
        few hundred wrong bytes (profiles/r6_replay_bytes.txt), which the 1-3 replays the tests compared could not see.  bench.py prints
        this as `replay_consistency`; callers that capture at their own sizes should run it once."""
        first = self.replay().clone()
        differing, worst = 0, 0
        for _ in range(int(replays)):
            o = self.replay()
            if not torch.equal(o, first):
                differing += 1
                worst = max(worst, int((o.to(torch.int16) - first.to(torch.int16)).abs().max()))
        torch.cuda.synchronize(first.device)
        return differing, worst

    def __call__(self, frames_u8, flow_masks_u8, masks_dilated_u8):
        self._load(frames_u8, flow_masks_u8, masks_dilated_u8)
        return self.replay()
