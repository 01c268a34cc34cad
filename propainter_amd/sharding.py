"""Sub-video sharding of one long clip across the GPUs of a node.

The reference processes long clips as overlapping sub-videos whose boundaries are fixed by ``--subvideo_length``
from frame 0 (inference_propainter.py:341-404) and as sliding generator windows (:410-452).  Those chunk boundaries
are part of the result, so the shard boundaries coincide with them: rank r owns the frames
``[r*B, (r+1)*B)`` with ``B = subvideo_length * ceil(n_subvideos / world)`` and executes exactly the chunks / windows
of the global schedule that start inside its range.  What a chunk needs from outside the owner's range (the +-5 flow
halo of flow completion, the +-10 frame halo of image propagation, neighbour / reference frames of the generator
windows, the blend contributions of windows straddling a boundary) is exchanged point-to-point with the owning rank --
``torch.distributed`` send/recv (RCCL over xGMI on the GPU box, gloo in the CPU tests); there is no all-reduce /
all-gather on the data path.  The composited frames are bit-identical to the single-GPU pass.

``sharded_clip_steps`` is written as a *generator* that yields ``Exchange`` requests: ``run_clip_sharded`` drives it
with real point-to-point transfers (one process per GPU), ``run_logical_shards`` drives N logical ranks in one
process by handing the tensors over directly (single-GPU validation of the sharding logic against the unsharded pass).
"""
from dataclasses import dataclass, field

import os

import numpy as np
import torch

from . import hip
from .pipeline import (InferenceConfig, _dev_index, _window_streams, compute_flows, subvideo_chunks, window_schedule)


# ----------------------------------------------------------------------------------------------------------------
# plan
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class ShardPlan:
    """Frame ownership and per-stage needs of every rank (pure host arithmetic; identical on all ranks)."""
    L: int
    cfg: InferenceConfig
    world: int
    block: int = 0
    own: list = field(default_factory=list)            # [(lo, hi)] frames per rank (may be empty: lo == hi)

    def __post_init__(self):
        S = self.cfg.subvideo_length
        if self.world > 1:
            if S > 100:
                raise ValueError("sharded inference needs subvideo_length <= 100 (image propagation uses min(100, S) chunks, "
                                 "inference_propainter.py:372)")
            if self.L - 1 <= S:
                raise ValueError(f"a {self.L}-frame clip is a single sub-video at subvideo_length={S}: nothing to shard "
                                 "(use one GPU per clip instead)")
        nsub = -(-self.L // S)
        self.block = S * (-(-nsub // self.world))
        self.own = [(min(self.L, r * self.block), min(self.L, (r + 1) * self.block)) for r in range(self.world)]
        self.fl = self.L - 1
        ns = self.cfg.neighbor_length // 2
        self.windows = []                                # (f, neighbor_ids, ref_ids) of the global schedule
        sched = window_schedule(self.L, self.cfg.neighbor_length, self.cfg.ref_stride, S)
        for f, (nb, ref) in zip(range(0, self.L, ns), sched):
            self.windows.append((f, nb, ref))

    def owner(self, frame):
        return min(self.world - 1, frame // self.block)

    # ---- index ranges (half-open, clipped); an empty range is (0, 0)
    def flows_own(self, r):
        lo, hi = self.own[r]
        return (lo, min(hi, self.fl)) if lo < min(hi, self.fl) else (0, 0)

    def fc_chunks(self, r):
        """flow-completion chunks (s, e, pad_s, pad_e) owned by rank r (flow index space, pad 5; :341-364)."""
        return [c for c in subvideo_chunks(self.fl, self.cfg.subvideo_length, 5) if self.owner(c[0] + c[2]) == r]

    def ip_chunks(self, r):
        """image-propagation chunks owned by rank r (frame index space, pad 10; :373-398)."""
        sv = min(100, self.cfg.subvideo_length)
        return [c for c in subvideo_chunks(self.L, sv, 10) if self.owner(c[0] + c[2]) == r]

    def rank_windows(self, r):
        return [w for w in self.windows if self.owner(w[0]) == r]

    @staticmethod
    def _env(ranges):
        ranges = [x for x in ranges if x[1] > x[0]]
        return (min(a for a, _ in ranges), max(b for _, b in ranges)) if ranges else (0, 0)

    def need_gt_flows(self, r):
        return self._env([(s, e) for s, e, _, _ in self.fc_chunks(r)])

    def need_pred_flows(self, r):
        rs = [(s, e - 1) for s, e, _, _ in self.ip_chunks(r)]
        rs += [(nb[0], nb[-1]) for _, nb, _ in self.rank_windows(r)]
        return self._env(rs)

    def need_updated(self, r):
        return self._env([(min(nb + ref), max(nb + ref) + 1) for _, nb, ref in self.rank_windows(r)])

    def need_raw(self, r):
        """frames / masks every rank reads from the (host-resident) input: envelope of all of its stage inputs."""
        lo, hi = self.own[r]
        if hi <= lo:
            return (0, 0)
        rs = [(lo, min(self.L, hi + 1))]                                                   # RAFT: own pairs + 1 frame
        rs += [(s, e + 1) for s, e, _, _ in self.fc_chunks(r)]                             # flow masks of the FC chunks
        rs += [(s, e) for s, e, _, _ in self.ip_chunks(r)]
        rs.append(self.need_updated(r))
        return self._env(rs)

    def blend_routes(self):
        """{(src, dst): [(frame, f)]}: composites a rank's windows produce for frames another rank owns, in the
        order both sides enumerate them (windows by f, frames ascending)."""
        routes = {}
        for f, nb, _ in self.windows:
            src = self.owner(f)
            for idx in nb:
                dst = self.owner(idx)
                if dst != src:
                    routes.setdefault((src, dst), []).append((idx, f))
        return routes


def plan_exchange_bytes(L, cfg, world, H, W, elem_bytes):
    """Bytes every rank sends / receives per exchange stage of a sharded pass, from the plan's index ranges alone (no tensors):
    [{tag: (sent, received)}] per rank for the four exchanges of sharded_clip_steps -- RAFT flows (+-5 pairs), completed flows (+-10),
    updated frames + masks of the neighbour / reference frames (4 channels), uint8 composites of the windows that straddle a boundary
    (SURVEY.md section 8(e)).  elem_bytes = 2 (fp16 stages) or 4.  tests/test_sharding_cpu.py checks real gloo runs against it."""
    plan = ShardPlan(L, cfg, world)
    ov = lambda a, b: max(0, min(a[1], b[1]) - max(a[0], b[0]))
    fo = [plan.flows_own(r) for r in range(world)]
    routes = plan.blend_routes()
    out = []
    for r in range(world):
        rec = {}
        for tag, own, need, per_item in (("gt_flows", fo, plan.need_gt_flows, 2 * 2 * H * W * elem_bytes),
                                         ("pred_flows", fo, plan.need_pred_flows, 2 * 2 * H * W * elem_bytes),
                                         ("updated_frames", plan.own, plan.need_updated, 4 * H * W * elem_bytes)):
            rec[tag] = (sum(ov(own[r], need(q)) for q in range(world) if q != r) * per_item,
                        sum(ov(own[q], need(r)) for q in range(world) if q != r) * per_item)
        rec["blend"] = (sum(len(v) for (s_, d), v in routes.items() if s_ == r) * H * W * 3,
                        sum(len(v) for (s_, d), v in routes.items() if d == r) * H * W * 3)
        out.append(rec)
    return out


# ----------------------------------------------------------------------------------------------------------------
# exchange plumbing
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class Exchange:
    """One point-to-point exchange step: ``send[q]`` goes to rank q; ``recv[q] = (shape, dtype)`` arrives from rank q.
    The driver answers with {q: tensor}."""
    send: dict
    recv: dict
    tag: str = ""


class Span:
    """A tensor whose dimension `dim` covers the global index range [g0, g0 + size)."""

    def __init__(self, t, g0, dim=1):
        self.t, self.g0, self.dim = t, g0, dim

    def sl(self, a, b):
        return self.t.narrow(self.dim, a - self.g0, b - a)

    def take(self, ids, device):
        return self.t.index_select(self.dim, _dev_index([i - self.g0 for i in ids], device))


def _isect(a, b):
    lo, hi = max(a[0], b[0]), min(a[1], b[1])
    return (lo, hi) if hi > lo else None


def _halo(x, own, needs, rank, dim, tag):
    """Sub-generator: `x` covers own[rank] along `dim`; returns a Span covering needs[rank].  Every rank sends the part
    of its own range that another rank needs and receives the parts of its need that others own."""
    world = len(own)
    send, recv = {}, {}
    for q in range(world):
        if q == rank:
            continue
        s = _isect(own[rank], needs[q])
        if s is not None:
            send[q] = x.narrow(dim, s[0] - own[rank][0], s[1] - s[0]).contiguous()
        r = _isect(own[q], needs[rank])
        if r is not None:
            shape = list(x.shape)
            shape[dim] = r[1] - r[0]
            recv[q] = (tuple(shape), x.dtype)
    got = yield Exchange(send, recv, tag)
    need = needs[rank]
    if need[1] <= need[0]:
        return Span(x.narrow(dim, 0, 0), 0, dim)
    pieces = []
    for q in range(world):
        r = _isect(own[q], need)
        if r is None:
            continue
        pieces.append(x.narrow(dim, r[0] - own[rank][0], r[1] - r[0]) if q == rank else got[q])
    return Span(pieces[0] if len(pieces) == 1 else torch.cat(pieces, dim), need[0], dim)


# ----------------------------------------------------------------------------------------------------------------
# the sharded pass
# ----------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def sharded_clip_steps(models, frames_u8, flow_masks_u8, masks_dilated_u8, cfg: InferenceConfig, device, rank, world, lane_key=None):
    """Generator: rank `rank`'s part of the pass over one clip.  Inputs are the WHOLE clip (uint8 frames [L,H,W,3],
    masks [L,H,W] {0,255}; numpy or tensors, normally host-resident -- every rank reads its slice, no exchange of raw
    input).  Yields ``Exchange`` requests; returns ``(lo, comp_u8[hi-lo,H,W,3])``, the rank's composited frames.
    ``lane_key``: private generator lanes for this rank (hip.side_streams; several logical ranks captured into one hipGraph)."""
    fix_raft, fix_flow_complete, model = models
    L = len(frames_u8)
    plan = ShardPlan(L, cfg, world)
    lo, hi = plan.own[rank]
    active = hi > lo
    ranks = range(world)
    to_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    H, W = frames_u8.shape[1], frames_u8.shape[2]
    dt_stage = torch.float16 if cfg.fp16 else torch.float32

    # ---- raw inputs of this rank (global range [r0, r1))
    r0, r1 = plan.need_raw(rank)
    fr_u8 = to_t(frames_u8[r0:r1]).to(device)
    frames = Span(fr_u8.permute(0, 3, 1, 2).float().div(255)[None] * 2 - 1, r0)             # [1,n,3,H,W] in [-1,1]
    flow_masks = Span(to_t(flow_masks_u8[r0:r1]).to(device).float().div(255)[None, :, None], r0)
    masks_dilated = Span(to_t(masks_dilated_u8[r0:r1]).to(device).float().div(255)[None, :, None], r0)

    # ---- stage A: RAFT on the own pairs (needs frame `hi` of the right neighbour: raw input)
    fo = [plan.flows_own(q) for q in ranks]
    if fo[rank][1] > fo[rank][0]:
        ff, fb = compute_flows(fix_raft, frames.sl(fo[rank][0], fo[rank][1] + 1), cfg.raft_iter, streams=cfg.raft_streams)
        gt = torch.stack([ff, fb], 0)                                                        # [2,1,n,2,H,W]
    else:
        gt = torch.zeros((2, 1, 0, 2, H, W), dtype=torch.float32, device=device)
    if cfg.fp16:                                                                             # (:333-337)
        frames.t, flow_masks.t, masks_dilated.t = frames.t.half(), flow_masks.t.half(), masks_dilated.t.half()
        gt = gt.half()
    gt = yield from _halo(gt, fo, [plan.need_gt_flows(q) for q in ranks], rank, 2, "gt_flows")

    # ---- stage B: flow completion on the own chunks (+-5 flow halo)
    pf, pb = [], []
    for s, e, ps, pe in plan.fc_chunks(rank):
        sub = (gt.sl(s, e)[0], gt.sl(s, e)[1])
        fm = flow_masks.sl(s, e + 1)
        pred, _ = fix_flow_complete.forward_bidirect_flow(sub, fm)
        pred = fix_flow_complete.combine_flow(sub, pred, fm)
        pf.append(pred[0][:, ps:e - s - pe])
        pb.append(pred[1][:, ps:e - s - pe])
    if pf:
        pred_own = torch.stack([torch.cat(pf, 1), torch.cat(pb, 1)], 0)
    else:
        pred_own = torch.zeros((2, 1, 0, 2, H, W), dtype=dt_stage, device=device)
    pred = yield from _halo(pred_own, fo, [plan.need_pred_flows(q) for q in ranks], rank, 2, "pred_flows")

    # ---- stage C: image propagation on the own chunks (+-10 frame halo of completed flows; frames/masks are raw)
    uf, um = [], []
    for s, e, ps, pe in plan.ip_chunks(rank):
        fr, md = frames.sl(s, e), masks_dilated.sl(s, e)
        sub_flows = (pred.sl(s, e - 1)[0], pred.sl(s, e - 1)[1])
        prop, upd_m = model.img_propagation(fr * (1 - md), sub_flows, md, 'nearest')
        upd_f = fr * (1 - md) + prop * md
        uf.append(upd_f[:, ps:e - s - pe])
        um.append(upd_m[:, ps:e - s - pe])
    if uf:
        upd_own = torch.cat([torch.cat(uf, 1), torch.cat(um, 1)], 2)                         # [1,n,4,H,W]: frame | mask
    else:
        upd_own = torch.zeros((1, 0, 4, H, W), dtype=dt_stage, device=device)
    upd = yield from _halo(upd_own, plan.own, [plan.need_updated(q) for q in ranks], rank, 1, "updated_frames")

    # ---- stage D: the own generator windows; composites of frames owned elsewhere go to their owner afterwards
    routes = plan.blend_routes()
    foreign_f = {}                                                   # own frame -> window ids f of foreign contributions
    for (src, dst), items in routes.items():
        if dst == rank:
            for idx, f in items:
                foreign_f.setdefault(idx, []).append(f)
    comp = torch.zeros((max(0, hi - lo), H, W, 3), dtype=torch.uint8, device=device)
    done = [False] * max(0, hi - lo)
    deferred = {}                                                    # own frame with foreign contributions -> [(f, cur)]
    outbox = {}                                                      # dst rank -> [cur] in route order
    enc_all, enc_pos, clip_cache = None, {}, None
    my_windows = plan.rank_windows(rank)
    used = sorted({i for _, nb, ref in my_windows for i in nb + ref})      # own frames + the strided references
    enc_pos = {g: k for k, g in enumerate(used)}
    if len(used) >= 2 and hasattr(model, "prepare_clip") and hasattr(model, "forward_window"):
        # The per-clip generator cache of the unsharded pass (pipeline.run_clip), over the COMPACT clip of the frames this rank's
        # windows touch: encoder features, 1/4-resolution flows / masks and propagation side inputs once per frame / pair, the windows'
        # feature propagation batched, the windows on the generator lanes.  The local frames of the rank's windows are one contiguous
        # range [a0, b0) of the clip (windows every neighbor_stride frames, +-neighbor_stride locals), so they are contiguous in the
        # compact clip as well; its other flow pairs (between references) are never read and stay zero.  Same kernels on the same
        # values as the unsharded pass: results are identical.
        a0, b0 = min(nb[0] for _, nb, _ in my_windows), max(nb[-1] for _, nb, _ in my_windows) + 1
        p0 = enc_pos[a0]
        assert used[p0:p0 + b0 - a0] == list(range(a0, b0)), "local frames of a rank's windows: one contiguous range"
        uu = upd.take(used, device)
        cflows = torch.zeros((2, 1, len(used) - 1, 2, H, W), dtype=pred.t.dtype, device=device)
        if b0 - a0 > 1:
            cflows[:, :, p0:p0 + b0 - a0 - 1] = pred.sl(a0, b0 - 1)
        clip_cache = model.prepare_clip(uu[:, :, :3].contiguous(), (cflows[0], cflows[1]), masks_dilated.take(used, device),
                                        uu[:, :, 3:4].contiguous())
        if cfg.batch_propagation and hasattr(model, "propagate_windows"):
            model.propagate_windows(clip_cache, [(enc_pos[nb[0]], len(nb)) for _, nb, _ in my_windows])
    elif my_windows and hasattr(model, "encode_frames"):
        # encoder features once per frame this rank's windows actually touch
        uu = upd.take(used, device)
        enc_all = model.encode_frames(uu[:, :, :3].contiguous(), masks_dilated.take(used, device), uu[:, :, 3:4].contiguous())
    empty_ref = torch.zeros((0,), dtype=torch.long, device=device)

    def blend(idx, cur):
        k = idx - lo
        if done[k]:
            cur = (comp[k].float() * 0.5 + cur.float() * 0.5).to(torch.uint8)
        comp[k] = cur
        done[k] = True

    def window(nb, ref):
        if clip_cache is not None:
            return model.forward_window(clip_cache, enc_pos[nb[0]], len(nb),
                                        _dev_index([enc_pos[i] for i in ref], device) if ref else empty_ref)
        ids = nb + ref
        u = upd.take(ids, device)
        kw = {} if enc_all is None else {"enc_feat": enc_all.index_select(0, _dev_index([enc_pos[i] for i in ids], device))}
        fl = pred.sl(nb[0], nb[-1])
        return model(u[:, :, :3].contiguous(), (fl[0], fl[1]), masks_dilated.take(ids, device), u[:, :, 3:4].contiguous(),
                     len(nb), **kw)

    def composite(f, nb, out):
        img = (((out[0] + 1) / 2).permute(0, 2, 3, 1) * 255).to(torch.uint8)   # (:435-442) in the prediction's dtype, as pipeline.Compositor
        for i, idx in enumerate(nb):
            m = masks_dilated.sl(idx, idx + 1)[0, 0].permute(1, 2, 0).to(torch.uint8)
            ori = fr_u8[idx - r0]
            cur = img[i] * m + ori * (1 - m)
            dst = plan.owner(idx)
            if dst != rank:
                outbox.setdefault(dst, []).append(cur)
            elif idx in foreign_f:
                deferred.setdefault(idx, []).append((f, cur))
            else:
                blend(idx, cur)

    rolling = clip_cache is not None and "prop_plan" in clip_cache      # rolling batched propagation (pipeline.run_clip)
    ensure = (lambda nb: model.ensure_propagated(clip_cache, enc_pos[nb[0]], len(nb))) if rolling else (lambda nb: None)
    release = (lambda nb: model.release_window(clip_cache, enc_pos[nb[0]], len(nb))) if rolling else (lambda nb: None)
    lanes = _window_streams(device, cfg.window_streams, lane_key) if device.type == "cuda" else []
    if len(lanes) < 2:
        for f, nb, ref in my_windows:
            ensure(nb)
            composite(f, nb, window(nb, ref))
            release(nb)
    else:
        # consecutive windows on separate HIP streams, composited afterwards in window order (as pipeline.run_clip)
        cur_s = torch.cuda.current_stream(device)
        for i in range(0, len(my_windows), len(lanes)):
            group = []
            for f, nb, ref in my_windows[i:i + len(lanes)]:
                ensure(nb)
            for s_, (f, nb, ref) in zip(lanes, my_windows[i:i + len(lanes)]):
                s_.wait_stream(cur_s)
                with torch.cuda.stream(s_):
                    group.append((f, nb, window(nb, ref), s_))
            for f, nb, out, s_ in group:
                cur_s.wait_stream(s_)
                out.record_stream(cur_s)
                composite(f, nb, out)
            for f, nb, out, s_ in group:
                release(nb)
    send = {q: torch.stack(v, 0) for q, v in outbox.items()}
    recv = {src: ((len(items), H, W, 3), torch.uint8) for (src, dst), items in routes.items() if dst == rank}
    got = yield Exchange(send, recv, "blend")
    for (src, dst), items in routes.items():
        if dst == rank:
            for k, (idx, f) in enumerate(items):
                deferred.setdefault(idx, []).append((f, got[src][k]))
    for idx in sorted(deferred):                                     # ordered 0.5/0.5 blend: increasing window position f
        for f, cur in sorted(deferred[idx], key=lambda fc: fc[0]):
            blend(idx, cur)
    return lo, comp


# ----------------------------------------------------------------------------------------------------------------
# drivers
# ----------------------------------------------------------------------------------------------------------------
def can_shard(L, cfg, world):
    """True when a clip of L frames splits into sub-videos over `world` > 1 ranks (see ShardPlan); a short clip is a
    single sub-video and runs as one unsharded pass on one GPU instead."""
    return world > 1 and cfg.subvideo_length <= 100 and L - 1 > cfg.subvideo_length


def _dist_exchange(ex, device, group=None, stats=None, bufs=None):
    """Answers one Exchange with torch.distributed point-to-point ops (RCCL send/recv between the two GPUs' xGMI link;
    gloo stages through host memory).  ``stats`` (optional dict) accumulates per exchange tag the bytes sent / received
    by this rank and the wall time of the exchange (device-synchronised on both sides when the backend is RCCL).
    ``bufs`` (optional {q: tensor}): receive into these preallocated device tensors (ShardedClipGraph: the captured graphs read them)."""
    import time
    import torch.distributed as dist
    via_host = dist.get_backend(group) == "gloo"
    if stats is not None:
        if not via_host:
            torch.cuda.synchronize(device)
        t0 = time.perf_counter()
    static, bufs, ops, keep = bufs, {}, [], []
    for q, (shape, dtype) in sorted(ex.recv.items()):
        bufs[q] = static[q] if (static is not None and not via_host) else torch.empty(shape, dtype=dtype, device="cpu" if via_host else device)
        ops.append(dist.P2POp(dist.irecv, bufs[q], q, group))
    for q, t in sorted(ex.send.items()):
        t = t.cpu() if via_host else t.contiguous()
        keep.append(t)
        ops.append(dist.P2POp(dist.isend, t, q, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if static is not None and via_host:
        for q, b in bufs.items():
            static[q].copy_(b)
        bufs = static
    out = {q: b.to(device) for q, b in bufs.items()}
    if stats is not None:
        if not via_host:
            torch.cuda.synchronize(device)
        rec = stats.setdefault(ex.tag, {"sent_bytes": 0, "recv_bytes": 0, "ms": 0.0, "calls": 0})
        rec["sent_bytes"] += sum(t.numel() * t.element_size() for t in keep)
        rec["recv_bytes"] += sum(b.numel() * b.element_size() for b in bufs.values())
        rec["ms"] += (time.perf_counter() - t0) * 1e3
        rec["calls"] += 1
    return out


def run_clip_sharded(models, frames_u8, flow_masks_u8, masks_dilated_u8, cfg, device, group=None, stats=None):
    """One process per GPU: this rank's part of the clip.  Returns (lo, comp_u8) -- frames [lo, lo+len) of the result.
    ``stats``: see ``_dist_exchange``."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    gen = sharded_clip_steps(models, frames_u8, flow_masks_u8, masks_dilated_u8, cfg, device, rank, world)
    try:
        ex = next(gen)
        while True:
            ex = gen.send(_dist_exchange(ex, device, group, stats))
    except StopIteration as stop:
        return stop.value


class CaptureAborted(RuntimeError):
    """Raised on EVERY rank when one rank's segment failed during ShardedClipGraph.capture(vote=...)."""


class _RawSlice:
    """Stands for a whole-clip input array of which one rank reads exactly the slice [r0, r1): sharded_clip_steps only asks for the
    length, the frame shape and that slice."""

    def __init__(self, t, r0, total):
        self.t, self.r0, self.total = t, r0, total
        self.shape = (total,) + tuple(t.shape[1:])

    def __len__(self):
        return self.total

    def __getitem__(self, sl):
        assert isinstance(sl, slice) and sl.start == self.r0 and sl.stop == self.r0 + self.t.shape[0], (sl, self.r0, self.t.shape)
        return self.t


class ShardedClipGraph:
    """One rank's part of the sharded pass over clips of ONE shape as hipGraphs (the sharded counterpart of pipeline.ClipGraph).

    The eager sharded pass issues ~10 k launches per rank from Python (~170 us of host time each): with one sub-video per GPU the
    ranks are host-bound.  The pass has no data-dependent host control flow between two halo exchanges, so the generator
    ``sharded_clip_steps`` is run ONCE under hipGraph capture from one exchange request to the next -- one graph per compute segment
    (5 segments: the 4 exchanges of RAFT flows, completed flows, updated frames, boundary composites).  A replay is graph_0, exchange_0,
    graph_1, ... : the exchanges stay eager point-to-point ops (RCCL send / recv) on the send tensors the previous graph wrote and into
    receive buffers the next graph reads -- all at fixed addresses.  Same kernels, same order: bit-identical to the eager pass.

        g = ShardedClipGraph(models, L, H, W, cfg, device, rank, world)
        g.load(frames_u8, flow_masks_u8, masks_dilated_u8)          # whole-clip host arrays; the rank's slice is uploaded
        g.capture(exchange)                                          # exchange(ex, bufs): answers one Exchange into bufs
        lo, comp = g.replay(exchange)                                # (after another load(): the next clip)
    """

    def __init__(self, models, L, H, W, cfg, device, rank, world, pool=None):
        """``pool`` (optional graph memory pool handle): capture into this pool instead of a private one -- for several logical ranks of
        ONE process whose graphs are replayed strictly in their capture order (StreamingClipGraph(share_pool=True))."""
        import dataclasses
        cfg = dataclasses.replace(cfg, window_streams=1, raft_streams=1)      # no forked branches inside a captured segment (pipeline.ClipGraph)
        self.models, self.cfg, self.device, self.rank, self.world, self.L = models, cfg, torch.device(device), rank, world, L
        self._pool = pool
        self.r0, self.r1 = ShardPlan(L, cfg, world).need_raw(rank)
        n = max(0, self.r1 - self.r0)
        self.frames = torch.zeros((n, H, W, 3), dtype=torch.uint8, device=device)
        self.flow_masks = torch.zeros((n, H, W), dtype=torch.uint8, device=device)
        self.masks_dilated = torch.zeros((n, H, W), dtype=torch.uint8, device=device)
        self.segments = []          # [(graph, Exchange or None, {q: receive buffer})]
        self.result = None
        self._gen = None
        self._pinned = []

    def load(self, frames_u8, flow_masks_u8, masks_dilated_u8):
        to_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
        for dst, src in ((self.frames, frames_u8), (self.flow_masks, flow_masks_u8), (self.masks_dilated, masks_dilated_u8)):
            if dst.shape[0]:
                dst.copy_(to_t(src[self.r0:self.r1]), non_blocking=True)

    def _inputs(self):
        return tuple(_RawSlice(t, self.r0, self.L) for t in (self.frames, self.flow_masks, self.masks_dilated))

    def eager(self, exchange, vote=None):
        """One eager pass on the static inputs (builds engines, tables and cached index tensors; must precede capture()).
        ``vote``: see capture()."""
        gen = sharded_clip_steps(self.models, *self._inputs(), self.cfg, self.device, self.rank, self.world)
        step = lambda got: next(gen) if got is None else gen.send(got)
        got = None
        while True:
            try:
                ex = self._voted(lambda: step(got), vote, "eager warm-up")
            except StopIteration as stop:
                return stop.value
            bufs = {q: torch.empty(shape, dtype=dtype, device=self.device) for q, (shape, dtype) in ex.recv.items()}
            exchange(ex, bufs)
            got = bufs

    def _voted(self, fn, vote, what):
        """Runs one compute segment and -- when ``vote`` is given -- lets all ranks agree that it succeeded BEFORE any of them
        enters the exchange behind it: a rank whose segment raised votes 0 and re-raises as CaptureAborted, its peers get 0 back and
        raise CaptureAborted too, so nobody is left alone inside batch_isend_irecv.  StopIteration (the generator's return) passes."""
        err, out, stop = None, None, None
        try:
            out = fn()
        except StopIteration as e:
            stop = e
        except Exception as e:      # noqa: BLE001 -- whatever it was, the peers must hear about it
            err = e
        if vote is not None and not vote(err is None):
            self.reset()
            raise CaptureAborted(f"rank {self.rank}: {what}: " + (f"{type(err).__name__}: {err}" if err is not None else "a peer rank failed"))
        if err is not None:
            raise err
        if stop is not None:
            raise stop
        return out

    def reset(self):
        """Drops every captured segment and the half-run generator (after an abandoned capture): the object can capture() again."""
        self.segments, self.result, self._gen, self._pinned = [], None, None, []

    def capture_next(self, got):
        """Captures the next compute segment (from the answer `got` of the previous exchange -- None for the first -- to the next
        exchange request).  Returns that request with freshly allocated receive buffers, or (None, None) after the last segment."""
        from . import pipeline
        if self._gen is None:
            self._gen = sharded_clip_steps(self.models, *self._inputs(), self.cfg, self.device, self.rank, self.world)
        g = torch.cuda.CUDAGraph()
        pool = self._pool if self._pool is not None else (self.segments[0][0].pool() if self.segments else None)
        prev, pipeline._index_recorder = pipeline._index_recorder, self._pinned      # the graphs own the cached index tensors they read
        ex, done = None, False
        try:
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                try:
                    ex = next(self._gen) if got is None and not self.segments else self._gen.send(got)
                except StopIteration as stop:
                    self.result, done = stop.value, True
        finally:
            pipeline._index_recorder = prev
        if done:
            self.segments.append((g, None, None))
            return None, None
        bufs = {q: torch.empty(shape, dtype=dtype, device=self.device) for q, (shape, dtype) in ex.recv.items()}
        self.segments.append((g, ex, bufs))
        return ex, bufs

    def capture(self, exchange, vote=None):
        """Eager warm-up pass, then the capture pass (its exchanges are executed for real: the peers capture in lockstep).
        ``vote(ok) -> bool`` (optional; e.g. an all-reduce MIN over the ranks): called by every rank after every compute segment of both
        passes, before the exchange that follows it -- if any rank failed, ALL ranks raise ``CaptureAborted`` at the same point instead
        of one rank raising while its peers wait in a point-to-point exchange that will never be answered."""
        self.eager(exchange, vote)
        torch.cuda.synchronize(self.device)
        got = None
        while True:
            ex, bufs = self._voted(lambda: self.capture_next(got), vote, f"capture of segment {len(self.segments)}")
            if ex is None:
                break
            exchange(ex, bufs)
            got = bufs
        torch.cuda.synchronize(self.device)
        return self

    def replay(self, exchange):
        for g, ex, bufs in self.segments:
            g.replay()
            if ex is not None:
                exchange(ex, bufs)
        return self.result


def dist_exchanger(device, group=None, stats=None):
    """exchange(ex, bufs) over torch.distributed for ShardedClipGraph (receives into the static buffers)."""
    return lambda ex, bufs: _dist_exchange(ex, device, group, stats, bufs=bufs)


def gather_frames(lo, comp, total, dst=0, group=None):
    """Collects the ranks' composited frames on rank `dst` (uint8 [total,H,W,3] there, None elsewhere)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    meta = [None] * world
    dist.all_gather_object(meta, (lo, comp.shape[0]), group=group)
    via_host = dist.get_backend(group) == "gloo"
    if rank != dst:
        if comp.shape[0]:
            dist.send(comp.cpu() if via_host else comp.contiguous(), dst, group=group)
        return None
    out = torch.empty((total,) + tuple(comp.shape[1:]), dtype=comp.dtype, device=comp.device)
    for q, (qlo, n) in enumerate(meta):
        if n == 0:
            continue
        if q == rank:
            out[qlo:qlo + n] = comp
        else:
            buf = torch.empty((n,) + tuple(comp.shape[1:]), dtype=comp.dtype, device="cpu" if via_host else comp.device)
            dist.recv(buf, q, group=group)
            out[qlo:qlo + n] = buf.to(comp.device)
    return out


def run_logical_shards(models, frames_u8, flow_masks_u8, masks_dilated_u8, cfg, device, world):
    """Runs `world` logical ranks in ONE process (e.g. on one GPU), handing exchanged tensors over directly.  Used to
    validate the sharded pass against ``run_clip`` where only one device is available.  Returns uint8 [L,H,W,3]."""
    gens = [sharded_clip_steps(models, frames_u8, flow_masks_u8, masks_dilated_u8, cfg, device, r, world) for r in range(world)]
    reqs, results = [None] * world, [None] * world
    for r, g in enumerate(gens):
        reqs[r] = next(g)
    while any(r is not None for r in reqs):
        nxt = [None] * world
        for r, g in enumerate(gens):
            if reqs[r] is None:
                continue
            got = {}
            for q, (shape, dtype) in reqs[r].recv.items():
                t = reqs[q].send[r]
                assert tuple(t.shape) == tuple(shape) and t.dtype == dtype, (reqs[r].tag, r, q, t.shape, shape)
                got[q] = t
            try:
                nxt[r] = g.send(got)
            except StopIteration as stop:
                results[r] = stop.value
        # every rank must be at the same exchange (SPMD)
        tags = {x.tag for x in nxt if x is not None}
        assert len(tags) <= 1, tags
        reqs = nxt
    L = len(frames_u8)
    out = torch.zeros((L,) + tuple(results[0][1].shape[1:]), dtype=torch.uint8, device=device)
    for lo, comp in results:
        out[lo:lo + comp.shape[0]] = comp
    return out


def _copy_exchange(reqs):
    """In-process answer of one SPMD exchange step: reqs[r] = (Exchange or None, {q: receive buffer}) of logical rank r."""
    for r, (ex, bufs) in enumerate(reqs):
        if ex is None:
            continue
        for q, (shape, dtype) in ex.recv.items():
            t = reqs[q][0].send[r]
            assert tuple(t.shape) == tuple(shape) and t.dtype == dtype, (ex.tag, r, q, t.shape, shape)
            bufs[q].copy_(t)


def run_logical_shards_graphed(models, frames_u8, flow_masks_u8, masks_dilated_u8, cfg, device, world, replays=2, clips=None):
    """`world` logical ranks in ONE process, each as a ShardedClipGraph (captured in lockstep, exchanged tensors copied between the
    ranks' static buffers): validates the graph form of the sharded pass where only one GPU is available.  Returns the uint8
    [L,H,W,3] result of the LAST replay; `clips` (optional list of (frames, flow_masks, masks_dilated)) are loaded before the replays
    in turn (same shape), the default replays the clip that was captured."""
    L, H, W = len(frames_u8), frames_u8.shape[1], frames_u8.shape[2]
    graphs = [ShardedClipGraph(models, L, H, W, cfg, device, r, world) for r in range(world)]
    for g in graphs:
        g.load(frames_u8, flow_masks_u8, masks_dilated_u8)
    run_logical_shards(models, frames_u8, flow_masks_u8, masks_dilated_u8, cfg, device, world)      # eager warm-up of every rank's engines / tables
    torch.cuda.synchronize(device)
    got = [None] * world
    while True:
        reqs = [g.capture_next(got[r]) for r, g in enumerate(graphs)]
        assert len({ex.tag if ex is not None else None for ex, _ in reqs}) == 1, "ranks must stop at the same exchange"
        if reqs[0][0] is None:
            break
        _copy_exchange(reqs)
        got = [b for _, b in reqs]
    nseg = len(graphs[0].segments)
    assert all(len(g.segments) == nseg for g in graphs)
    for i in range(replays):
        if clips:
            for g in graphs:
                g.load(*clips[i % len(clips)])
        for s_ in range(nseg):
            for g in graphs:
                g.segments[s_][0].replay()
            _copy_exchange([(g.segments[s_][1], g.segments[s_][2]) for g in graphs])
    out = torch.zeros((L, H, W, 3), dtype=torch.uint8, device=device)
    for g in graphs:
        lo, comp = g.result
        out[lo:lo + comp.shape[0]] = comp
    return out, nseg


# ----------------------------------------------------------------------------------------------------------------
# streaming schedule of one long clip on ONE GPU
# ----------------------------------------------------------------------------------------------------------------
def wavefront_order(world, nseg, sources):
    """Host-side issue order of the (rank, segment) launches of a streaming pass.

    ``sources(r, s)`` = the ranks whose segment ``s`` output the exchange after segment ``s`` delivers to rank r (the keys of that
    Exchange's ``recv``): segment s + 1 of rank r may be issued once segment s of r itself and of every such rank has been issued.
    Among the ready launches the one with the smallest (rank + segment, segment) goes first -- a diagonal wavefront: when stage D
    (segment 3) of sub-video k is issued, stage C of k + 1, stage B of k + 2 and RAFT of k + 3 are issued right behind it, each on its
    own stream.  Pure host arithmetic (tested on CPU); raises if the dependencies admit no order."""
    pending = sorted((r + s, s, r) for r in range(world) for s in range(nseg))
    issued, order = set(), []
    while pending:
        for i, (_, s, r) in enumerate(pending):
            if s == 0 or ((r, s - 1) in issued and all((q, s - 1) in issued for q in sources(r, s - 1))):
                break
        else:
            raise RuntimeError("streaming schedule: no segment is ready (cyclic exchange dependencies)")
        pending.pop(i)
        issued.add((r, s))
        order.append((r, s))
    return order


def _generator_frames(gen):
    """The frames of a suspended generator and of every generator it drives: ``yield from`` delegates (gi_yieldfrom) and generators held
    in a local variable -- ``@torch.no_grad()`` wraps a generator function in a driver generator whose local ``gen`` is the real one."""
    import types
    frames, todo, seen = [], [gen], set()
    while todo:
        g = todo.pop()
        if g is None or id(g) in seen or getattr(g, "gi_frame", None) is None:
            continue
        seen.add(id(g))
        frames.append(g.gi_frame)
        todo.append(getattr(g, "gi_yieldfrom", None))
        todo.extend(v for v in g.gi_frame.f_locals.values() if isinstance(v, types.GeneratorType))
    return frames


def _generator_tensors(gen):
    """Every tensor reachable from the local variables of a suspended generator's frames: what is alive at a yield."""
    out, seen = [], set()

    def walk(v, depth=0):
        if id(v) in seen or depth > 6:
            return
        seen.add(id(v))
        if torch.is_tensor(v):
            out.append(v)
        elif isinstance(v, (list, tuple, set)):
            for x in v:
                walk(x, depth + 1)
        elif isinstance(v, dict):
            for x in v.values():
                walk(x, depth + 1)
        elif isinstance(v, Span):
            walk(v.t, depth + 1)
        elif hasattr(v, "__dict__") and type(v).__module__.startswith("propainter_amd"):
            walk(vars(v), depth + 1)
    for fr in _generator_frames(gen):
        walk(dict(fr.f_locals))
    return out


class StreamingClipGraph:
    """ONE long clip on ONE GPU as a pipeline over its sub-videos: RAFT of sub-video k + 3, flow completion of k + 2 and image
    propagation of k + 1 run next to the generator windows of sub-video k (the reference walks the four stages over
    the whole clip one after the other, inference_propainter.py:298-452).

    Every sub-video block is a LOGICAL rank of the sharded pass (``ShardedClipGraph``: its compute segments between the four halo
    exchanges as hipGraphs, captured once per clip shape); a replay issues the (rank, segment) graphs in ``wavefront_order``, each
    logical rank on its own HIP stream, the exchanges as device copies between the ranks' static buffers ordered by events.  A
    generator window of sub-video k reads reference frames of sub-video k + 1, whose image propagation needs completed flows reaching
    into k + 2, whose completion needs RAFT flows of k + 3 -- that dependency chain IS the halo exchange of the sharded pass, so the
    wavefront over its segments is the deepest overlap the data flow admits.  Same kernels on the same data as ``run_clip``: the
    composited frames are bit-identical (tests/test_modules_gpu.py).

    Memory: with ``share_pool=True`` (default) all (rank, segment) graphs are CAPTURED IN THE WAVEFRONT ORDER into ONE graph memory
    pool and replayed in exactly that order, one launch after the other (the chained default of ``replay``): a block a segment has
    released serves the segments captured after it, so the pass holds what is alive at one point of the pipeline -- sub-video scale --
    instead of the sum of ``world`` private pools.  Graphs that share a pool must never overlap or change order: ``share_pool=False``
    gives every logical rank its private pool, as the lockstep A/B order and the concurrent mode need.  The fp32 correlation volumes
    of RAFT's exact-f32 mode get ``volume_gb`` (default 40 GB, the single-pass budget) split over the ranks; one RAFT stream each.

        s = StreamingClipGraph(models, L, H, W, cfg, device)         # world = number of sub-video blocks of the clip
        s.load(frames_u8, flow_masks_u8, masks_dilated_u8); s.capture()
        out_u8 = s.replay()                                           # [L, H, W, 3]; load() + replay() for the next clip
    """

    NSEG = 5          # compute segments of sharded_clip_steps: RAFT | completion | image propagation | windows | boundary blends

    def __init__(self, models, L, H, W, cfg, device, world=None, volume_gb=40.0, share_pool=True, single_graph=False, validate=8):
        """DEFAULT (round 6): one graph per (rank, segment) in ONE memory pool, replayed chained in the wavefront order -- the form that
        has never produced a wrong byte.  The overlapped form is opt-in because its failure mode is silent and unexplained: the
        capture-time hazard checker (propainter_amd/hazard.py, tools/check_hazards.py) finds NO unordered access in the stage-pipelined
        capture -- neither with the shipped stage map nor with flow completion on a third branch -- yet the third-branch map gives
        frames that differ from the eager pass at 720x1280x320 (3 of 3 replays, profiles/r6_hazards.txt).  Whatever reorders those
        bytes sits below the submitted program (runtime / cache coherence between concurrently running branches), so a validation on
        the capture clip can only sample it.

        single_graph=True (round 5): the whole wavefront as ONE hipGraph, pipelined by stage -- RAFT of the sub-videos on a
        branch forked from the capture stream (round 5 had image propagation on a second one: PP_SG_STAGES), flow completion, image
        propagation, the generator windows and the boundary blends on the capture stream, the exchanges as direct tensor hand-overs ordered by captured events
        (``_capture_single_graph``); ``validate`` replays of the captured graph are compared bit for bit with the eager pass at capture
        time, and a graph that does not reproduce it is replaced by the chained per-segment graphs (with a warning).  The overlap of the
        schedule lives INSIDE one graph (parallel branches: the form the whole-pass ClipGraph and the generator lanes use), not in
        concurrent launches of several graphs, which give wrong frames on ROCm 7.2 (see ``replay``).  single_graph=False: one graph per
        (rank, segment), replayed chained (share_pool=True: one memory pool, capture order) or in the lockstep / concurrent A/B orders
        (share_pool=False)."""
        import dataclasses
        self.device = torch.device(device)
        nsub = -(-L // cfg.subvideo_length)
        self.world = world or nsub
        if not can_shard(L, cfg, self.world):
            raise ValueError(f"a {L}-frame clip does not split into {self.world} sub-video blocks of {cfg.subvideo_length} frames")
        self.models, self.L, self.H, self.W = models, L, H, W
        # one RAFT lane per rank (the ranks run next to each other) and no window lanes: no forked branches inside a captured graph beyond
        # the stage map's own (pipeline.ClipGraph: config 5 with lanes left wrong bytes in 3 of 64 replays, without in 0 of 84)
        self.cfg = dataclasses.replace(cfg, raft_streams=1, window_streams=1)
        self.volume_gb = float(volume_gb)
        self.single_graph = bool(single_graph)
        self.validate = int(validate)       # single_graph: replays checked bit for bit against the eager warm-up pass at capture time (0: off)
        self.share_pool = bool(share_pool) and not self.single_graph
        self._single, self._single_out, self._single_keep = None, None, None
        pool = torch.cuda.graph_pool_handle() if self.share_pool else None
        self.graphs = [ShardedClipGraph(models, L, H, W, self.cfg, device, r, self.world, pool=pool) for r in range(self.world)]
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.world)]
        self.order = None
        self._inputs = None

    def load(self, frames_u8, flow_masks_u8, masks_dilated_u8):
        self._inputs = (frames_u8, flow_masks_u8, masks_dilated_u8)
        for g in self.graphs:
            g.load(frames_u8, flow_masks_u8, masks_dilated_u8)

    def capture(self):
        """Eager warm-up of all logical ranks (engines, tables), then their segments captured in lockstep (as
        run_logical_shards_graphed); the issue order of the replays is fixed here."""
        if self._inputs is None:
            raise RuntimeError("StreamingClipGraph.capture(): load() a clip first (the warm-up and the capture pass run on it)")
        raft = self.models[0]
        saved = getattr(raft, "volume_budget_bytes", None)
        if saved is not None:
            raft.volume_budget_bytes = self.volume_gb * 1e9 / self.world
        try:
            warm = run_logical_shards(self.models, *self._inputs, self.cfg, self.device, self.world)
            torch.cuda.synchronize(self.device)
            warm = warm.cpu()                             # the eager pass's bytes: what the captured graph must reproduce (validate)
            torch.cuda.empty_cache()                      # the warm-up's cached blocks go back before `world` private graph pools grow
            if self.single_graph:
                self.order = self._capture_single_graph()
                torch.cuda.synchronize(self.device)
                if self.validate and not self._validate_single_graph(warm):
                    import warnings
                    warnings.warn("StreamingClipGraph: the stage-pipelined hipGraph did not reproduce the eager pass bit for bit in "
                                  f"{self.validate} replays on this runtime; falling back to chained launches of per-segment graphs "
                                  "(same frames, no overlap)", RuntimeWarning)
                    self._single, self._single_out, self._single_keep = None, None, None
                    torch.cuda.empty_cache()
                    self.single_graph, self.share_pool = False, True
                    pool = torch.cuda.graph_pool_handle()
                    self.graphs = [ShardedClipGraph(self.models, self.L, self.H, self.W, self.cfg, self.device, r, self.world, pool=pool)
                                   for r in range(self.world)]
                    self.load(*self._inputs)
                    order = self._capture_in_wavefront_order()
                    torch.cuda.synchronize(self.device)
                    self.order = wavefront_order(self.world, self.NSEG, lambda r, s_: sorted(self.graphs[r].segments[s_][1].recv))
                    assert order == self.order
                return self
            if self.share_pool:
                order = self._capture_in_wavefront_order()
            else:
                got = [None] * self.world
                while True:
                    reqs = [g.capture_next(got[r]) for r, g in enumerate(self.graphs)]
                    assert len({ex.tag if ex is not None else None for ex, _ in reqs}) == 1, "ranks must stop at the same exchange"
                    if reqs[0][0] is None:
                        break
                    _copy_exchange(reqs)
                    got = [b for _, b in reqs]
                order = None
        finally:
            if saved is not None:
                raft.volume_budget_bytes = saved
        torch.cuda.synchronize(self.device)
        nseg = len(self.graphs[0].segments)
        assert nseg == self.NSEG and all(len(g.segments) == nseg for g in self.graphs)
        self.order = wavefront_order(self.world, nseg, lambda r, s: sorted(self.graphs[r].segments[s][1].recv))
        assert order is None or order == self.order, "shared pool: the replay order must be the capture order"
        return self

    def _capture_in_wavefront_order(self):
        """Captures the (rank, segment) graphs in the order ``wavefront_order`` will issue them (same greedy rule; the sources of an
        exchange are known once the segment in front of it has been captured), answering every exchange from the captured send
        tensors.  Returns the order."""
        pending = sorted((r + s, s, r) for r in range(self.world) for s in range(self.NSEG))
        issued, order, reqs = set(), [], {}
        while pending:
            for i, (_, s, r) in enumerate(pending):
                if s == 0 or ((r, s - 1) in issued and all((q, s - 1) in issued for q in reqs[(r, s - 1)][0].recv)):
                    break
            else:
                raise RuntimeError("streaming schedule: no segment is ready (cyclic exchange dependencies)")
            pending.pop(i)
            got = None
            if s > 0:
                ex_prev, got = reqs[(r, s - 1)]
                for q, (shape, dtype) in ex_prev.recv.items():
                    t = reqs[(q, s - 1)][0].send[r]
                    assert tuple(t.shape) == tuple(shape) and t.dtype == dtype, (ex_prev.tag, r, q, t.shape, shape)
                    got[q].copy_(t)
            ex, bufs = self.graphs[r].capture_next(got)
            assert (ex is None) == (s == self.NSEG - 1), f"rank {r}: segment {s} of {self.NSEG}"
            reqs[(r, s)] = (ex, bufs)
            issued.add((r, s))
            order.append((r, s))
        return order

    def _validate_single_graph(self, expected_cpu):
        """The captured graph replayed ``self.validate`` times on the capture clip: every replay must give the eager pass's bytes.
        (A capture whose branches race is otherwise silent: the frames are plausible, a byte off here and there.)"""
        for _ in range(int(self.validate)):
            self._single.replay()
            torch.cuda.synchronize(self.device)
            if not torch.equal(self._single_out.cpu(), expected_cpu):
                return False
        return True

    def _capture_single_graph(self):
        """The wavefront as ONE hipGraph, pipelined BY STAGE: the logical ranks' generators (``sharded_clip_steps``) run under a single
        capture in ``wavefront_order`` (same greedy rule as the multi-graph capture), segment s of every rank on STAGE stream s --
        RAFT of all sub-videos on one branch, image propagation on another (side streams forked from the capture stream), flow
        completion, the generator windows (which fork their own lanes) and the boundary blends on the capture stream itself (see the
        stage map below for why flow completion is not on a branch of its own).  An
        exchange is answered by handing the sender's tensors over (no copy); segment s waits for the events recorded behind the
        segments s - 1 it reads from.  Every dependency therefore points from stage stream s - 1 to stage stream s: a stream never waits
        for a stream that waited for it.  That shape is forced by the runtime: on ROCm 7.2 hipStreamEndCapture SEGFAULTS when two forked
        streams depend on each other back and forth, or when a forked stream forks lanes of its own (minimal repro:
        tools/diag_capture_edges.py, profiles/r5_streaming_single_graph.txt) -- one stream per logical rank has both.  It is also the
        pipeline the schedule is after: RAFT of sub-video k + 3 runs next to the windows of sub-video k.

        Memory safety across the stage streams: a tensor that lives across a segment boundary is allocated on one stage stream and
        read on the next; the caching allocator would hand its block to a later allocation of the FIRST stream as soon as the generator
        drops it.  So every tensor alive in a generator at a segment boundary stays referenced until the capture ends (no block that
        crossed a stream is recycled inside the graph).  Returns the issue order."""
        from . import pipeline
        cfg, dev = self.cfg, self.device
        _window_streams(dev, cfg.window_streams, "streaming")        # the windows' lanes exist before the capture starts
        gens = [sharded_clip_steps(self.models, *g._inputs(), cfg, dev, r, self.world, lane_key="streaming")
                for r, g in enumerate(self.graphs)]
        graph = torch.cuda.CUDAGraph()
        keep = []
        prev_rec, pipeline._index_recorder = pipeline._index_recorder, keep
        pending = sorted((r + s, s, r) for r in range(self.world) for s in range(self.NSEG))
        issued, order, reqs, done, results = set(), [], {}, {}, [None] * self.world
        side = self.streams[:3]
        assert len(side) == 3 or self.world < 3
        while len(side) < 3:
            side.append(torch.cuda.Stream(dev))
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                cur = torch.cuda.current_stream(dev)
                # which stages leave the capture stream: RAFT (0) and image propagation (2) on branches of their own, flow completion (1)
                # with the windows on the capture stream.  MEASURED at BASELINE config 4 (720x1280x320, profiles/r5_streaming_single_graph.txt):
                # with flow completion on a third branch the LAST sub-video's completed flows differed from replay to replay (max 0.025 px,
                # +-1 byte in 0.2 % of that sub-video's bytes; smaller clips never showed it, the kernels are bit-stable next to unrelated
                # load, lanes on / off made no difference); with this map every stage of every rank equals the eager pass and the pass is
                # 2 % faster (flow completion is latency-bound and overlaps the windows' tail either way).  PP_SG_STAGES overrides (diagnosis).
                # Round 6 (profiles/r6_replay_bytes.txt): what goes wrong on a forked branch is a long chain of small dependent launches -- flow
                # completion here, a window's feature propagation in the whole-pass graph (~5 % of its replays until round 6).  Image
                # propagation (158 steps of 20 us) is such a chain too, so the default map keeps only RAFT's large launches on a branch.
                on_side = {int(v) for v in os.environ.get("PP_SG_STAGES", "0").split(",") if v != ""}
                # PP_SG_FENCE (diagnosis, round 6; bit 0: L2 write-back kernel in front of every segment's event, bit 1: L2 invalidate kernel
                # behind every cross-branch wait): tests whether the deviation of the three-branch map is a cache-visibility defect of
                # cross-queue graph edges (profiles/r6_graph_queues.txt)
                fence = int(os.environ.get("PP_SG_FENCE", "0"))
                stage = [side[i] if i in on_side else cur for i in range(3)] + [cur, cur]
                side = [st for st in side if st in stage]
                for st in side:
                    st.wait_stream(cur)
                while pending:
                    for i, (_, s, r) in enumerate(pending):
                        if s == 0 or ((r, s - 1) in issued and all((q, s - 1) in issued for q in reqs[(r, s - 1)].recv)):
                            break
                    else:
                        raise RuntimeError("streaming schedule: no segment is ready (cyclic exchange dependencies)")
                    pending.pop(i)
                    st = stage[s]
                    with torch.cuda.stream(st):
                        got, crossed = None, False
                        if s > 0:
                            got = {}
                            if stage[s - 1] is not st:
                                st.wait_event(done[(r, s - 1)])
                                crossed = True
                            for q, (shape, dtype) in sorted(reqs[(r, s - 1)].recv.items()):
                                t = reqs[(q, s - 1)].send[r]
                                assert tuple(t.shape) == tuple(shape) and t.dtype == dtype, (reqs[(r, s - 1)].tag, r, q, t.shape, shape)
                                if stage[s - 1] is not st:
                                    st.wait_event(done[(q, s - 1)])
                                    crossed = True
                                got[q] = t
                            if crossed and fence & 2:      # [diagnosis] acquire by hand behind the cross-branch waits
                                hip.debug_cache_fence(dev, 2)
                        try:
                            ex = next(gens[r]) if s == 0 else gens[r].send(got)
                        except StopIteration as stop:
                            ex, results[r] = None, stop.value
                        assert (ex is None) == (s == self.NSEG - 1), f"rank {r}: segment {s} of {self.NSEG}"
                        reqs[(r, s)] = ex
                        keep.append((ex, got, _generator_tensors(gens[r])))
                        if os.environ.get("PP_SG_DEBUG") == "1":      # tools/diag_stream2.py: named stage outputs
                            self._single_debug = getattr(self, "_single_debug", {})
                            self._single_debug[(r, s)] = {k: (v.t if isinstance(v, Span) else v) for fr_ in _generator_frames(gens[r])
                                                          for k, v in fr_.f_locals.items() if torch.is_tensor(v) or isinstance(v, Span)}
                        if fence & 1:                      # [diagnosis] release by hand in front of the event other branches wait for
                            hip.debug_cache_fence(dev, 1)
                        ev = torch.cuda.Event()
                        ev.record(st)
                        done[(r, s)] = ev
                    issued.add((r, s))
                    order.append((r, s))
                for st in side:
                    cur.wait_stream(st)
                out = torch.zeros((self.L, self.H, self.W, 3), dtype=torch.uint8, device=dev)
                for lo, comp in results:
                    out[lo:lo + comp.shape[0]] = comp
        finally:
            pipeline._index_recorder = prev_rec
        self._single, self._single_out = graph, out
        self._single_keep = [t for t in keep if torch.is_tensor(t)]      # the cached index tensors the graph reads (see pipeline._dev_index)
        return order

    def replay(self, lockstep=False, concurrent=None):
        """One pass over the loaded clip.  lockstep=True: segment by segment over all ranks on the current stream (the schedule of
        run_logical_shards_graphed: the A/B reference of the streaming order).

        concurrent (default: env PP_STREAM_CONCURRENT == "1", else False; single_graph=False only): let the ranks' segment graphs
        overlap on their streams, ordered by the exchange events only.  UNSAFE on ROCm 7.2 / MI355X, kept for diagnosis: 10-14 of 16
        passes give wrong frames (tools/diag_stream2.py, profiles/r5_streaming_single_graph.txt).  Round 5 narrowed it down: it is a race
        between graph launches that are in flight at the same time, not stale input -- a second pass over the same inputs is wrong as
        often (round 4 read "never" off a luckier box); it does not need the copy engine (inputs written by a device kernel: same rate),
        shared module state (private engines per rank: same), forked branches inside the graphs (window_streams=1: same), memcpy
        exchanges (kernel copies: same) nor a runtime flag (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, GPU_MAX_HW_QUEUES=8,
        HIP_FORCE_DEV_KERNARG=0: same); the kernels themselves are bit-exact next to unrelated load (tools/diag_noise.py) and on a
        forked branch of one graph (tools/diag_branch.py).  The overlapped schedule therefore ships as ONE graph
        (``single_graph=True``: 0 wrong first passes of 24, 17 % faster than the chained launches at the diagnostic size); this
        multi-graph form stays chained by default (wavefront issue order, no overlap) and warns when the overlap is requested."""
        if concurrent is None:
            concurrent = os.environ.get("PP_STREAM_CONCURRENT") == "1"
        if self.order is None:
            raise RuntimeError("StreamingClipGraph.replay(): capture() first")
        if self._single is not None:
            if lockstep:
                raise ValueError("StreamingClipGraph(single_graph=True) holds one graph: there is no lockstep order to replay")
            self._single.replay()                         # on the current stream, behind the uploads of load()
            return self._single_out.clone()               # a fresh tensor like the chained form's: the next replay overwrites the static buffer
        if concurrent:
            import warnings
            warnings.warn("StreamingClipGraph.replay(concurrent=True): overlapping launches of several hipGraphs give WRONG frames in most "
                          "passes on ROCm 7.2 / MI355X (profiles/r5_streaming_single_graph.txt); use single_graph=True for the overlapped "
                          "schedule", RuntimeWarning, stacklevel=2)
        if self.share_pool and (lockstep or concurrent):
            raise ValueError("StreamingClipGraph(share_pool=True): the graphs share one memory pool and must replay in their capture "
                             "order, one after the other; build with share_pool=False for the lockstep / concurrent orders")
        cur = torch.cuda.current_stream(self.device)
        nseg = len(self.graphs[0].segments)
        if lockstep:
            for s in range(nseg):
                for g in self.graphs:
                    g.segments[s][0].replay()
                _copy_exchange([(g.segments[s][1], g.segments[s][2]) for g in self.graphs])
        else:
            done, prev = {}, None
            for st in self.streams:
                st.wait_stream(cur)                       # the uploads of load() / the previous pass's readers are ordered before this pass
            for r, s in self.order:
                st = self.streams[r]
                with torch.cuda.stream(st):
                    if s > 0:                             # the sources of exchange s - 1 have delivered into this rank's receive buffers
                        for q in sorted(self.graphs[r].segments[s - 1][1].recv):
                            st.wait_event(done[(q, s - 1)])
                    if not concurrent and prev is not None:
                        st.wait_event(done[prev])        # chained launches: see the docstring
                    self.graphs[r].segments[s][0].replay()
                    # PUSH: the sender delivers -- the copies of exchange s run on ITS stream, right behind the graph that produced the
                    # tensors, into the receivers' static buffers (one per exchange and source: written once per pass, read by the receiver
                    # only after the event below).  Round 3 had the RECEIVER copy on its own stream after waiting for the sender's event:
                    # on MI355X that read stale tensors of the previous clip in ~1 of 4 passes (tools/diag_stream.py, profiles/r4_streaming_race.txt)
                    ex_s = self.graphs[r].segments[s][1]
                    if ex_s is not None:
                        for q, t in sorted(ex_s.send.items()):
                            shape, dtype = self.graphs[q].segments[s][1].recv[r]
                            assert tuple(t.shape) == tuple(shape) and t.dtype == dtype, (ex_s.tag, r, q, t.shape, shape)
                            self.graphs[q].segments[s][2][r].copy_(t)
                    ev = torch.cuda.Event()
                    ev.record(st)
                    done[(r, s)] = ev
                    prev = (r, s)
            for st in self.streams:
                cur.wait_stream(st)
        out = torch.zeros((self.L, self.H, self.W, 3), dtype=torch.uint8, device=self.device)
        for g in self.graphs:
            lo, comp = g.result
            out[lo:lo + comp.shape[0]] = comp
        return out
