"""Capture-time memory-hazard checker for the multi-stream submission paths (pipeline.ClipGraph lanes, sharding.StreamingClipGraph).

What it answers: "is there a pair of device accesses to overlapping memory, at least one a write, that the recorded stream / event
edges do NOT order?" -- the question a byte-level replay comparison can only sample (round 5: +-1 byte in 0.2 % of one sub-video, 1 pass
in N).  The reference's chunk loops are sequential (inference_propainter.py:342-398), so ANY such pair in this engine's schedule is a bug
of the schedule, whether or not it shows on a given box.

How: a host-side happens-before analysis of the program the host SUBMITS (works the same in eager mode and under hipGraph capture; the
GPU is needed only because the addresses come from torch's caching allocator, whose per-stream block reuse is what is being checked):

  * every engine launch reports the tensors it was handed (``hip._p`` / ``hip._pw`` / ``hip.conv2d_raw`` when a recorder is installed);
    every torch (aten) kernel is seen by a TorchDispatchMode -- reads = tensor arguments, writes = schema-mutated arguments + outputs;
  * a FRESH allocation (an op output whose storage is not one of its inputs') at an address range starts a new GENERATION of that range
    and closes the generations it overlaps: the caching allocator has recycled the block.  Every access of the new generation must
    happen after every access of the closed ones (class "alias": needs no read / write knowledge -- the first touch of recycled memory
    is a write).  The allocator itself only guarantees that for the allocating stream; a reader on another captured stream needs an edge;
  * inside one generation, two accesses on different streams with overlapping byte ranges, at least one a write, must be ordered (class
    "race": an ordinary missing ``wait_stream``);
  * ordering = vector clocks over streams: ``Event.record`` snapshots the stream's clock, ``Event.wait`` joins it (``wait_stream`` /
    ``wait_event`` / ``record_event`` are built on those two), ``torch.cuda.synchronize`` joins everything; ``tensor.record_stream``
    marks a generation as protected by the allocator for that stream (outside capture it defers the reuse to an event query);
  * a captured graph is summarised (merged read / write intervals); each ``CUDAGraph.replay`` is ONE access of those intervals on the
    replaying stream, so graph launches in flight together are checked against each other (the multi-graph streaming form).

Limits (stated, not hidden): allocations made INSIDE aten kernels (library workspaces) are invisible; a launch's byte range is the
contiguous extent of the tensor view it was handed (channel windows of one NHWC row overlap as ranges: reported as class "race?" when
both sides are strided windows of one buffer); raw pointers in ConvArgs come with the extent of the NHWC buffer they address.

    with hazard.Recorder(device) as rec:
        ...eager pass, capture, replays...
    report = rec.report()          # {"alias": [...], "race": [...], "race?": [...], "launches": n, ...}
"""
import bisect
import contextlib
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

_active = None


def active():
    return _active


class _Gen:
    """One life of an address range in the caching allocator."""
    __slots__ = ("lo", "hi", "gid", "born", "stream", "pred", "last", "acc", "recorded", "label", "has_write", "opaque")

    def __init__(self, lo, hi, gid, born, stream, label):
        self.lo, self.hi, self.gid, self.born, self.stream, self.label = lo, hi, gid, born, stream, label
        self.pred = {}            # stream -> (clock value that must be visible, description of the access that set it)
        self.last = {}            # stream -> (clock value of this generation's last access on it, description)
        self.acc = []             # (stream, n, lo, hi, is_write, strided, name, where)
        self.recorded = set()     # streams the allocator itself protects (record_stream)
        self.has_write = False
        self.opaque = label == "pre-existing"       # contents written by something the recorder did not see (no uninitialised-read check)


def _merge(iv):
    iv = sorted(iv)
    out = []
    for lo, hi in iv:
        if out and lo <= out[-1][1]:
            out[-1][1] = max(out[-1][1], hi)
        else:
            out.append([lo, hi])
    return out


def _overlap(a, b):
    """first overlapping pair of two merged interval lists, or None"""
    i = j = 0
    while i < len(a) and j < len(b):
        if a[i][1] <= b[j][0]:
            i += 1
        elif b[j][1] <= a[i][0]:
            j += 1
        else:
            return max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
    return None


class Recorder(TorchDispatchMode):
    def __init__(self, device, stacks=False, max_findings=200):
        super().__init__()
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:       # tensors report cuda:N, never a bare "cuda"
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.fresh = 0                    # allocations observed (sanity figure: a pass allocates thousands of tensors)
        self.aten_accesses = 0
        self.stacks = stacks
        self.max_findings = max_findings
        self.clock = {}                   # stream -> {stream: n}
        self.host = {}                    # what the host has waited for
        self.events = {}                  # id(event) -> clock snapshot
        self.starts, self.gens = [], []   # live generations sorted by lo
        self.ngen = 0
        self.launches = 0
        self.pending = []                 # pointers of the engine launch being assembled
        self.findings = {"alias": [], "race": [], "race?": [], "uninit": [], "intra": []}
        self.cap_id = 0                   # > 0 while a graph is being captured: such accesses describe the GRAPH, not the host timeline
        self.ncap = 0
        self.seen = set()
        self.names = {}                   # stream handle -> short name
        self.capturing = None             # summary under construction: {"r": [], "w": []}
        self.graph_summaries = {}         # id(graph) -> {"r": merged, "w": merged, "n": launches}
        self.labels = []                  # label stack (hazard.label("stage B"))
        self._patches = []

    # ------------------------------------------------------------------ streams / clocks
    def _sid(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        h = s.cuda_stream
        if h not in self.names:
            self.names[h] = f"s{len(self.names)}"
        return h

    def _clk(self, sid):
        c = self.clock.get(sid)
        if c is None:
            c = self.clock[sid] = dict(self.host)
        return c

    @staticmethod
    def _join(dst, src):
        for k, v in src.items():
            if dst.get(k, 0) < v:
                dst[k] = v

    def _tick(self, sid):
        c = self._clk(sid)
        self._join(c, self.host)
        c[sid] = c.get(sid, 0) + 1
        return c

    # stream-order events (the patched torch.cuda.Event / Stream / synchronize call these; CPU tests call them directly)
    def on_record(self, ev, sid):
        self.events[id(ev)] = (ev, dict(self._clk(sid)))

    def on_wait(self, ev, sid):
        """sid = the waiting stream, or None when the HOST waits (Event.synchronize)"""
        snap = self.events.get(id(ev))
        if snap is not None:
            self._join(self.host if sid is None else self._clk(sid), snap[1])

    def on_host_sync(self, sid):
        """the host has waited for stream sid (None: for the whole device)"""
        for c in ([self._clk(sid)] if sid is not None else list(self.clock.values())):
            self._join(self.host, c)

    def end_capture(self):
        """The accesses recorded under a capture were checked against each other as they came; they did not EXECUTE on the host's
        timeline (a replay does, on the replaying stream: graph_access).  Forget them, and treat what they touched as written."""
        cid, self.cap_id = self.cap_id, 0
        if not cid:
            return
        for g in self.gens:
            if any(a[8] == cid for a in g.acc):
                g.acc = [a for a in g.acc if a[8] != cid]
                g.opaque = True
                g.has_write = any(a[4] for a in g.acc)
            g.last = {k: v for k, v in g.last.items() if v[2] != cid}
            g.pred = {k: v for k, v in g.pred.items() if v[2] != cid}

    def on_alloc(self, lo, hi, sid, label="alloc"):
        return self._new_gen(lo, hi, sid, True, label)

    # ------------------------------------------------------------------ generations
    def _find(self, addr):
        i = bisect.bisect_right(self.starts, addr) - 1
        if i >= 0 and self.gens[i].lo <= addr < self.gens[i].hi:
            return self.gens[i]
        return None

    def _where(self):
        lab = "/".join(self.labels)
        if not self.stacks:
            return lab
        fr = [f for f in traceback.extract_stack(limit=14) if "propainter_amd" in f.filename and "hazard.py" not in f.filename]
        return lab + " @ " + " < ".join(f"{f.filename.rsplit('/', 1)[-1]}:{f.lineno}" for f in fr[-4:][::-1])

    def _new_gen(self, lo, hi, sid, fresh, label):
        """fresh=True: an allocation was observed -> the generations it overlaps are closed and constrain the new one"""
        i0 = bisect.bisect_right(self.starts, lo) - 1
        if i0 < 0 or self.gens[i0].hi <= lo:
            i0 += 1
        i1 = i0
        pred = {}
        while i1 < len(self.gens) and self.gens[i1].lo < hi:
            g = self.gens[i1]
            if fresh:
                for s_, (n, d, c_) in list(g.pred.items()) + [(s_, v) for s_, v in g.last.items() if s_ not in g.recorded]:
                    if pred.get(s_, (0, None, 0))[0] < n:
                        pred[s_] = (n, d, c_)
            i1 += 1
        if not fresh and i1 > i0:          # a pointer into memory already tracked: no new generation
            return self.gens[i0]
        del self.starts[i0:i1], self.gens[i0:i1]
        self.ngen += 1
        g = _Gen(lo, hi, self.ngen, self.launches, sid, label)
        g.pred = pred
        self.starts.insert(i0, lo)
        self.gens.insert(i0, g)
        return g

    def _storage_range(self, t):
        st = t.untyped_storage()
        return st.data_ptr(), st.data_ptr() + st.nbytes()

    def _view_range(self, t):
        """byte extent [lo, hi) of the elements a (possibly strided) view addresses, and whether it is dense"""
        if t.numel() == 0:
            return None
        es = t.element_size()
        lo = t.data_ptr()
        span = 1 + sum((n - 1) * abs(s) for n, s in zip(t.shape, t.stride()))
        return lo, lo + span * es, span != t.numel()

    # ------------------------------------------------------------------ accesses
    def _report(self, kind, key, msg):
        if key in self.seen:
            return
        self.seen.add(key)
        if len(self.findings[kind]) < self.max_findings:
            self.findings[kind].append(msg)

    def access(self, name, reads, writes, sid=None):
        """reads / writes: lists of (lo, hi, strided) byte ranges (already resolved); one launch on stream sid"""
        sid = self._sid() if sid is None else sid
        self.names.setdefault(sid, f"s{len(self.names)}")
        c = self._tick(sid)
        n = c[sid]
        self.launches += 1
        where = self._where()
        desc = f"{name} [{where}] on {self.names[sid]}#{n}"
        if self.capturing is not None:
            self.capturing["r"].extend((lo, hi) for lo, hi, _ in reads)
            self.capturing["w"].extend((lo, hi) for lo, hi, _ in writes)
            self.capturing["n"] += 1
        # class "intra": ONE launch reads and writes overlapping bytes.  Well defined for element-wise kernels (each thread reads the element
        # it writes); a race for anything that reads neighbours (warps, resamplers, stencils) -- blocks of a launch are unordered.  Listed by
        # kernel name for review; strided channel windows (convolutions) are checked exactly by conv.check_inplace instead.
        for wlo, whi, wstr in writes:
            for rlo, rhi, rstr in reads:
                if rlo < whi and wlo < rhi and not (wstr and rstr):
                    self._report("intra", ("intra", name), f"{desc} reads [{rlo:#x},{rhi:#x}) and writes [{wlo:#x},{whi:#x}) in the same launch")
        for is_write, ranges in ((False, reads), (True, writes)):
            for lo, hi, strided in ranges:
                g = self._find(lo)
                if g is None:
                    g = self._new_gen(lo, hi, sid, False, "pre-existing")
                # class "alias": every access of the closed generations of this range must be visible to this launch
                for s_, (pn, pdesc, _pc) in g.pred.items():
                    if s_ != sid and c.get(s_, 0) < pn:
                        self._report("alias", (name, pdesc.split(" on ")[0], self.names[s_], self.names[sid]),
                                     f"{desc} touches [{lo:#x},{hi:#x}) of a block recycled by the allocator (generation {g.gid}, {g.label}) "
                                     f"without an edge from its previous life's access {pdesc} (sees {self.names[s_]}#{c.get(s_, 0)} < #{pn})")
                # class "race": inside the generation
                if is_write or g.has_write:
                    for (s_, an, alo, ahi, aw, astr, aname, awhere, _cap) in g.acc:
                        if s_ != sid and (aw or is_write) and alo < hi and lo < ahi and c.get(s_, 0) < an:
                            kind = "race?" if (strided and astr) else "race"
                            self._report(kind, (kind, name, aname, self.names[s_], self.names[sid], is_write, aw),
                                         f"{desc} {'writes' if is_write else 'reads'} [{lo:#x},{hi:#x}) while {aname} [{awhere}] on "
                                         f"{self.names[s_]}#{an} {'writes' if aw else 'reads'} [{alo:#x},{ahi:#x}) unordered "
                                         f"(generation {g.gid}, {g.label})")
                # class "uninit": a read of memory nothing has written in this life of the block (the allocator hands out stale bytes:
                # whatever the previous tenant left -- the result then depends on the allocation pattern, not on the inputs)
                if not is_write and not g.opaque and not any(a[4] and a[2] < hi and lo < a[3] for a in g.acc):
                    self._report("uninit", ("uninit", name, g.label.split(" [")[0]),
                                 f"{desc} reads [{lo:#x},{hi:#x}) of generation {g.gid} ({g.label}) that no launch has written")
                g.has_write = g.has_write or is_write
                # keep the list short: an access on the same stream covering an older one of the same kind supersedes it
                g.acc = [a for a in g.acc if not (a[0] == sid and a[4] == is_write and lo <= a[2] and a[3] <= hi)]
                g.acc.append((sid, n, lo, hi, is_write, strided, name, where, self.cap_id))
                g.last[sid] = (n, desc, self.cap_id)

    # engine launches (propainter_amd.hip): pointers are collected by _p / _pw, the launch is closed by _check
    def note(self, t, write=False):
        if t is None or not torch.is_tensor(t) or not t.is_cuda:
            return
        r = self._view_range(t)
        if r is not None:
            lo, hi = self._storage_range(t)
            if self._find(r[0]) is None and hi > lo:
                self._new_gen(lo, hi, self._sid(), False, "pre-existing")
            self.pending.append((r, write))

    def note_range(self, ptr, nbytes, write=False, strided=True):
        """a raw device pointer + extent (ConvArgs: NHWC buffer of N*H*W rows of cstride channels; strided = the launch touches a
        channel window of every row, not the whole rows)"""
        if ptr and nbytes > 0:
            self.pending.append(((int(ptr), int(ptr) + int(nbytes), bool(strided)), write))

    def flush(self, name):
        if not self.pending:
            return
        p, self.pending = self.pending, []
        self.access(name, [r for r, w in p if not w], [r for r, w in p if w])

    # aten kernels
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        try:
            self._aten(func, args, kwargs, out)
        except Exception as e:      # noqa: BLE001 -- a diagnostic must never change the program it watches
            self._report("race?", ("internal", str(func)), f"recorder error in {func}: {type(e).__name__}: {e}")
        return out

    def _aten(self, func, args, kwargs, out):
        flat_in = []
        schema = getattr(func, "_schema", None)
        written = set()

        def walk(v, is_w):
            if torch.is_tensor(v):
                if v.is_cuda and v.device == self.device:
                    flat_in.append(v)
                    if is_w:
                        written.add(id(v))
            elif isinstance(v, (list, tuple)):
                for x in v:
                    walk(x, is_w)
        sargs = schema.arguments if schema is not None else []
        for i, v in enumerate(args):
            ai = sargs[i].alias_info if i < len(sargs) else None
            walk(v, bool(ai is not None and ai.is_write))
        for k, v in kwargs.items():
            ai = next((a.alias_info for a in sargs if a.name == k), None)
            walk(v, bool(ai is not None and ai.is_write))
        outs = []

        def walk_out(v):
            if torch.is_tensor(v):
                if v.is_cuda and v.device == self.device:
                    outs.append(v)
            elif isinstance(v, (list, tuple)):
                for x in v:
                    walk_out(x)
        walk_out(out)
        name = str(func)
        if name.startswith("aten.record_stream"):
            g = self._find(args[0].untyped_storage().data_ptr()) if torch.is_tensor(args[0]) and args[0].is_cuda else None
            if g is not None:
                s = args[1]
                g.recorded.add(s.cuda_stream if hasattr(s, "cuda_stream") else torch.cuda.Stream(stream_id=s.stream_id, device_index=s.device_index,
                                                                                            device_type=s.device_type).cuda_stream)
            return
        if not flat_in and not outs:
            return
        in_bases = {t.untyped_storage().data_ptr() for t in flat_in if t.untyped_storage().nbytes() > 0}
        sid = self._sid()
        fresh = []
        for t in outs:
            st = t.untyped_storage()
            if st.nbytes() == 0:
                continue
            if st.data_ptr() not in in_bases:
                lo, hi = self._storage_range(t)
                g = self._find(lo)
                if g is None or g.lo != lo or g.hi != hi or g.acc:       # (an untouched generation of the same range: the op returned its input's alloc)
                    self._new_gen(lo, hi, sid, True, f"allocated by {name} [{self._where()}]")
                    self.fresh += 1
                fresh.append(t)
        if getattr(func, "is_view", False) or name.startswith(("aten.empty", "aten.new_empty", "aten.empty_like", "aten.empty_strided",
                                                                "aten.detach", "aten.alias", "aten.lift_fresh", "aten._unsafe_view")):
            return                       # no kernel
        reads = [r for r in (self._view_range(t) for t in flat_in if id(t) not in written) if r is not None]
        writes = [r for r in (self._view_range(t) for t in flat_in if id(t) in written) if r is not None]
        writes += [r for r in (self._view_range(t) for t in fresh) if r is not None]
        if reads or writes:
            self.aten_accesses += 1
            self.access(name, reads, writes, sid)
        if name.startswith("aten._local_scalar_dense") or (name.startswith(("aten._to_copy", "aten.copy_")) and not kwargs.get("non_blocking", False)
                                                            and not (len(args) > 2 and args[2] is True)
                                                            and any(torch.is_tensor(v) and not v.is_cuda for v in list(args) + ([out] if torch.is_tensor(out) else []))):
            self._join(self.host, self._clk(sid))        # a blocking device -> host copy: the host has waited for this stream

    # ------------------------------------------------------------------ patches
    def _patch(self, obj, attr, make):
        orig = getattr(obj, attr)
        setattr(obj, attr, make(orig))
        self._patches.append((obj, attr, orig))

    def __enter__(self):
        global _active
        if _active is not None:
            raise RuntimeError("a hazard.Recorder is already installed")
        rec = self

        def ev_record(orig):
            def record(ev, stream=None):
                orig(ev, stream)
                rec.on_record(ev, rec._sid(stream))
            return record

        def ev_wait(orig):
            def wait(ev, stream=None):
                orig(ev, stream)
                rec.on_wait(ev, rec._sid(stream))
            return wait

        def ev_sync(orig):
            def synchronize(ev):
                orig(ev)
                rec.on_wait(ev, None)
            return synchronize

        def st_sync(orig):
            def synchronize(st):
                orig(st)
                rec.on_host_sync(rec._sid(st))
            return synchronize

        def dev_sync(orig):
            def synchronize(device=None):
                orig(device)
                rec.on_host_sync(None)
            return synchronize

        def empty_cache(orig):
            def f():
                orig()                     # releases cached blocks with hipFree: device-synchronising
                rec.on_host_sync(None)
            return f

        def g_begin(orig):
            def capture_begin(g, *a, **k):
                orig(g, *a, **k)
                rec.capturing = {"r": [], "w": [], "n": 0}
                rec.ncap += 1
                rec.cap_id = rec.ncap
            return capture_begin

        def g_end(orig):
            def capture_end(g):
                orig(g)
                cap, rec.capturing = rec.capturing, None
                rec.end_capture()
                if cap is not None:
                    rec.graph_summaries[id(g)] = {"r": _merge(cap["r"]), "w": _merge(cap["w"]), "n": cap["n"], "graph": g}
            return capture_end

        def g_replay(orig):
            def replay(g):
                orig(g)
                s = rec.graph_summaries.get(id(g))
                if s is not None:
                    rec.graph_access(s)
            return replay

        self._patch(torch.cuda.Event, "record", ev_record)
        self._patch(torch.cuda.Event, "wait", ev_wait)
        self._patch(torch.cuda.Event, "synchronize", ev_sync)
        self._patch(torch.cuda.Stream, "synchronize", st_sync)
        self._patch(torch.cuda, "synchronize", dev_sync)
        self._patch(torch.cuda, "empty_cache", empty_cache)
        self._patch(torch.cuda.CUDAGraph, "capture_begin", g_begin)
        self._patch(torch.cuda.CUDAGraph, "capture_end", g_end)
        self._patch(torch.cuda.CUDAGraph, "replay", g_replay)
        _active = self
        return super().__enter__()

    def __exit__(self, *exc):
        global _active
        r = super().__exit__(*exc)
        for obj, attr, orig in reversed(self._patches):
            setattr(obj, attr, orig)
        self._patches = []
        _active = None
        return r

    # ------------------------------------------------------------------ graph launches
    def graph_access(self, summ):
        """one replay of a captured graph on the current stream: its merged read / write intervals against the launches in flight"""
        sid = self._sid()
        c = self._tick(sid)
        n = c[sid]
        self.launches += 1
        inflight = getattr(self, "_graph_launches", [])
        desc = f"graph replay ({summ['n']} launches) [{self._where()}] on {self.names[sid]}#{n}"
        for (s_, an, asumm, adesc) in inflight:
            if s_ == sid or c.get(s_, 0) >= an:
                continue
            for kind, a, b in (("write/write", summ["w"], asumm["w"]), ("write/read", summ["w"], asumm["r"]), ("read/write", summ["r"], asumm["w"])):
                ov = _overlap(a, b)
                if ov is not None:
                    self._report("race", ("graph", id(summ), id(asumm), kind),
                                 f"{desc} and {adesc} are in flight together and {kind} overlap at [{ov[0]:#x},{ov[1]:#x})")
        inflight = [x for x in inflight if not (x[0] == sid)] + [(sid, n, summ, desc)]
        self._graph_launches = inflight[-64:]

    # ------------------------------------------------------------------ result
    def report(self):
        return {"launches": self.launches, "aten_launches": self.aten_accesses, "generations": self.ngen, "allocations_seen": self.fresh,
                "streams": len(self.names),
                "graphs": {str(i): {"launches": s["n"], "read_MB": sum(h - l for l, h in s["r"]) / 1e6, "written_MB": sum(h - l for l, h in s["w"]) / 1e6}
                           for i, s in enumerate(self.graph_summaries.values())},
                **{k: list(v) for k, v in self.findings.items()}}


@contextlib.contextmanager
def label(name):
    """names the part of the pass the launches inside belong to (shows in the findings)"""
    r = _active
    if r is None:
        yield
        return
    r.labels.append(name)
    try:
        yield
    finally:
        r.labels.pop()
