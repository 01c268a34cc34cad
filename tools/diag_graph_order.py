#!/usr/bin/env python
"""Stand-alone test of cross-branch ORDERING in a multi-branch hipGraph -- no engine code, integer arithmetic only.

Why: StreamingClipGraph's stage-pipelined single graph with three side branches deviates from the eager pass at 720x1280x320 although the
host-side happens-before analysis of the submitted program finds nothing (profiles/r6_graph_queues.txt): identical with the graph's branches
forced onto one hardware queue, different on three or more.  This script captures a graph of the SAME SHAPE -- `world` logical ranks x five
stages, stage s of every rank on stage stream s (stages 0..2 on forked side streams as PP_SG_STAGES selects, the rest on the capture stream,
stage 3 forking two lanes like the generator windows), issued in the wavefront order, every dependency an event recorded behind segment
(q, s - 1) and waited for in front of segment (r, s) -- but the segments are chains of exact integer kernels (torch int32 add / and on
`--mb` MB tensors) whose result depends on a per-replay counter, so a consumer that runs before its producer reads the PREVIOUS replay's
values and the final checksum differs from the serial execution of the same program.  Static buffers only (no allocation inside the
capture): the caching allocator is out of the picture.

    python tools/diag_graph_order.py [--replays 50] [--mb 32] [--stages 0,1,2] [--alloc]
    DEBUG_HIP_FORCE_GRAPH_QUEUES=1 python tools/diag_graph_order.py ...

Prints one line `GRAPH_ORDER {...}` with the number of replays whose result differs from the serial one.
--alloc: temporaries are allocated INSIDE the capture (kept alive until it ends), as the engine's segments do."""
import argparse
import json
import os

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--replays", type=int, default=50)
ap.add_argument("--mb", type=int, default=32)
ap.add_argument("--world", type=int, default=4)
ap.add_argument("--stages", default="0,1,2")
ap.add_argument("--alloc", action="store_true")
ap.add_argument("--kernels", default="40,8,2,30,2", help="kernels per segment of stage 0..4")
args = ap.parse_args()
dev = torch.device("cuda")
n = args.mb * (1 << 20) // 4
W, NSEG = args.world, 5
K = [int(v) for v in args.kernels.split(",")]
on_side = {int(v) for v in args.stages.split(",") if v != ""}
MASK = 0xFFFF

counter = torch.zeros(1, dtype=torch.int32, device=dev)
seed = torch.arange(n, dtype=torch.int32, device=dev) & MASK
buf = {(r, s): torch.zeros(n, dtype=torch.int32, device=dev) for r in range(W) for s in range(NSEG)}
tmp = {(r, s, j): torch.zeros(n, dtype=torch.int32, device=dev) for r in range(W) for s in range(NSEG) for j in range(2)}
keep = []


def neighbours(r):
    return [q for q in (r - 1, r + 1) if 0 <= q < W]


def segment(r, s, lanes=None, cur=None):
    """x[r][s] = chain(x[r][s-1] + sum of neighbours' x[q][s-1] + counter); stage 3 splits the tensor over two forked lanes"""
    src = seed if s == 0 else buf[(r, s - 1)]
    nb = [] if s == 0 else [buf[(q, s - 1)] for q in neighbours(r)]
    out = buf[(r, s)]

    def chain(lo, hi, j):
        a = tmp[(r, s, j)][lo:hi] if not args.alloc else torch.empty(hi - lo, dtype=torch.int32, device=dev)
        if torch.cuda.is_current_stream_capturing():
            keep.append(a)
        torch.add(src[lo:hi], counter, out=a)
        for t in nb:
            a.add_(t[lo:hi])
        a.bitwise_and_(MASK)
        for k in range(K[s]):
            a.add_(k + 1 + 3 * r + 7 * s).bitwise_and_(MASK) if k % 2 else a.mul_(3).add_(counter).bitwise_and_(MASK)
        out[lo:hi].copy_(a)

    if s == 3 and lanes:
        h = n // 2
        for ln in lanes:
            ln.wait_stream(cur)
        for j, ln in enumerate(lanes):
            with torch.cuda.stream(ln):
                chain(j * h, (j + 1) * h if j == 0 else n, j)
        for ln in lanes:
            cur.wait_stream(ln)
    else:
        chain(0, n, 0)


def serial():
    for s in range(NSEG):
        for r in range(W):
            segment(r, s)
    return torch.stack([buf[(r, NSEG - 1)].sum(dtype=torch.int64) for r in range(W)]).cpu()


side = [torch.cuda.Stream(dev) for _ in range(3)]
lanes = [torch.cuda.Stream(dev) for _ in range(2)]
graph = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(graph, capture_error_mode="thread_local"):
    cur = torch.cuda.current_stream(dev)
    counter.add_(1)
    stage = [side[i] if i in on_side else cur for i in range(3)] + [cur, cur]
    used = [st for st in side if st in stage]
    for st in used:
        st.wait_stream(cur)
    pending = sorted((r + s, s, r) for r in range(W) for s in range(NSEG))
    issued, done = set(), {}
    while pending:
        for i, (_, s, r) in enumerate(pending):
            if s == 0 or ((r, s - 1) in issued and all((q, s - 1) in issued for q in neighbours(r))):
                break
        else:
            raise RuntimeError("no segment ready")
        pending.pop(i)
        st = stage[s]
        with torch.cuda.stream(st):
            if s > 0 and stage[s - 1] is not st:
                st.wait_event(done[(r, s - 1)])
                for q in neighbours(r):
                    st.wait_event(done[(q, s - 1)])
            segment(r, s, lanes if st is cur else None, cur)
            ev = torch.cuda.Event()
            ev.record(st)
            done[(r, s)] = ev
        issued.add((r, s))
    for st in used:
        cur.wait_stream(st)
    sums = torch.stack([buf[(r, NSEG - 1)].sum(dtype=torch.int64) for r in range(W)])
torch.cuda.synchronize()

bad, first_bad, bad_list = 0, None, []
c0 = int(counter.item())
for i in range(args.replays):
    graph.replay()
    torch.cuda.synchronize()
    got = sums.cpu().clone()
    # the serial execution of the same program at the same counter value (the graph incremented it once)
    cval = int(counter.item())
    ref = serial()
    if not torch.equal(got, ref):
        bad += 1
        if len(bad_list) < 6:
            bad_list.append({"replay": i, "ranks": [int(r) for r in range(W) if got[r] != ref[r]]})
        if first_bad is None:
            first_bad = {"replay": i, "counter": cval, "ranks_differing": [int(r) for r in range(W) if got[r] != ref[r]]}
    # serial() overwrote the buffers with the CURRENT counter's values: a stale read in the next replay still sees other numbers
print("GRAPH_ORDER " + json.dumps({"replays": args.replays, "replays_differing_from_serial": bad, "first": first_bad, "bad": bad_list, "stages_on_branches": sorted(on_side),
                                   "world": W, "MB_per_tensor": args.mb, "kernels_per_segment": K, "alloc_inside_capture": args.alloc,
                                   "DEBUG_HIP_FORCE_GRAPH_QUEUES": os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES"),
                                   "graph_nodes_approx": sum(K) * 2 * W}), flush=True)
