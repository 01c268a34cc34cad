#!/usr/bin/env python
"""Stand-alone test: a long chain of SMALL dependent kernels on one forked branch of a hipGraph next to LARGE kernels on another --
the ingredient the two mis-behaving graph forms of this engine have in common (profiles/r6_replay_bytes.txt, r6_graph_queues.txt).
No engine code; exact int32 arithmetic that depends on a per-replay counter, so any kernel that runs before its predecessor (or reads a
stale input) changes the checksum.

Per replay `--groups` fork / join groups, as the generator windows are issued: the capture stream produces the group's input, lane A runs
a chain of `--chain` tiny kernels (`--small` int32 elements each) on it, lane B `--big-kernels` kernels over `--big-mb` MB; the
capture stream joins both and folds their sums into the running result, which the next group's input depends on.

    python tools/diag_graph_chain.py [--replays 200] [--groups 16] [--chain 120] [--small 4096] [--big-mb 64] [--big-kernels 12]
Prints `GRAPH_CHAIN {...}`: replays whose result differs from the serial execution of the same program."""
import argparse
import json
import os

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--replays", type=int, default=200)
ap.add_argument("--groups", type=int, default=16)
ap.add_argument("--chain", type=int, default=120)
ap.add_argument("--small", type=int, default=4096)
ap.add_argument("--big-mb", type=int, default=64)
ap.add_argument("--big-kernels", type=int, default=12)
ap.add_argument("--both-chains", action="store_true", help="lane B runs a chain of small kernels too")
args = ap.parse_args()
dev = torch.device("cuda")
MASK = 0xFFFF
nb = args.big_mb * (1 << 20) // 4
counter = torch.zeros(1, dtype=torch.int32, device=dev)
acc = torch.zeros(1, dtype=torch.int32, device=dev)
small = [torch.zeros(args.small, dtype=torch.int32, device=dev) for _ in range(args.groups)]
small_b = [torch.zeros(args.small, dtype=torch.int32, device=dev) for _ in range(args.groups)]
big = [torch.zeros(nb, dtype=torch.int32, device=dev) for _ in range(2)]
ramp_s = torch.arange(args.small, dtype=torch.int32, device=dev) & MASK
ramp_b = torch.arange(nb, dtype=torch.int32, device=dev) & MASK


def chain(t, src, g):
    torch.add(src, acc, out=t)
    t.add_(counter).bitwise_and_(MASK)
    for k in range(args.chain // 2):
        t.mul_(3).add_(k + g).bitwise_and_(MASK)
        t.add_(t.roll(1)).bitwise_and_(MASK)          # (reads a neighbour: a kernel that overtakes its predecessor sees a mix of old and new values)
    return t


def bigwork(t, g):
    torch.add(ramp_b, acc, out=t)
    for k in range(args.big_kernels):
        t.mul_(5).add_(counter).add_(k + g).bitwise_and_(MASK)
    return t


def program(lanes=None, cur=None):
    acc.zero_()
    acc.add_(counter)
    for g in range(args.groups):
        if lanes:
            for ln in lanes:
                ln.wait_stream(cur)
            with torch.cuda.stream(lanes[0]):
                a = chain(small[g], ramp_s, g).sum(dtype=torch.int32)
            with torch.cuda.stream(lanes[1]):
                b = (chain(small_b[g], ramp_s, g + 100) if args.both_chains else bigwork(big[g % 2], g)).sum(dtype=torch.int32)
            for ln in lanes:
                cur.wait_stream(ln)
        else:
            a = chain(small[g], ramp_s, g).sum(dtype=torch.int32)
            b = (chain(small_b[g], ramp_s, g + 100) if args.both_chains else bigwork(big[g % 2], g)).sum(dtype=torch.int32)
        acc.add_(a).add_(b).bitwise_and_(MASK)
    return acc.clone()


lanes = [torch.cuda.Stream(dev) for _ in range(2)]
program()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode="thread_local"):
    cur = torch.cuda.current_stream(dev)
    counter.add_(1)
    out = program(lanes, cur)
torch.cuda.synchronize()
bad = []
for i in range(args.replays):
    graph.replay()
    torch.cuda.synchronize()
    got = int(out.item())
    ref = int(program().item())          # serial, same counter value
    if got != ref and len(bad) < 8:
        bad.append(i)
    elif got != ref:
        bad.append(i)
print("GRAPH_CHAIN " + json.dumps({"replays": args.replays, "replays_differing_from_serial": len(bad), "first": bad[:8], "groups_per_replay": args.groups,
                                   "chain_kernels": args.chain * 2 + 3, "small_elems": args.small, "big_MB": args.big_mb, "big_kernels": args.big_kernels,
                                   "both_chains": args.both_chains, "DEBUG_HIP_FORCE_GRAPH_QUEUES": os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES")}), flush=True)
