#!/usr/bin/env python
"""GPU idle time inside a hipGraph replay, from a rocprofv3 --kernel-trace CSV: the union of the kernels' busy intervals over the LAST
`--window-ms` of the trace (a window that lies inside the final replay of `bench.py --steps N`), the idle gaps between them (count,
total, histogram) and the mean number of kernels in flight.  Usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [--window-ms 1400]"""
import argparse
import csv
import glob
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--window-ms", type=float, default=1400.0)
    args = ap.parse_args()
    rows = []
    for f in glob.glob(args.root + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t_end = max(e for _, e, _ in rows)
    t0 = t_end - int(args.window_ms * 1e6)
    win = [(max(s, t0), e) for s, e, _ in rows if e > t0]
    busy, gaps, cur_s, cur_e, inflight = 0, [], None, None, 0
    for s, e in win:
        inflight += e - s
        if cur_e is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
    busy += cur_e - cur_s
    span = t_end - t0
    hist = {"<2us": 0, "2-5us": 0, "5-10us": 0, "10-50us": 0, ">=50us": 0}
    for g in gaps:
        u = g / 1e3
        hist["<2us" if u < 2 else "2-5us" if u < 5 else "5-10us" if u < 10 else "10-50us" if u < 50 else ">=50us"] += 1
    print(json.dumps({"window_ms": span / 1e6, "kernels_in_window": len(win), "busy_fraction": busy / span, "idle_ms": (span - busy) / 1e6,
                      "gaps": len(gaps), "mean_gap_us": (sum(gaps) / len(gaps) / 1e3) if gaps else 0.0, "gap_histogram": hist,
                      "mean_kernels_in_flight": inflight / span}, indent=1))


if __name__ == "__main__":
    main()
