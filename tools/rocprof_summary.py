#!/usr/bin/env python
"""Summarises rocprofv3 output directories into the committed profiles/ files.

    rocprof_summary.py stats   <dir> <out.md>            # --kernel-trace --stats run: per-kernel and per-family table
    rocprof_summary.py traffic <fetch_dir> <write_dir> <out.json>   # two --pmc passes (FETCH_SIZE / WRITE_SIZE)

Families group the template instantiations of one kernel (e.g. every conv_gemm_v2_kernel<...> tile) under the class
name bench.py's KernelProfiler uses, so the two can be compared line by line.

HBM traffic (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of the memory-side
request counters; on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes, so wide streaming reads are
DOUBLED here (`fetch_x2`); WRITE_SIZE is uncalibrated and is reported as is.  Units: this rocprofv3 reports both in
kilobytes (TCC_EA0_*REQ * 64 B / 1024)."""
import collections
import csv
import glob
import json
import re
import os
import sys

SPLIT_FAMILY = "conv_gemm_f16x3 (split-plane LDS-DMA implicit GEMM)"
FAMILIES = [   # (substring of the demangled OR mangled kernel name, family)
    ("conv_halo_kernel", "conv_gemm_f16 (LDS-DMA implicit GEMM)"),
    ("conv_ast_kernel", "conv_gemm_f16 (LDS-DMA implicit GEMM)"),
    ("conv_gemm_v2_kernel", "conv_gemm_f16 (LDS-DMA implicit GEMM)"),
    ("conv_dcn_patch_kernel", "conv_gemm_dcn (patch-staged)"),
    ("corr_otf_split_kernel", "corr_lookup_otf_split"),
    ("corr_otf_kernel", "corr_lookup_otf"),
    ("corr_feature_pool", "corr_feature_pyramid"),
    ("conv_gemm_kernel<_Float16", "conv_gemm_f16/dcn (register-staged)"),
    ("conv_gemm_kernelIDF16_", "conv_gemm_f16/dcn (register-staged)"),
    ("conv_gemm_kernel<float", "conv_gemm_f32"),
    ("conv_gemm_kernelIf", "conv_gemm_f32"),
    ("attn_mfma", "sparse_window_attention"),
    ("attn_compact", "sparse_window_attention"),
    ("attn_ref_kernel", "sparse_window_attention (scalar)"),
    ("fold_tokens_kernel", "fold_tokens"),
    ("corr_lookup", "corr_lookup"),
    ("corr_avgpool", "corr_avgpool"),
    ("gru_gate", "gru_gate"),
    ("inorm_", "instance_norm"),
    ("layernorm", "layernorm"),
    ("depthwise_pool", "depthwise_pool"),
    ("upsample2x", "upsample2x"),
    ("dcn_offmask", "dcn_offset_mask_act"),
    ("flow_warp", "flow_warp"),
    ("fb_check", "fb_check"),
    ("img_prop", "img_prop_step"),
    ("convex_upsample", "convex_upsample"),
    ("nchw_to_nhwc", "nchw_to_nhwc"),
    ("pack_nhwc8", "nchw_to_nhwc"),
    ("nhwc_to_nchw", "nhwc_to_nchw"),
    ("window_mask", "window_mask"),
    ("raft_flow_taps", "raft_flow_taps"),
    ("composite_window", "composite_window"),
]


def _template_args(name, kernel):
    """Template arguments of `kernel` in a demangled ("kernel<a, b, ...>(") or Itanium-mangled ("kernelILi8ELb1E...E") name."""
    i = name.find(kernel + "<")
    if i >= 0:
        j = name.find(">", i)
        return [t.strip() for t in name[i + len(kernel) + 1:j].split(",")]
    i = name.find(kernel + "I")
    if i >= 0:
        return ["true" if t == "b1" else "false" if t == "b0" else t[1:] for t in re.findall(r"L([ib]\d+)E", name[i:])]
    return []


# position of the SPLIT template argument (split-plane "f16x3" instantiations): conv_halo_kernel<TH, TW, KH, KW, BN, PROF, STAGGER, WMT,
# SPLIT, PRIVB>, conv_gemm_v2_kernel<BM, BN, BK, WM, WN, STAGES, UNI, CFG, SPLIT, TRI>
_SPLIT_ARG = {"conv_halo_kernel": 8, "conv_gemm_v2_kernel": 8}


def family(name):
    for kernel, pos in _SPLIT_ARG.items():
        if kernel in name:
            args = _template_args(name, kernel)
            if len(args) > pos and args[pos] == "true":
                return SPLIT_FAMILY
    for pat, fam in FAMILIES:
        if pat in name:
            return fam
    return "other (torch elementwise / copies)"


def short(name, n=110):
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= n else name[:n - 3] + "..."


def cmd_stats(root, out):
    rows = collections.OrderedDict()
    files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
    trace = []
    for f in files:
        trace += list(csv.DictReader(open(f)))
    trace.sort(key=lambda r: int(r["Start_Timestamp"]))
    # bench.py --steady-pass: engine-building pass, a 1.5 s pause, the steady-state pass.  Keep the kernels after the LAST pause of
    # >= 1 s between two dispatches (a plain --single-pass trace has no such pause after its first kernels and is kept whole)
    note, cut, last_end = "", 0, None
    for i, r in enumerate(trace):
        if last_end is not None and int(r["Start_Timestamp"]) - last_end >= 1_000_000_000 and i > len(trace) // 3:
            cut = i
        last_end = max(last_end or 0, int(r["End_Timestamp"]))
    if cut:
        note = f"; steady-state pass only: the {cut} dispatches before the marker pause (engine build + first pass) dropped"
        trace = trace[cut:]
    if True:
        for r in trace:
            k = r.get("Kernel_Name", "?")
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3      # us
            a = rows.setdefault(k, [0, 0.0, 1e30, 0.0])
            a[0] += 1
            a[1] += dur
            a[2] = min(a[2], dur)
            a[3] = max(a[3], dur)
    total = sum(v[1] for v in rows.values()) or 1.0
    fam = collections.OrderedDict()
    for k, v in rows.items():
        a = fam.setdefault(family(k), [0, 0.0])
        a[0] += v[0]
        a[1] += v[1]
    with open(out, "w") as fo:
        fo.write(f"# rocprofv3 --kernel-trace summary ({len(files)} trace file(s), {sum(v[0] for v in rows.values())} dispatches, "
                 f"{total / 1e3:.1f} ms of kernel time{note})\n\n## By kernel family (bench.py KernelProfiler classes)\n\n")
        fo.write("| family | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            fo.write(f"| {k} | {v[0]} | {v[1] / 1e3:.2f} | {v[1] / v[0]:.1f} | {100 * v[1] / total:.1f} |\n")
        fo.write("\n## By kernel (top 40)\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
            fo.write(f"| `{short(k)}` | {v[0]} | {v[1] / 1e3:.2f} | {v[1] / v[0]:.1f} | {v[2]:.1f} | {v[3]:.1f} | {100 * v[1] / total:.1f} |\n")
    print(open(out).read()[:3000])


def _counter_sums(root, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg[family(r.get("Kernel_Name", "?"))]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def cmd_traffic(fetch_dir, write_dir, out):
    fe, wr = _counter_sums(fetch_dir, "FETCH_SIZE"), _counter_sums(write_dir, "WRITE_SIZE")
    res = {}
    for fam in sorted(set(fe) | set(wr)):
        nf, f = fe.get(fam, [0, 0.0])
        nw, w = wr.get(fam, [0, 0.0])
        n = max(nf, nw, 1)
        res[fam] = {"launches": n, "fetch_kb_raw_per_launch": f / max(nf, 1), "write_kb_raw_per_launch": w / max(nw, 1),
                    # gfx950 correction: FETCH_SIZE tallies 128-B requests at 64 B -> x2 for wide streaming reads
                    "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0}
    meta = {}
    if len(sys.argv) > 5:         # commit the passes were measured at, its time, RAFT precision of the profiled command
        meta = {"commit": sys.argv[5], "commit_time": int(sys.argv[6]) if len(sys.argv) > 6 and sys.argv[6].isdigit() else 0,
                "raft_dtype": sys.argv[7] if len(sys.argv) > 7 else "f16x3"}
    try:          # the digest of the kernel sources the profiled library was built from (bench.py refuses a profile of other sources)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from propainter_amd import build as _b
        meta["csrc_digest"] = _b.source_digest()
    except Exception as e:      # noqa: BLE001
        meta["csrc_digest"] = None
    json.dump({"note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE as reported; "
                       "both in KiB; separate --pmc passes", **meta, "families": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
        print(f"{k:46s} n={v['launches']:6d}  HBM bytes/launch = {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB")


def cmd_counters(root, out, names):
    """Per-family (and per-kernel for the conv / attention kernels) sums of arbitrary PMC counters of one --pmc pass."""
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    ker = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in names:
                k = r.get("Kernel_Name", "?")
                fam[family(k)][r["Counter_Name"]] += float(r["Counter_Value"])
                ker[short(k, 90)][r["Counter_Name"]] += float(r["Counter_Value"])
    json.dump({"families": fam, "kernels": ker}, open(out, "w"), indent=1)
    for title, d in (("family", fam), ("kernel", ker)):
        print(f"-- per {title}")
        for k, v in sorted(d.items(), key=lambda kv: -kv[1].get(names[-1], 0))[:24]:
            ratio = v.get(names[0], 0) / max(1.0, v.get(names[-1], 0))
            print(f"{k:92s} " + " ".join(f"{n}={v.get(n, 0):.3e}" for n in names) + f"  {names[0]}/{names[-1]}={ratio:.3f}")


def cmd_mfma(root, out, commit=""):
    """--pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE pass taken WITH --kernel-trace: per kernel family the matrix-pipe
    utilisation (busy cycles / (duration x clock x 1024 SIMDs)) at the nominal 2.4 GHz and at the effective clock
    (GRBM_GUI_ACTIVE / duration; the counter is summed over the 8 XCDs)."""
    NSIMD, NOMINAL_HZ, NXCD = 256 * 4, 2.4e9, 8
    dur = {}
    for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r.get("Kernel_Name", "?"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            fm = family(r.get("Kernel_Name", "?"))
            fam[fm][r["Counter_Name"]] += float(r["Counter_Value"])
            did = r.get("Dispatch_Id")
            if did not in seen[fm]:
                seen[fm].add(did)
                fam[fm]["launches"] += 1
                fam[fm]["seconds"] += dur.get(did, ("?", 0.0))[1]
    res = {}
    for k, v in fam.items():
        s = v["seconds"]
        if s <= 0:
            continue
        gui = v.get("GRBM_GUI_ACTIVE", 0.0)
        clk = gui / NXCD / s if gui else None
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        res[k] = {"launches": int(v["launches"]), "ms": s * 1e3, "mfma_busy_cycles": busy,
                  "mfma_ops_f16": v.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0), "grbm_gui_active": gui,
                  "effective_clock_ghz_if_gui_counts_per_xcd": None if clk is None else clk / 1e9,
                  "effective_clock_ghz_if_gui_counts_once": None if not gui else gui / s / 1e9,
                  "mfma_util_at_nominal_2p4ghz": busy / (s * NOMINAL_HZ * NSIMD),
                  "sq_busy_cycles": v.get("SQ_BUSY_CYCLES", 0.0), "sq_wave_cycles": v.get("SQ_WAVE_CYCLES", 0.0)}
    json.dump({"commit": commit, "note": "one rocprofv3 --pmc pass with --kernel-trace over bench.py --single-pass (profiled passes run ~3 % slower "
                                         "than un-profiled ones); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel seconds x 2.4 GHz x 1024 SIMDs): a "
                                         "16x16x32 fp16 MFMA occupies its pipe 16 cycles, which IS the dense peak rate, so the ratio is the fraction "
                                         "of the 2.5 PFLOP/s peak the matrix pipes were busy", "families": res}, open(out, "w"), indent=1)
    print(f"{'family':58s} {'n':>6s} {'ms':>9s} {'MFMA util @2.4GHz':>18s} {'GUI/s/8 GHz':>12s} {'GUI/s GHz':>10s}")
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["ms"]):
        a, b = v["effective_clock_ghz_if_gui_counts_per_xcd"], v["effective_clock_ghz_if_gui_counts_once"]
        print(f"{k:58s} {v['launches']:6d} {v['ms']:9.2f} {v['mfma_util_at_nominal_2p4ghz']:18.3f} {a if a is None else round(a, 3)!s:>12s} {b if b is None else round(b, 3)!s:>10s}")


if __name__ == "__main__":
    if sys.argv[1] == "mfma":
        cmd_mfma(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
    elif sys.argv[1] == "counters":
        cmd_counters(sys.argv[2], sys.argv[3], sys.argv[4].split(","))
    elif sys.argv[1] == "stats":
        cmd_stats(sys.argv[2], sys.argv[3])
    else:
        cmd_traffic(sys.argv[2], sys.argv[3], sys.argv[4])
