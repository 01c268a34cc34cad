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
import sys

FAMILIES = [   # (substring of the demangled OR mangled kernel name, family)
    ("conv_halo_kernel", "conv_gemm_f16 (LDS-DMA implicit GEMM)"),
    ("conv_ast_kernel", "conv_gemm_f16 (LDS-DMA implicit GEMM)"),
    ("conv_gemm_v2_kernel", "conv_gemm_f16 (LDS-DMA implicit GEMM)"),
    ("conv_dcn_patch_kernel", "conv_gemm_dcn (patch-staged)"),
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
    ("nhwc_to_nchw", "nhwc_to_nchw"),
    ("window_mask", "window_mask"),
]


def family(name):
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
    for f in files:
        for r in csv.DictReader(open(f)):
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
                 f"{total / 1e3:.1f} ms of kernel time)\n\n## By kernel family (bench.py KernelProfiler classes)\n\n")
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
    json.dump({"note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE as reported; "
                       "both in KiB; separate --pmc passes", "families": res}, open(out, "w"), indent=1)
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


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        cmd_counters(sys.argv[2], sys.argv[3], sys.argv[4].split(","))
    elif sys.argv[1] == "stats":
        cmd_stats(sys.argv[2], sys.argv[3])
    else:
        cmd_traffic(sys.argv[2], sys.argv[3], sys.argv[4])
