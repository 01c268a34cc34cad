#!/usr/bin/env python
"""Builds the diagnostic VARIANT libraries of the lane-defect hunt (profiles/r6_replay_bytes.txt, calls 6 and 7) -- never the product:

  build/excl/        every launch asks for dynamic LDS = min(63 KB, 159 KB - the kernel's static LDS): no LDS-using block shares a CU with another block
  build/excl_<fam>/  the same for ONE kernel family only (dcn | halo | v2 | rest), the other objects taken from the in-tree build
  build/poison/      the halo kernels fill their whole LDS allocation with fp16 NaNs before staging anything

    python tools/build_lds_variants.py            # needs the in-tree library built (propainter_amd/lib/obj/*.o)
    PP_LIB_PATH=build/excl_halo/libpropainter_hip.so PP_CHAIN_IN_LANES=1 python tools/diag_replay_bytes.py 100 2 1
"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "propainter_amd", "csrc")
OBJ = os.path.join(ROOT, "propainter_amd", "lib", "obj")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include")]
TUS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(CSRC, "*.hip")))

HELPER = '''// [variant build only] every launch asks for enough dynamic LDS that a block of an LDS-using kernel cannot share its CU with any other block
#include <map>
#include <mutex>
template <typename K> static inline unsigned pp_dyn_lds(K kern) {
  static std::mutex mu; static std::map<const void*, unsigned> cache;
  std::lock_guard<std::mutex> g(mu);
  auto it = cache.find((const void*)kern);
  if (it != cache.end()) return it->second;
  hipFuncAttributes a; unsigned pad = 0;
  if (hipFuncGetAttributes(&a, (const void*)kern) == hipSuccess) {
    long v = 159 * 1024 - (long)a.sharedSizeBytes; if (v > 63 * 1024) v = 63 * 1024; if (v < 0) v = 0; pad = (unsigned)v;
  }
  cache[(const void*)kern] = pad; return pad;
}
'''
POISON = '''  // [variant build] poison the whole LDS allocation with fp16 NaNs before anything is staged
  {
    for (int o = tid * 16; o < LDS_BYTES; o += (TH * TW * 128 / WMT) * 16)
      *reinterpret_cast<u32x4*>(lds + o) = u32x4{0x7e007e00u, 0x7e007e00u, 0x7e007e00u, 0x7e007e00u};
    __syncthreads();
  }
'''


def exclusive_launches(text):
    """hipLaunchKernelGGL(kernel, grid, block, 0, stream, ...) -> ... pp_dyn_lds(kernel) ... (launches with their own dynamic LDS are left alone)"""
    out, i, n = [], 0, 0
    while True:
        j = text.find("hipLaunchKernelGGL(", i)
        if j < 0:
            out.append(text[i:])
            return "".join(out), n
        out.append(text[i:j])
        k = j + len("hipLaunchKernelGGL(")
        depth, p, commas = 0, k, []
        while len(commas) < 4:
            c = text[p]
            if c in "(<[":
                depth += 1
            elif c in ")>]":
                depth -= 1
            elif c == "," and depth == 0:
                commas.append(p)
            p += 1
        kern, a3 = text[k:commas[0]].strip(), text[commas[2] + 1:commas[3]].strip()
        if a3 == "0":
            out.append(text[j:commas[2] + 1] + " pp_dyn_lds(" + kern + ")" + text[commas[3]:p])
            n += 1
        else:
            out.append(text[j:p])
        i = p


def compile_all(src_dir, out_dir, names):
    os.makedirs(out_dir, exist_ok=True)
    procs = [(b, subprocess.Popen([HIPCC] + FLAGS + ["-c", os.path.join(src_dir, b + ".hip"), "-o", os.path.join(out_dir, b + ".o")],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)) for b in names]
    for b, pr in procs:
        err = pr.communicate()[1].decode()
        if pr.returncode:
            sys.exit(f"{b}: {err[-2000:]}")


def link(name, variant_objs_dir, from_variant):
    d = os.path.join(ROOT, "build", name)
    os.makedirs(d, exist_ok=True)
    objs = []
    for b in TUS:
        if b in from_variant:
            objs.append(os.path.join(variant_objs_dir, b + ".o"))
        else:
            cand = sorted(glob.glob(os.path.join(OBJ, b + ".*.o")), key=os.path.getmtime)
            if not cand:
                sys.exit("build the in-tree library first (python -c 'from propainter_amd import build; build.build()')")
            objs.append(cand[-1])
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libpropainter_hip.so")] + objs)
    print("built", os.path.join("build", name, "libpropainter_hip.so"))


def main():
    # exclusive-LDS sources
    ex = os.path.join(ROOT, "build", "excl_src")
    shutil.rmtree(ex, ignore_errors=True)
    shutil.copytree(CSRC, ex)
    for f in glob.glob(os.path.join(ex, "*.hip")) + glob.glob(os.path.join(ex, "*.h")):
        t, n = exclusive_launches(open(f).read())
        if n:
            open(f, "w").write(t)
    c = open(os.path.join(ex, "common.h")).read()
    i = c.index("namespace pp")
    open(os.path.join(ex, "common.h"), "w").write(c[:i] + HELPER + c[i:])
    compile_all(ex, os.path.join(ROOT, "build", "excl", "obj"), TUS)
    eo = os.path.join(ROOT, "build", "excl", "obj")
    link("excl", eo, set(TUS))
    fam = {"dcn": {"conv_dcn"}, "halo": {"conv_gemm_v3", "conv_gemm_v3s"}, "v2": {"conv_gemm_v2", "conv_gemm_v2s"}}
    fam["rest"] = set(TUS) - set().union(*fam.values())
    for k, v in fam.items():
        link("excl_" + k, eo, v)
    # LDS-poison sources (halo kernels only)
    po = os.path.join(ROOT, "build", "poison_src")
    shutil.rmtree(po, ignore_errors=True)
    shutil.copytree(CSRC, po)
    h = open(os.path.join(po, "conv_halo.h")).read()
    mark = "  // ---- prologue: patch of block 0 (all pieces) + weights of step 0\n"
    assert mark in h
    open(os.path.join(po, "conv_halo.h"), "w").write(h.replace(mark, POISON + mark, 1))
    compile_all(po, os.path.join(ROOT, "build", "poison"), ["conv_gemm_v3", "conv_gemm_v3s"])
    link("poison", os.path.join(ROOT, "build", "poison"), {"conv_gemm_v3", "conv_gemm_v3s"})


if __name__ == "__main__":
    main()
