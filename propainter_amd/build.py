"""Builds libpropainter_hip.so for gfx950 in-tree with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpropainter_hip.so")
SOURCES = ["api.hip", "conv_gemm.hip", "conv_gemm_v2.hip", "conv_gemm_v3.hip", "conv_gemm_ast.hip", "sampling.hip", "raft_ops.hip", "token_ops.hip", "attention.hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_params.h"),
                                                         os.path.join(PKG_DIR, "..", "include", "propainter_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 -O3 -shared -fPIC; objects are compiled in parallel."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objdir = os.path.join(LIB_DIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
    procs = []
    for s in SOURCES:
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, s), "-o", obj]
        procs.append((s, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for s, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode(errors='replace')}")
        if verbose and out.strip():
            sys.stderr.write(out.decode(errors="replace"))
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
