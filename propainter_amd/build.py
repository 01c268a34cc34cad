"""Builds libpropainter_hip.so for gfx950 in-tree with hipcc (cross-compiles without a GPU).

Staleness is decided by CONTENT, not by time stamps: the SHA-256 over every ``csrc/*.hip``, ``csrc/*.h`` and
``include/*.h`` is stored next to the library (``lib/build_stamp.txt``) when it is built, and ``hip.lib()`` rebuilds
whenever the sources no longer match it -- an edited kernel or header can never run against a stale binary, and a
snapshot copied to another box (where mtimes mean nothing) is not rebuilt needlessly.  Concurrent builders (the ranks
of a torchrun job on a fresh checkout) are serialised by a file lock; objects and the library are written to
process-private temporaries and moved into place atomically, so a reader never dlopens a half-written file.
"""
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(PKG_DIR, "..", "include")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpropainter_hip.so")
STAMP_PATH = os.path.join(LIB_DIR, "build_stamp.txt")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
if os.environ.get("PP_DIAG") == "1":       # tuning builds: compiles the diagnostic / ablation kernel variants tools/kbench and
    FLAGS.append("-DPP_DIAG")            # tools/bench_dcn.py select through pp_conv_args_t.impl (never in the shipped library)


def sources():
    """Every translation unit of the library (all of csrc/*.hip)."""
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "*.hip")))


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                  glob.glob(os.path.join(INCLUDE, "*.h")))
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    try:
        with open(STAMP_PATH) as f:
            return f.read().strip() != source_digest()
    except OSError:
        return True


def build(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 -O3 -shared -fPIC; objects are compiled in parallel.  Returns the library path."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)          # one builder at a time; the others find the fresh library afterwards
        try:
            if not force and not needs_build():
                return LIB_PATH
            digest = source_digest()
            hipcc = _hipcc()
            tmp = tempfile.mkdtemp(prefix="build_", dir=LIB_DIR)
            try:
                # object cache keyed by content (flags + this source + every header): an edit recompiles only what it touches
                cache = os.path.join(LIB_DIR, "obj")
                os.makedirs(cache, exist_ok=True)
                hdr = hashlib.sha256(" ".join(FLAGS).encode())
                for d in sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))):
                    with open(d, "rb") as f:
                        hdr.update(f.read())
                procs, objs = [], []
                for s in sources():
                    h = hdr.copy()
                    with open(os.path.join(CSRC, s), "rb") as f:
                        h.update(f.read())
                    obj = os.path.join(cache, f"{s[:-4]}.{h.hexdigest()[:16]}.o")
                    objs.append(obj)
                    if os.path.exists(obj):
                        continue
                    for old in glob.glob(os.path.join(cache, f"{s[:-4]}.*.o")):
                        os.unlink(old)
                    obj_tmp = os.path.join(tmp, s.replace(".hip", ".o"))
                    cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", obj_tmp]
                    procs.append((s, obj_tmp, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
                errors = []
                for s, obj_tmp, obj, p in procs:
                    out, _ = p.communicate()
                    if p.returncode != 0:
                        errors.append(f"hipcc failed on {s}:\n{out.decode(errors='replace')}")
                        continue
                    if verbose and out.strip():
                        sys.stderr.write(out.decode(errors="replace"))
                    os.replace(obj_tmp, obj)
                if errors:
                    raise RuntimeError("\n".join(errors))
                so_tmp = os.path.join(tmp, "libpropainter_hip.so")
                r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so_tmp] + objs,
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                if r.returncode != 0:
                    raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
                os.replace(so_tmp, LIB_PATH)
                stamp_tmp = os.path.join(tmp, "stamp")
                with open(stamp_tmp, "w") as f:
                    f.write(digest + "\n")
                os.replace(stamp_tmp, STAMP_PATH)
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
