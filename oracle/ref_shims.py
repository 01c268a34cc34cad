"""Import the REAL reference (sczhou/ProPainter @ /root/reference) on CPU.

Test infrastructure only.  Three shims, no source edits:
  1. ``torch.__version__`` is patched to a plain ``X.Y.Z`` string while the
     reference is imported, because ``model/misc.py:56-57`` cannot parse
     ``2.10.0+rocm7.0``.
  2. a stub ``cv2`` module (``RAFT/utils/__init__.py:2`` ->
     ``RAFT/utils/frame_utils.py:6-8`` import it; never used on the hot path).
  3. a stub ``torchvision`` exposing ``ops.deform_conv2d`` — the wheel is absent;
     the restatement lives in ``oracle/deform_conv_ref.py`` (semantics pinned by
     known-answer tests, "parity unpinned" at the torchvision boundary).

``/root/reference`` does not exist on the GPU box: nothing imported by the
``-m gpu`` tests, ``smoke()`` or ``bench.py`` may call :func:`load_reference`.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PROPAINTER_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "model", "propainter.py"))


_loaded = None


def load_reference():
    """Returns a namespace with the reference's hot-path classes/functions."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    import torch
    from . import deform_conv_ref

    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.setNumThreads = lambda *a, **k: None
        cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda *a, **k: None)
        sys.modules["cv2"] = cv2
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.ops = types.ModuleType("torchvision.ops")
        tv.ops.deform_conv2d = deform_conv_ref.deform_conv2d
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.ops"] = tv.ops

    real_version = torch.__version__
    torch.__version__ = real_version.split("+")[0]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        ns = types.SimpleNamespace()
        ns.propainter = importlib.import_module("model.propainter")
        ns.rfc = importlib.import_module("model.recurrent_flow_completion")
        ns.flow_comp_raft = importlib.import_module("model.modules.flow_comp_raft")
        ns.sparse_transformer = importlib.import_module("model.modules.sparse_transformer")
        ns.flow_loss_utils = importlib.import_module("model.modules.flow_loss_utils")
        ns.raft = importlib.import_module("RAFT.raft")
        ns.corr = importlib.import_module("RAFT.corr")
        ns.InpaintGenerator = ns.propainter.InpaintGenerator
        ns.RecurrentFlowCompleteNet = ns.rfc.RecurrentFlowCompleteNet
        ns.RAFT = ns.raft.RAFT
        ns.flow_warp = ns.flow_loss_utils.flow_warp
        ns.fbConsistencyCheck = ns.propainter.fbConsistencyCheck
    finally:
        torch.__version__ = real_version
        sys.path.remove(REFERENCE_ROOT)
    _loaded = ns
    return ns


def build_reference_raft():
    """RAFT exactly as ``model/modules/flow_comp_raft.py:10-24`` configures it
    (small=False, mixed_precision=False, alternate_corr=False), without a checkpoint."""
    import argparse
    ns = load_reference()
    args = argparse.Namespace(small=False, mixed_precision=False, alternate_corr=False)
    return ns.RAFT(args).eval()
