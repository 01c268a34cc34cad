"""MI355X-native ProPainter inference path (RAFT -> flow completion -> propagation -> sparse transformer).

Host code is Python on PyTorch-ROCm (device memory, streams); the arithmetic runs in hand-written gfx950 HIP
kernels behind the C-ABI of ``include/propainter_hip.h`` (``propainter_amd/csrc``).  The public surface mirrors
the reference: ``RAFT_bi``, ``RecurrentFlowCompleteNet``, ``InpaintGenerator`` (same constructor / forward
signatures and state-dict keys) and the ``inference_propainter.py`` CLI at the repository root.
"""
__version__ = "0.1.0"
