"""Test infrastructure ONLY (never imported by the product path).

``oracle/`` holds (1) shims that let the *real* reference (``/root/reference``,
pure-Python PyTorch) be imported on CPU in the authoring container, (2) an
independent plain-PyTorch CPU fp32 restatement of the reference algorithm for
the hot path (``propainter_oracle.py``), which *can* travel to the GPU box, and
(3) the script that generated the committed golden vectors (``make_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from here.
"""
