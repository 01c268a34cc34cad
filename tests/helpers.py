"""Shared test helpers: seeded models, error metrics."""
import os

import numpy as np
import torch

from propainter_amd.synthetic import seeded_weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


from propainter_amd.synthetic import seeded_models  # noqa: E402,F401


def seeded_sds():
    raft, fc, gen = seeded_models("cpu")
    return {"raft": {k: v.float() for k, v in raft.fix_raft.state_dict().items()},
            "fc": {k: v.float() for k, v in fc.state_dict().items()},
            "gen": {k: v.float() for k, v in gen.state_dict().items()}}


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def report(name, got, ref):
    d = (got.double() - ref.double()).abs()
    i = int(d.argmax())
    return (f"{name}: max|d|={d.max().item():.3e} mean|d|={d.mean().item():.3e} ref max={ref.abs().max().item():.3e} "
            f"at flat index {i} (got {got.flatten()[i].item():.5f}, ref {ref.flatten()[i].item():.5f})")
