"""Deterministic synthetic inputs and seeded weights.

The reference ships no pretrained weights in the snapshot and there is no network, so
benchmarks and parity tests use seeded weights of the reference architecture.  The fill is
a pure function of (sorted key order, shapes, seed) on the CPU generator, so it reproduces
bit-identically in the authoring container and on the GPU box.
"""
import math

import numpy as np
import torch


# per-model recipes used by the parity tests and the benchmark (see tests/README in DESIGN.md §oracle)
RECIPES = {
    "raft": dict(seed=11, gain=0.7, scales={"update_block.flow_head.conv2.weight": 0.15}),
    "fc": dict(seed=12, gain=1.6),
    "gen": dict(seed=13, gain=1.0),
}


def seeded_weights(kind, template):
    """kind in {'raft','fc','gen'}: the repo-wide deterministic weights for that model."""
    return seeded_state_dict(template, **RECIPES[kind])


def seeded_state_dict(template, seed=2023, gain=1.6, offset_scale=0.4, scales=None):
    """template: mapping name -> tensor (only shape/dtype are read).  Returns a new dict.

    * conv / linear weights (ndim >= 2): U(-a, a), a = sqrt(3 * gain / fan_in) so that
      activations stay O(1) through deep ReLU stacks (the reference's default N(0, 0.02)
      init, model/modules/base_module.py:22-56, gives outputs ~1e-3: poorly conditioned);
    * the zero-initialised last layer of every offset net (``conv_offset.6``,
      model/propainter.py:53-54, model/recurrent_flow_completion.py:27-28) is randomised
      (scaled by ``offset_scale``) so learned offsets / modulation masks are exercised;
    * ``pool_layer`` (depthwise 4x4, init 1/16) is perturbed away from a plain average;
    * norm scales ~U(0.8,1.2), running_var ~U(0.5,1.5) so BatchNorm folding is exercised.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    out = {}
    for name in sorted(template.keys()):
        t = template[name]
        shape = tuple(t.shape)
        if not torch.is_floating_point(t):
            out[name] = t.clone()
            continue
        u = torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            v = 1.0 + 0.5 * u
        elif leaf == "running_mean":
            v = 0.1 * u
        elif "pool_layer" in name and leaf == "weight":
            v = 1.0 / 16 + 0.03 * u
        elif leaf == "weight" and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            a = math.sqrt(3.0 * gain / fan_in)
            if ".conv_offset.6." in name:
                a *= offset_scale
            v = a * u
        elif leaf == "weight":           # 1-D: norm scale
            v = 1.0 + 0.2 * u
        elif leaf == "bias":
            v = 0.05 * u
        else:
            v = 0.1 * u
        for pat, mul in (scales or {}).items():
            if pat in name:
                v = v * mul
        out[name] = v.to(t.dtype)
    return out


def synthetic_clip(length, height, width, seed=2023):
    """smooth random texture translated by (2i, i) px per frame with wrap
    plus N(0, 0.02) noise, quantised to uint8.  Returns uint8 array [L, H, W, 3] (RGB)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(height, dtype=np.float64),
                         np.arange(width, dtype=np.float64), indexing="ij")
    base = np.zeros((height, width, 3), dtype=np.float64)
    for c in range(3):
        for _ in range(8):
            fy = rng.randint(1, 7) * 2 * np.pi / height
            fx = rng.randint(1, 7) * 2 * np.pi / width
            ph = rng.uniform(0, 2 * np.pi)
            amp = rng.uniform(0.03, 0.12)
            base[:, :, c] += amp * np.sin(fy * yy + fx * xx + ph)
    base = base + 0.5
    frames = np.empty((length, height, width, 3), dtype=np.uint8)
    for i in range(length):
        f = np.roll(base, shift=(i % height, (2 * i) % width), axis=(0, 1))
        f = f + rng.normal(0, 0.02, size=f.shape)
        frames[i] = np.clip(f * 255.0, 0, 255).astype(np.uint8)
    return frames


def synthetic_mask(height, width):
    """Static centred rectangle rows [H/3, 2H/3), cols [W/3, 2W/3) (11 % area), uint8 {0,255}."""
    m = np.zeros((height, width), dtype=np.uint8)
    m[height // 3: 2 * height // 3, width // 3: 2 * width // 3] = 255
    return m


def seeded_models(device="cpu", raft_dtype=None, raft_precision=None):
    """The three drop-in modules with the repo-wide seeded weights (the values the goldens were generated with);
    used by the parity tests, smoke() and bench.py (no pretrained checkpoints exist offline)."""
    from .model.modules.flow_comp_raft import RAFT_bi
    from .model.propainter import InpaintGenerator
    from .model.recurrent_flow_completion import RecurrentFlowCompleteNet
    raft = RAFT_bi(model_path=None, device="cpu", compute_dtype=raft_dtype, precision=raft_precision)
    raft.fix_raft.load_state_dict(seeded_weights("raft", raft.fix_raft.state_dict()), strict=True)
    fc = RecurrentFlowCompleteNet()
    fc.load_state_dict(seeded_weights("fc", fc.state_dict()), strict=True)
    gen = InpaintGenerator(init_weights=True)
    gen.load_state_dict(seeded_weights("gen", gen.state_dict()), strict=True)
    return raft.to(device).eval(), fc.to(device).eval(), gen.to(device).eval()
