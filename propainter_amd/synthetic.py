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


# "stress" recipes (round 5): the weights the tame recipes scale down are left at full size -- RAFT's flow head x1.0 (flows of
# tens of pixels after 20 iterations instead of a few: the correlation windows of a tile spread over the map, lookups leave it), the
# offset heads of every deformable alignment x1.0 (offsets up to the 3 / 5 * tanh limit + flow: sampling corners far from the
# mean-shifted patch).  Same seeds, same fill order: only the two scale factors differ.
RECIPES_STRESS = {
    "raft": dict(seed=11, gain=0.7),
    "fc": dict(seed=12, gain=1.6, offset_scale=1.0),
    "gen": dict(seed=13, gain=1.0, offset_scale=1.0),
}


def seeded_weights(kind, template, recipe="tame"):
    """kind in {'raft','fc','gen'}: the repo-wide deterministic weights for that model (recipe "tame": the goldens' weights;
    "stress": RECIPES_STRESS)."""
    return seeded_state_dict(template, **(RECIPES if recipe == "tame" else RECIPES_STRESS)[kind])


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


def stress_clip(length, height, width, seed=2024):
    """Non-tame motion (round 5): two textured layers moving in OPPOSITE directions at 8-48 px/frame (the speed changes every frame)
    and a slow occluder on top -- large, discontinuous, time-varying flow with disocclusions, where synthetic_clip is one rigid
    (2, 1) px/frame translation.  The layers carry fine detail (periods down to ~6 px) so that a 1-px flow error changes pixels.
    Returns uint8 [L, H, W, 3] (RGB)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")

    def texture(nwaves, fmax):
        t = np.zeros((height, width, 3), dtype=np.float64)
        for c in range(3):
            for _ in range(nwaves):
                fy = rng.randint(1, fmax) * 2 * np.pi / height
                fx = rng.randint(1, fmax) * 2 * np.pi / width
                t[:, :, c] += rng.uniform(0.02, 0.09) * np.sin(fy * yy + fx * xx + rng.uniform(0, 2 * np.pi))
        return t + 0.5
    back, front = texture(10, 40), texture(10, 60) * 0.8 + 0.1
    # the front layer covers a band of rows with a wavy edge (a motion boundary every window row crosses)
    band = (yy > height * (0.42 + 0.08 * np.sin(xx * 2 * np.pi / (width / 3.0)))) & (yy < height * 0.82)
    speeds = 8 + 40 * rng.rand(length)                         # px / frame, 8..48
    pos = np.concatenate([[0.0], np.cumsum(speeds)])[:length]
    frames = np.empty((length, height, width, 3), dtype=np.uint8)
    occ_h, occ_w = max(8, height // 6), max(8, width // 10)
    for i in range(length):
        sb = int(round(pos[i]))
        f = np.roll(back, shift=(0, sb % width), axis=(0, 1))
        fr_l = np.roll(front, shift=((3 * i) % height, (-sb) % width), axis=(0, 1))
        bm = np.roll(band, shift=(-sb) % width, axis=1)
        f = np.where(bm[..., None], fr_l, f)
        oy, ox = (height // 5 + 5 * i) % max(1, height - occ_h), (width // 8 + 11 * i) % max(1, width - occ_w)
        f[oy:oy + occ_h, ox:ox + occ_w] = 0.08 + 0.05 * np.sin(0.7 * yy[oy:oy + occ_h, ox:ox + occ_w])[..., None]
        f = f + rng.normal(0, 0.02, size=f.shape)
        frames[i] = np.clip(f * 255.0, 0, 255).astype(np.uint8)
    return frames


def stress_mask(height, width):
    """A border mask a la outpainting (inference_propainter.py:117-156: the canvas grows by 1 / 6 per side, everything outside the
    original frame is hole) plus a lattice of small holes whose pitch is one attention window (5 x 9 tokens = 60 x 108 px), so that
    EVERY window of every transformer layer holds a masked token -- the worst case of the sparse attention (all windows take the
    full key set), where the tame rectangle masks 25 % of the windows.  uint8 {0, 255}; ~35 % of the area."""
    m = np.zeros((height, width), dtype=np.uint8)
    bh, bw = height // 12, width // 12
    m[:bh], m[-bh:], m[:, :bw], m[:, -bw:] = 255, 255, 255, 255
    sq = max(8, min(height, width) // 30)
    for y in range(30, height, 60):
        for x in range(54, width, 108):
            m[y:y + sq, x:x + sq] = 255
    return m


def case_inputs(length, height, width, recipe="tame"):
    """(frames uint8 [L,H,W,3], dilated masks uint8 [L,H,W] {0,255}) of a synthetic case: the clip bench.py times ("tame") or its stress
    leg ("stress"), masks dilated 4x like the driver (--mask_dilation 4, inference_propainter.py:96,105).  The committed goldens
    (tests/golden/synth_*.npz) store SHA-256 digests of exactly these arrays."""
    import scipy.ndimage
    if recipe == "tame":
        clip, m = synthetic_clip(length, height, width), synthetic_mask(height, width)
    else:
        clip, m = stress_clip(length, height, width), stress_mask(height, width)
    m = scipy.ndimage.binary_dilation(m, iterations=4).astype(np.uint8) * 255
    return clip, np.repeat(m[None], length, 0)


def seeded_models(device="cpu", raft_dtype=None, raft_precision=None, recipe="tame"):
    """The three drop-in modules with the repo-wide seeded weights (recipe "tame": the values the goldens were generated with;
    "stress": RECIPES_STRESS); used by the parity tests, smoke() and bench.py (no pretrained checkpoints exist offline)."""
    from .model.modules.flow_comp_raft import RAFT_bi
    from .model.propainter import InpaintGenerator
    from .model.recurrent_flow_completion import RecurrentFlowCompleteNet
    raft = RAFT_bi(model_path=None, device="cpu", compute_dtype=raft_dtype, precision=raft_precision)
    raft.fix_raft.load_state_dict(seeded_weights("raft", raft.fix_raft.state_dict(), recipe), strict=True)
    fc = RecurrentFlowCompleteNet()
    fc.load_state_dict(seeded_weights("fc", fc.state_dict(), recipe), strict=True)
    gen = InpaintGenerator(init_weights=True)
    gen.load_state_dict(seeded_weights("gen", gen.state_dict(), recipe), strict=True)
    return raft.to(device).eval(), fc.to(device).eval(), gen.to(device).eval()
