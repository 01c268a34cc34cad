"""Host-side logic that needs no GPU: state-dict schemas, C-ABI exports, index tables, weight packing / K-table
emulation, chunk scheduling."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import propainter_oracle as O
from propainter_amd import build as pbuild
from propainter_amd import hip, pipeline
from propainter_amd.conv import pack_weight
from tests.helpers import GOLDEN, seeded_models

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_schemas_match_reference():
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_schema.json")))
    raft, fc, gen = seeded_models("cpu")
    for name, mod in (("raft", raft.fix_raft), ("fc", fc), ("gen", gen)):
        mine = {k: list(v.shape) for k, v in mod.state_dict().items()}
        assert mine == ref[name], (name, set(mine) ^ set(ref[name]))
    assert all(k.startswith("fix_raft.") for k in raft.state_dict())


def test_cabi_exports_every_declared_symbol():
    lib = ctypes.CDLL(pbuild.build(verbose=False))
    header = open(os.path.join(ROOT, "include", "propainter_hip.h")).read()
    names = set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), n
    assert lib.pp_version() >= 100
    assert lib.pp_sizeof_conv_args() == ctypes.sizeof(hip.ConvArgs)
    assert lib.pp_sizeof_attn_args() == ctypes.sizeof(hip.AttnArgs)


def test_cabi_argument_errors_are_reported():
    L = hip.lib()
    a = hip.ConvArgs()
    a.dtype = 7
    assert L.pp_conv2d(ctypes.byref(a), None) == -2           # PP_ERR_DTYPE before any launch
    assert b"dtype" in L.pp_last_error_string()
    with pytest.raises(RuntimeError):
        hip.flow_warp(torch.zeros(1, 4, 4, 8), torch.zeros(1, 4, 4, 2))   # CPU tensors: no fallback


@pytest.mark.parametrize("grid", [(20, 36), (10, 9), (60, 108), (5, 18)])
def test_window_tables_match_reference_roll_semantics(grid):
    own, rolled = hip.window_tables(*grid)
    o2, r2 = O.window_key_index(*grid)
    assert np.array_equal(own, o2.numpy()) and np.array_equal(rolled, r2.numpy()) and rolled.shape[1] == 148


def _emulate_conv(srcs, layer_w, bias, stride, pad, dil, groups, src_channels, pad_mode="zeros"):
    """numpy/torch emulation of the kernel's table-driven gather + packed-weight GEMM (validates pack_weight and
    pp_conv_build_ktable against F.conv2d)."""
    packed, K, cout_g = pack_weight(layer_w, src_channels, groups)
    kh, kw = layer_w.shape[2:]
    taps = [(ky * dil, kx * dil) for ky in range(kh) for kx in range(kw)]
    kt = hip.build_ktable(taps, [(c + 7) // 8 * 8 for c in src_channels])
    assert kt.shape[0] % 8 == 1 and not kt[-1].any(), "kchunks padded to 8 + the trailing 16-byte zero page"
    kt = kt[:-1]
    assert kt.shape[0] * 8 == K
    N, H, W, _ = srcs[0].shape
    OH = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    out = torch.zeros(N, OH, OW, groups * cout_g)
    for g in range(groups):
        A = torch.zeros(N, OH, OW, K)
        for kc, (dy, dx, s, choff) in enumerate(kt):
            s &= 0xff
            if s == 255:
                continue
            src = srcs[s]
            cb = g * src_channels[s] if groups > 1 else 0
            for oy in range(OH):
                for ox in range(OW):
                    iy, ix = oy * stride - pad + dy, ox * stride - pad + dx
                    if pad_mode == "replicate":
                        iy, ix = min(max(iy, 0), H - 1), min(max(ix, 0), W - 1)
                    elif not (0 <= iy < H and 0 <= ix < W):
                        continue
                    A[:, oy, ox, kc * 8:kc * 8 + 8] = src[:, iy, ix, cb + choff:cb + choff + 8]
        out[..., g * cout_g:(g + 1) * cout_g] = A @ packed[g, :cout_g].t()
    return out + bias


@pytest.mark.parametrize("cfg", [dict(cin=[5], cout=12, k=3, stride=2, pad=1, dil=1, groups=1),
                                 dict(cin=[16, 3], cout=10, k=3, stride=1, pad=2, dil=2, groups=1),
                                 dict(cin=[8, 24], cout=16, k=3, stride=1, pad=1, dil=1, groups=2)])
def test_weight_packing_and_ktable_reproduce_conv2d(cfg):
    torch.manual_seed(0)
    N, H, W = 1, 6, 7
    groups = cfg["groups"]
    per_group = cfg["cin"]
    w = torch.randn(cfg["cout"], sum(per_group), cfg["k"], cfg["k"])
    b = torch.randn(cfg["cout"])
    # sources in NHWC with channel padding to 8 (zeros); grouped sources hold groups * per-group channels
    srcs, nchw_groups = [], [[] for _ in range(groups)]
    for c in per_group:
        real = torch.randn(N, groups * c, H, W)
        cp = (c + 7) // 8 * 8 if groups == 1 else c
        t = torch.zeros(N, H, W, max(cp, groups * c))
        t[..., :groups * c] = real.permute(0, 2, 3, 1)
        srcs.append(t)
        for g in range(groups):
            nchw_groups[g].append(real[:, g * c:(g + 1) * c])
    x = torch.cat([torch.cat(parts, 1) for parts in nchw_groups], 1)
    ref = F.conv2d(x, w, b, cfg["stride"], cfg["pad"], cfg["dil"], groups).permute(0, 2, 3, 1)
    got = _emulate_conv(srcs, w, b, cfg["stride"], cfg["pad"], cfg["dil"], groups, per_group)
    assert (got - ref).abs().max() < 1e-4


def test_softsplit_and_ffn_rewrites_are_exact():
    """SoftSplit == 7x7/s3 conv; fc2(gelu(unfold(fold(h)/n))) == conv7x7/s3(gelu(fold(h)/n))  (engine rewrites)."""
    torch.manual_seed(1)
    x = torch.randn(2, 16, 12, 21)
    wt, bs = torch.randn(32, 16 * 49), torch.randn(32)
    ref = F.linear(F.unfold(x, 7, 1, 3, 3).permute(0, 2, 1), wt, bs)
    got = F.conv2d(x, wt.view(32, 16, 7, 7), bs, 3, 3).flatten(2).permute(0, 2, 1)
    assert (ref - got).abs().max() < 1e-4
    fh, fw = O.token_grid(12), O.token_grid(21)
    hdn = torch.randn(2, fh * fw, 5 * 49)
    w2, b2 = torch.randn(8, 5 * 49), torch.randn(8)
    folded = F.fold(hdn.permute(0, 2, 1), (12, 21), 7, 1, 3, 3)
    norm = F.fold(torch.ones_like(hdn).permute(0, 2, 1), (12, 21), 7, 1, 3, 3)
    ref = F.linear(F.gelu(F.unfold(folded / norm, 7, 1, 3, 3).permute(0, 2, 1)), w2, b2)
    got = F.conv2d(F.gelu(folded / norm), w2.view(8, 5, 7, 7), b2, 3, 3).flatten(2).permute(0, 2, 1)
    assert (ref - got).abs().max() < 1e-4


def test_schedule_matches_oracle_driver():
    for L, sub, nl, rs in ((80, 80, 10, 10), (10, 6, 4, 3), (170, 80, 10, 10), (23, 20, 6, 5)):
        ns = nl // 2
        ref_num = sub // rs if L > sub else -1
        sched = pipeline.window_schedule(L, nl, rs, sub)
        for f, (nb, ref) in zip(range(0, L, ns), sched):
            nb2 = list(range(max(0, f - ns), min(L, f + ns + 1)))
            assert nb == nb2 and ref == O.get_ref_index(f, nb2, L, rs, ref_num)
    assert pipeline.subvideo_chunks(170, 80, 5) == [(0, 85, 0, 5), (75, 165, 5, 5), (155, 170, 5, 0)]
    assert [pipeline.raft_clip_length(w) for w in (432, 720, 1280, 1920)] == [12, 8, 4, 2]


def test_models_refuse_cpu_tensors():
    raft, fc, gen = seeded_models("cpu")
    with pytest.raises(RuntimeError):
        raft(torch.zeros(1, 2, 3, 128, 128))
    with pytest.raises(RuntimeError):
        fc(torch.zeros(1, 2, 2, 64, 64), torch.zeros(1, 2, 1, 64, 64))
    with pytest.raises(RuntimeError):
        gen(torch.zeros(1, 2, 3, 64, 64), (torch.zeros(1, 1, 2, 64, 64),) * 2, torch.zeros(1, 2, 1, 64, 64),
            torch.zeros(1, 2, 1, 64, 64), 2)


def _compositor_case(dtype, device):
    """Three overlapping windows (frames visited 1x, 2x and 3x) of random predictions that cover the whole [-1, 1] range
    incl. the exact end points and values whose scaled image lands within one fp16 ulp of an integer."""
    from propainter_amd.pipeline import Compositor
    g = torch.Generator().manual_seed(5)
    L, H, W = 7, 24, 40
    frames = torch.randint(0, 256, (L, H, W, 3), generator=g, dtype=torch.uint8)
    md = (torch.rand(1, L, 1, H, W, generator=g) > 0.4).to(dtype)
    windows = [[0, 1, 2, 3], [2, 3, 4, 5], [3, 4, 5, 6]]
    preds = []
    for nb in windows:
        p = torch.rand(len(nb), 3, H, W, generator=g) * 2 - 1
        p[0, 0, 0, :8] = torch.tensor([-1.0, 1.0, 0.0, 0.999, -0.999, 0.00392, 0.5, -0.5])
        q = torch.randint(0, 256, (H, W), generator=g).float()       # pre-images of integers: (2k/255 - 1) +- rounding
        p[-1, 1] = q * 2 / 255 - 1
        preds.append(p.to(dtype))
    comp = Compositor(frames.to(device), md.to(device))
    ref = [None] * L
    for nb, p in zip(windows, preds):
        comp.add(nb, p.to(device))
        O.composite_window(ref, nb, p, md[0, nb], frames.numpy())
    done = [i for i in range(L) if ref[i] is not None]
    return comp.comp.cpu().numpy()[done], np.stack([ref[i] for i in done])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["f32", "f16"])
def test_compositor_bytes_equal_the_reference_blend(dtype):
    """G12 is byte work: identical float predictions -> identical uint8 composites, in fp32 and in the reference's
    --fp16 arithmetic (scaling in float16), through three overlapping windows (inference_propainter.py:435-450)."""
    got, ref = _compositor_case(dtype, "cpu")
    assert got.dtype == np.uint8 and np.array_equal(got, ref), f"{(got != ref).mean():.3e} of bytes differ"


@pytest.mark.parametrize("cfg", [dict(cout=126, cin=[192, 64], k=(3, 3), groups=1), dict(cout=256, cin=[32, 48], k=(3, 3), groups=8),
                                 dict(cout=128, cin=[128, 128, 5], k=(3, 3), groups=1), dict(cout=512, cin=[40], k=(7, 7), groups=1),
                                 dict(cout=20, cin=[324], k=(1, 1), groups=1), dict(cout=128, cin=[128], k=(3, 3), groups=1, dcn=16)],
                         ids=lambda c: f"c{c['cout']}")
def test_c_weight_packer_equals_the_python_packer(cfg):
    """pp_conv_pack_weight (C, for foreign binders) == propainter_amd.conv.pack_weight on the same K table."""
    import ctypes as C
    from propainter_amd import hip
    from propainter_amd.conv import pack_weight, pad8
    g = torch.Generator().manual_seed(3)
    kh, kw = cfg["k"]
    groups, cin = cfg["groups"], cfg["cin"]
    w = torch.randn(cfg["cout"], sum(cin), kh, kw, generator=g)
    kt = hip.build_ktable([(ky, kx) for ky in range(kh) for kx in range(kw)], [pad8(c) for c in cin], cfg.get("dcn", 0))
    ref, K, cout_g = pack_weight(w, cin, groups, ktable=kt)
    L = hip.lib()
    L.pp_conv_pack_weight.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                      C.c_int, C.c_void_p, C.c_int64]
    sc = (C.c_int32 * len(cin))(*cin)
    wn = np.ascontiguousarray(w.numpy())
    ktn = np.ascontiguousarray(kt)
    kchunks = kt.shape[0] - 1
    cout_pad = L.pp_conv_pack_weight(wn.ctypes.data, cfg["cout"], kh, kw, len(cin), sc, groups, ktn.ctypes.data, kchunks, hip.PP_F32, None, 0)
    assert cout_pad == ref.shape[1] and K == kchunks * 8
    out = np.zeros((groups, cout_pad, K), dtype=np.float32)
    rc = L.pp_conv_pack_weight(wn.ctypes.data, cfg["cout"], kh, kw, len(cin), sc, groups, ktn.ctypes.data, kchunks, hip.PP_F32,
                               out.ctypes.data, out.size)
    assert rc == cout_pad and np.array_equal(out, ref.numpy())
    out16 = np.zeros((groups, cout_pad, K), dtype=np.float16)
    rc = L.pp_conv_pack_weight(wn.ctypes.data, cfg["cout"], kh, kw, len(cin), sc, groups, ktn.ctypes.data, kchunks, hip.PP_F16,
                               out16.ctypes.data, out16.size)
    assert rc == cout_pad and np.array_equal(out16, ref.numpy().astype(np.float16))


def test_shipped_lds_swizzles_are_conflict_free_under_the_lane_group_model():
    """tools/lds_swizzle_check.py: the fragment layouts the kernels ship must be conflict-free under ds_read_b128's lane groups
    (MI355X_MICROARCH.md, LDS) -- the halo patch at EVERY start row, the weight tile at 16-aligned starts; the round-1 patch key
    is kept in the table as the counter-example (2-way conflicts at 24 of 32 alignments)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lds_swizzle_check.py")
    spec = importlib.util.spec_from_file_location("lds_swizzle_check", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = {name: mod.worst_and_bad(fn, starts, kks) for name, fn, starts, kks in mod.CASES}
    shipped = [k for k in res if "[shipped]" in k or "16-aligned" in k or "32-channel" in k]      # (the dcn cases are checked below)
    assert len(shipped) == 3
    for k in shipped:
        assert res[k][0] == 1 and res[k][1] == 0, (k, res[k])
    old = [k for k in res if "[round 1]" in k][0]
    assert res[old][0] == 2 and res[old][1] * 4 == res[old][2] * 3      # 24 of 32 alignments
    # deformable-conv patch rotation: round 2's was 2-way conflicted at every alignment (PMC: 0.50), the shipped one is clean
    dcn_new, dcn_old = [k for k in res if "[shipped dcn]" in k][0], [k for k in res if "[round 2]" in k][0]
    assert res[dcn_new][0] == 1 and res[dcn_new][1] == 0, res[dcn_new]
    assert res[dcn_old][0] == 2 and res[dcn_old][1] == res[dcn_old][2], res[dcn_old]


def test_fork_join_is_sequential_without_a_gpu():
    from propainter_amd import hip
    import torch
    out = hip.fork_join("cpu", [lambda: torch.ones(2), lambda: (torch.zeros(1), torch.ones(1))], 2)
    assert torch.equal(out[0], torch.ones(2)) and isinstance(out[1], tuple)


def test_window_index_cache_is_lru_and_graphs_keep_their_tensors(monkeypatch):
    """pipeline._dev_index: least-recently-used eviction one entry at a time (never a wholesale clear), and every tensor handed out
    while a ClipGraph is being built is also referenced by that graph -- a captured hipGraph replays from those device pointers."""
    from propainter_amd import pipeline as P
    monkeypatch.setattr(P, "_INDEX_CACHE_MAX", 3)
    monkeypatch.setattr(P, "_index_cache", type(P._index_cache)())
    pinned = []
    monkeypatch.setattr(P, "_index_recorder", pinned)
    a = P._dev_index([0, 1, 2], "cpu")
    b = P._dev_index([3, 4], "cpu")
    assert P._dev_index([0, 1, 2], "cpu") is a            # hit: refreshed, handed out again
    monkeypatch.setattr(P, "_index_recorder", None)
    c = P._dev_index([5], "cpu")
    d = P._dev_index([6], "cpu")                          # evicts the least recently used entry: [3, 4]
    keys = [k[0] for k in P._index_cache]
    assert (3, 4) not in keys and (0, 1, 2) in keys and len(keys) == 3
    assert pinned[0] is a and pinned[1] is b and pinned[2] is a and b.tolist() == [3, 4]      # the graph's references survive eviction
    assert P._dev_index([3, 4], "cpu") is not b          # re-created on demand for later eager passes


def test_evaluation_metrics_restate_the_reference():
    """scripts/evaluate_propainter.py: PSNR = core/metrics.py:20-36; SSIM = scikit-image's compare_ssim(data_range=255, multichannel=True,
    win_size=w) as core/metrics.py:47-51 calls it, restated on scipy (scikit-image is absent) -- checked against a brute-force
    evaluation of the published definition (uniform w x w windows fully inside the image, sample covariance, K1 0.01, K2 0.03)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("evaluate_propainter", os.path.join(os.path.dirname(os.path.dirname(__file__)), "scripts", "evaluate_propainter.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    rng = np.random.RandomState(0)
    a = rng.randint(0, 256, (23, 31, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.randint(-20, 21, a.shape), 0, 255).astype(np.uint8)
    assert ev.calculate_psnr(a.astype(np.float64), a.astype(np.float64)) == float("inf")
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    assert abs(ev.calculate_psnr(a.astype(np.float64), b.astype(np.float64)) - 10 * np.log10(255.0 ** 2 / mse)) < 1e-9
    w = 7
    C1, C2, n = (0.01 * 255) ** 2, (0.03 * 255) ** 2, w * w
    vals = []
    for ch in range(3):
        X, Y = a[..., ch].astype(np.float64), b[..., ch].astype(np.float64)
        acc = []
        for y in range(a.shape[0] - w + 1):
            for x in range(a.shape[1] - w + 1):
                px, py = X[y:y + w, x:x + w].ravel(), Y[y:y + w, x:x + w].ravel()
                ux, uy = px.mean(), py.mean()
                vx, vy, vxy = px.var(ddof=1), py.var(ddof=1), ((px - ux) * (py - uy)).sum() / (n - 1)
                acc.append(((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2)))
        vals.append(np.mean(acc))
    assert abs(ev.structural_similarity(a, b, 255, w) - np.mean(vals)) < 1e-9
    assert abs(ev.structural_similarity(a, a, 255, w) - 1.0) < 1e-12
    with pytest.raises(ValueError):
        ev.structural_similarity(a, b, 255, 65)            # the reference's window needs >= 65 pixels per side
    args = ev.build_parser().parse_args([])
    assert (args.height, args.width, args.neighbor_length, args.ref_stride, args.raft_iter, args.task) == (240, 432, 20, 10, 20, "video_completion")


def test_float_blend_compositor_is_the_evaluation_scripts_composite():
    """pipeline.Compositor(float_blend=True) against scripts/evaluate_propainter.py:160-178 of the reference restated in numpy: the
    uint8 prediction pasted inside the mask, frames seen by several windows averaged in float32 with NO truncation in between (the
    inference script's composite truncates after every blend: the two differ by up to 0.75 of a grey level after three windows)."""
    from propainter_amd.pipeline import Compositor
    rng = np.random.RandomState(3)
    L, H, W = 7, 12, 16
    ori = rng.randint(0, 256, (L, H, W, 3)).astype(np.uint8)
    masks = (rng.rand(L, H, W, 1) > 0.5).astype(np.uint8)
    windows = [([0, 1, 2, 3], rng.rand(4, 3, H, W).astype(np.float32) * 2 - 1), ([2, 3, 4, 5], rng.rand(4, 3, H, W).astype(np.float32) * 2 - 1),
               ([3, 4, 5, 6], rng.rand(4, 3, H, W).astype(np.float32) * 2 - 1)]
    comp_ref = [None] * L
    for ids, pred in windows:                                            # (:160-178)
        pi = (torch.from_numpy(pred) + 1) / 2
        pi = pi.permute(0, 2, 3, 1).numpy() * 255
        for i, idx in enumerate(ids):
            img = np.array(pi[i]).astype(np.uint8) * masks[idx] + ori[idx] * (1 - masks[idx])
            comp_ref[idx] = img if comp_ref[idx] is None else comp_ref[idx].astype(np.float32) * 0.5 + img.astype(np.float32) * 0.5
    c = Compositor(torch.from_numpy(ori), torch.from_numpy(masks).permute(0, 3, 1, 2)[None].float(), float_blend=True)
    for ids, pred in windows:
        c.add(ids, torch.from_numpy(pred))
    assert c.comp.dtype == torch.float32
    for idx in range(L):
        assert np.array_equal(c.comp[idx].numpy(), np.asarray(comp_ref[idx], dtype=np.float32)), idx
    assert (c.comp[3] != c.comp[3].floor()).any()                        # frame 3 sits in three windows: quarter levels survive


def _load_script(name):
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", name + ".py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _RecordingRaft:
    """Stands in for RAFT_bi in the host-logic tests: the 'flow' of a pair is (index of its first frame, index of its second frame)."""

    def __init__(self):
        self.calls = []

    def __call__(self, frames, iters=20):
        b, t = frames.shape[:2]
        self.calls.append((t, iters))
        idx = frames[0, :, 0, 0, 0]
        h, w = frames.shape[-2:]
        f = torch.stack([idx[:-1], idx[1:]], 1)[None, :, :, None, None].expand(1, t - 1, 2, h, w)
        return f.clone(), f.flip(2).clone()


def test_flow_scripts_chunk_a_clip_so_that_every_pair_is_computed_once():
    """scripts/compute_flow.py (chunks share their boundary frame) and scripts/evaluate_flow_completion.py (the reference's chunks of
    60 with one frame of overlap, evaluate_flow_completion.py:94-110): every consecutive pair exactly once, in order."""
    cf, ev = _load_script("compute_flow"), _load_script("evaluate_flow_completion")
    t, h, w = 11, 16, 24
    frames = np.zeros((t, h, w, 3), np.uint8)
    frames[:, :, :, :] = (np.arange(t) * 10)[:, None, None, None]
    raft = _RecordingRaft()
    ff, fb = cf.clip_flows(raft, frames, (h, w), "cpu", iters=20, chunk=4)
    assert [c[0] for c in raft.calls] == [4, 4, 4, 2] and ff.shape == fb.shape == (t - 1, h, w, 2)
    val = lambda i: (i * 10 / 255.0) * 2 - 1
    for i in range(t - 1):
        assert abs(ff[i, 0, 0, 0] - val(i)) < 1e-6 and abs(ff[i, 0, 0, 1] - val(i + 1)) < 1e-6
        assert abs(fb[i, 0, 0, 0] - val(i + 1)) < 1e-6 and abs(fb[i, 0, 0, 1] - val(i)) < 1e-6
    x = torch.arange(130, dtype=torch.float32)[None, :, None, None, None].expand(1, 130, 3, 8, 8)
    raft = _RecordingRaft()
    a, b = ev.raft_flows(raft, x, raft_iter=7)
    assert raft.calls == [(60, 7), (61, 7), (11, 7)] and a.shape == (1, 129, 2, 8, 8)
    assert torch.equal(a[0, :, 0, 0, 0], torch.arange(129.)) and torch.equal(a[0, :, 1, 0, 0], torch.arange(1, 130.))
    assert torch.equal(b[0, :, 0, 0, 0], torch.arange(1, 130.))
    raft = _RecordingRaft()
    ev.raft_flows(raft, x[:, :60], raft_iter=20)
    assert raft.calls == [(60, 20)]
    assert cf.frame_stem("00012.jpg") == "00012" and cf.frame_stem("00012.png") == "00012"


def test_flow_colour_coding_and_flow_resize():
    """Middlebury colour wheel of the flow PNGs (55 hues; zero flow white, unknown flow black, hue by direction, saturation by the
    magnitude relative to the frame's largest) and the loader's flow resize (displacements scale with the image, flow_util.py:6-11)."""
    ev = _load_script("evaluate_flow_completion")
    wheel = ev.make_colorwheel()
    assert wheel.shape == (55, 3) and wheel.min() == 0.0 and wheel.max() == 1.0
    assert np.allclose(wheel[0], [1, 0, 0]) and np.allclose(wheel[15], [1, 1, 0]) and np.allclose(wheel[21], [0, 1, 0])
    assert np.allclose(wheel[25], [0, 1, 1]) and np.allclose(wheel[36], [0, 0, 1]) and np.allclose(wheel[49], [1, 0, 1])
    flow = np.zeros((2, 3, 2), np.float32)
    flow[0, 1] = (3.0, 0.0)
    flow[0, 2] = (0.0, 1.5)
    flow[1, 0] = (np.nan, 0.0)
    flow[1, 1] = (1e9, 0.0)
    rgb = ev.flow2rgb(flow)
    assert rgb.shape == (2, 3, 3) and np.allclose(rgb[0, 0], 1.0) and np.allclose(rgb[1, 0], 0.0) and np.allclose(rgb[1, 1], 0.0)
    assert rgb.min() >= 0.0 and rgb.max() <= 1.0
    assert rgb[0, 1].min() < 0.05                      # the largest flow of the frame is fully saturated
    assert 0.45 < rgb[0, 2].min() < 0.55               # half its magnitude: half saturated
    assert not np.allclose(rgb[0, 1], 2 * rgb[0, 2] - 1)      # and a different hue
    fl = torch.zeros(1, 2, 4, 6)
    fl[:, 0], fl[:, 1] = 2.0, -1.0
    out = ev.resize_flows(fl, 8, 18)
    assert out.shape == (1, 2, 8, 18) and torch.allclose(out[:, 0], torch.full((1, 8, 18), 6.0)) and torch.allclose(out[:, 1], torch.full((1, 8, 18), -2.0))
    assert ev.resize_flows(fl, 4, 6) is not None and torch.equal(ev.resize_flows(fl, 4, 6), fl)


def test_resize_u8_linear_is_non_antialiased_bilinear():
    """video_io.resize_u8_linear = cv2.resize(..., INTER_LINEAR) restated (fixed-point weights): within one level of float bilinear
    interpolation WITHOUT antialiasing (core/dataset.py:186-188 shrinks DAVIS frames to 432x240 this way), and clearly different from
    PIL's BILINEAR, which filters over the whole footprint when shrinking (the evaluation inputs)."""
    import torch.nn.functional as F
    from PIL import Image
    from propainter_amd import video_io
    rng = np.random.RandomState(3)
    a = rng.randint(0, 256, (480, 854, 3)).astype(np.uint8)
    for size in ((432, 240), (427, 240), (854, 480), (1000, 600)):
        out = video_io.resize_u8_linear(a, size)
        assert out.shape == (size[1], size[0], 3) and out.dtype == np.uint8
        fl = F.interpolate(torch.from_numpy(a).permute(2, 0, 1)[None].float(), size=(size[1], size[0]), mode="bilinear", align_corners=False)
        assert (torch.from_numpy(out).float() - fl[0].permute(1, 2, 0)).abs().max().item() <= 1.0
    pil = np.asarray(Image.fromarray(a).resize((432, 240), Image.BILINEAR))
    assert np.abs(pil.astype(int) - video_io.resize_u8_linear(a, (432, 240)).astype(int)).max() > 20      # antialiased vs not: not the same image
    half = video_io.resize_u8_linear(a, (427, 240))
    assert np.array_equal(half, ((a[0::2, 0::2].astype(int) + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8))
