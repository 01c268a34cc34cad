"""Parity of the split-plane ("f16x3") kernels -- RAFT at the reference's fp32 precision class on the fp16 matrix cores
(the reference keeps RAFT fp32 even under --fp16: inference_propainter.py:311) -- against fp64 PyTorch on the very
values the planes represent.  A split-plane tensor carries 22 significand bits per value and every product is
hi*W_hi + lo*W_hi + hi*W_lo with fp32 accumulation, so the bar is fp32-class: 3e-6 of the output range (a plain fp16
layer sits at ~1e-3, the exact fp32 MFMA kernel at ~1e-6)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import propainter_oracle as O
from tests.cpu_emulation import merge_planes, split_planes
from tests.helpers import report

pytestmark = pytest.mark.gpu
RTOL = 3e-6


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from propainter_amd import hip
    hip.lib()
    return torch.device("cuda:0")


def planes(x_nchw, cpad=None):
    """fp32 NCHW (CPU) -> split-plane NHWC on the device, and the fp64 values the planes represent (NCHW, CPU)."""
    t = split_planes(x_nchw.permute(0, 2, 3, 1).contiguous(), cpad).cuda()
    return t, merge_planes(t.cpu(), 0, x_nchw.shape[1]).double().permute(0, 3, 1, 2)


def check(name, got, ref, rtol=RTOL):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    lim = rtol * max(ref.abs().max().item(), 1e-3)
    print(f"SPLIT_PARITY {name}: max|d| {err:.3e} (limit {lim:.3e}, ref max {ref.abs().max().item():.3e})")
    assert math.isfinite(err) and err <= lim, report(name, got.float(), ref.float()) + f" limit {lim:.3e}"


CASES = [
    # halo-tile kernel (3x3 / 1x5 / 5x1 over 64-channel-multiple sources), every cout tile width
    dict(name="halo3x3_c128", cin=[128], cout=128, k=(3, 3), pad=1, act="relu"),
    dict(name="halo3x3_c64_two_src", cin=[192, 64], cout=64, k=(3, 3), pad=1, act="relu"),
    dict(name="halo3x3_c126_window", cin=[192, 64], cout=126, k=(3, 3), pad=1, act="relu", window=128),
    dict(name="halo1x5_c256", cin=[128, 128], cout=256, k=(1, 5), pad=(0, 2), act="sigmoid"),
    dict(name="halo5x1_c128", cin=[128, 128], cout=128, k=(5, 1), pad=(2, 0), act="tanh"),
    dict(name="halo3x3_c2_f32out", cin=[256], cout=2, k=(3, 3), pad=1, out_f32=True),
    dict(name="halo3x3_residual_relu2", cin=[64], cout=64, k=(3, 3), pad=1, act="relu", residual=True, act2="relu"),
    dict(name="halo3x3_linear_residual", cin=[128], cout=128, k=(3, 3), pad=1, residual=True, act2="relu"),
    dict(name="halo3x3_preadd", cin=[128], cout=128, k=(3, 3), pad=1, act="tanh", preadd=True),
    # v2 LDS-DMA kernel: strided 7x7 over the 3-channel image, 96-channel layers, 1x1 over the 324-channel lookup, wide fp32 output
    dict(name="v2_7x7s2_c3", cin=[3], cout=64, k=(7, 7), stride=2, pad=3, out_f32=True, exp_tri=False),            # 8-channel source: plain three-walk format
    dict(name="v2_3x3s2_c96", cin=[64], cout=96, k=(3, 3), stride=2, pad=1, act="relu"),
    dict(name="v2_3x3_c96_residual", cin=[96], cout=96, k=(3, 3), pad=1, act="relu", residual=True, act2="relu", tri=False, exp_tri=False),
    dict(name="v2_1x1_c324", cin=[324], cout=256, k=(1, 1), pad=0, act="relu"),
    dict(name="v2_1x1_576_f32out", cin=[256], cout=576, k=(1, 1), pad=0, out_f32=True, out_scale=0.25),
    dict(name="v2_7x1_c16", cin=[16], cout=128, k=(7, 1), pad=(3, 0), act="relu", exp_tri=False),
    dict(name="v2_1x1_linear_residual_c128", cin=[128], cout=128, k=(1, 1), pad=0, residual=True),
    # halo-eligible layers forced through the v2 kernel's plain split format (every block walked three times)
    dict(name="v2_plain_3x3_c128_preadd", cin=[128], cout=128, k=(3, 3), pad=1, act="tanh", preadd=True, tri=False, exp_tri=False),
    dict(name="v2_plain_3x3_c64_two_src", cin=[192, 64], cout=64, k=(3, 3), pad=1, act="relu", tri=False, exp_tri=False),
    dict(name="v2_plain_1x1_c324", cin=[324], cout=256, k=(1, 1), pad=0, act="relu", tri=False, exp_tri=False),
    # (the other v2_* cases above -- strided 3x3, 1x1 over 324 / 256 / 128 channels -- take the v2 kernel's TRI step; the 324-channel source
    #  has a ragged last block: zero chunks in both halves of its steps, generic per-chunk gather)
    dict(name="v2_tri_3x3s2_c72_ragged", cin=[72], cout=96, k=(3, 3), stride=2, pad=1, act="relu", tri=True),
    # tri-product format over a 96-channel source (32-channel blocks) and 32 + 64 channel sources
    dict(name="halo3x3_c96_src96", cin=[96], cout=96, k=(3, 3), pad=1, act="relu", residual=True, act2="relu"),
    dict(name="halo5x1_c64_src32_64", cin=[32, 64], cout=64, k=(5, 1), pad=(2, 0), act="relu"),
]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_split_plane_conv(dev, case):
    from propainter_amd.conv import ConvLayer, pad8
    g = torch.Generator().manual_seed(11)
    N, H, W = 2, 37, 45
    cin, cout, k = case["cin"], case["cout"], case["k"]
    stride, pad = case.get("stride", 1), case["pad"]
    srcs, vals = zip(*[planes(torch.randn(N, c, H, W, generator=g) * (1.0 + 3.0 * i), pad8(c)) for i, c in enumerate(cin)])
    w = torch.randn(cout, sum(cin), *k, generator=g) / math.sqrt(sum(cin) * k[0] * k[1])
    b = torch.randn(cout, generator=g) * 0.3
    layer = ConvLayer(w, b, stride=stride, padding=pad, src_channels=cin, dtype=torch.float16, device=dev, split=True, tri=case.get("tri"))
    assert layer.kchunks % 8 == 0 and layer.split and layer.tri == case.get("exp_tri", True), (layer.tri, case)
    # the weights the kernel multiplies with: W_hi + W_lo (22 bits of w)
    w_eff = (w.half().double() + (w - w.half().float()).half().double())
    ref = F.conv2d(torch.cat(vals, 1), w_eff, b.double(), stride, pad) * case.get("out_scale", 1.0)
    OH, OW = ref.shape[-2:]
    kw = dict(act=case.get("act"), act2=case.get("act2"), out_scale=case.get("out_scale", 1.0))
    if case.get("preadd"):
        pt, pv = planes(torch.randn(N, cout, OH, OW, generator=g) * 2)
        kw["preadd"] = pt
        ref = ref + pv
    act = {None: lambda v: v, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[case.get("act")]
    ref = act(ref)
    if case.get("residual"):
        rt, rv = planes(torch.randn(N, cout, OH, OW, generator=g) * 2)
        kw["residual"] = rt
        ref = ref + rv
    if case.get("act2") == "relu":
        ref = torch.relu(ref)
    if case.get("out_f32"):
        out = layer(list(srcs), out_dtype=torch.float32, **kw)
        torch.cuda.synchronize()
        assert out.dtype == torch.float32 and out.shape[-1] == pad8(cout)
        got = out[..., :cout]
        assert pad8(cout) == cout or (out[..., cout:] == 0).all()
    elif case.get("window"):
        Cp = case["window"]
        buf = torch.full((N, OH, OW, 2 * Cp), 7.0, dtype=torch.float16, device=dev)
        layer(list(srcs), out=buf, out_choff=0, **kw)
        torch.cuda.synchronize()
        assert (buf[..., cout:Cp] == 7).all() and (buf[..., Cp + cout:] == 7).all(), "channels outside the window must stay untouched"
        got = merge_planes(buf, 0, cout)
    else:
        out = layer(list(srcs), **kw)
        torch.cuda.synchronize()
        assert out.dtype == torch.float16 and out.shape[-1] == 2 * pad8(cout)
        got = merge_planes(out, 0, cout)
    # transcendental epilogues use v_exp / v_rcp forms (1 ulp of fp32 on values <= 1)
    check(case["name"], got.permute(0, 3, 1, 2), ref, RTOL if case.get("act") not in ("tanh", "sigmoid") else 2 * RTOL)


@pytest.mark.parametrize("impl", [0, 12], ids=["halo", "v2"])
@pytest.mark.parametrize("k,pad", [((1, 5), (0, 2)), ((5, 1), (2, 0))], ids=["1x5", "5x1"])
def test_split_plane_fused_gru(dev, k, pad, impl):
    """SepConvGRU half step (RAFT/update.py:45-60) with split-plane state: partial sums as pre-activation addends (both planes
    through the matrix cores in the halo kernel), z / r*h / (1-z)*h + z*q out of the epilogues as hi + lo planes."""
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(55)
    N, H, W, C = 2, 19, 27, 128
    (hd, h0), (inpd, inp), (mfd, mf) = (planes(torch.randn(N, C, H, W, generator=g) * 0.7) for _ in range(3))
    wz, wr, wq = (torch.randn(C, 3 * C, *k, generator=g) / math.sqrt(3 * C * 5) for _ in range(3))
    bz, br, bq = (torch.randn(C, generator=g) * 0.1 for _ in range(3))
    eff = lambda w: w.half().double() + (w - w.half().float()).half().double()
    hx = torch.cat([h0, inp, mf], 1)
    z = torch.sigmoid(F.conv2d(hx, eff(wz), bz.double(), 1, pad))
    r = torch.sigmoid(F.conv2d(hx, eff(wr), br.double(), 1, pad))
    mk = lambda w, b, sc: ConvLayer(w, b, padding=pad, src_channels=sc, dtype=torch.float16, device=dev, split=True, tri=(impl == 0))
    wzr, bzr = torch.cat([wz, wr], 0), torch.cat([bz, br], 0)
    it = lambda w: torch.cat([w[:, :C], w[:, 2 * C:]], 1)
    zr_pre, q_pre = mk(wzr[:, C:2 * C], bzr, [C]), mk(wq[:, C:2 * C], bq, [C])
    zr_it, q_it = mk(it(wzr), None, [C, C]), mk(it(wq), None, [C, C])
    zr_it.impl = q_it.impl = impl
    assert zr_it.tri == (impl == 0)          # halo: tri-product format; v2: every block walked three times
    pzr, pq = zr_pre([inpd]), q_pre([inpd])
    zbuf = torch.empty((N, H, W, 2 * C), dtype=torch.float16, device=dev)
    rh = torch.empty((N, H, W, 2 * C), dtype=torch.float16, device=dev)
    net = hd.clone()
    zr_it([net, mfd], out=zbuf, act="sigmoid", preadd=pzr, fuse=dict(kind="gru_zr", h=net, out2=rh, split=C))
    torch.cuda.synchronize()
    check("z", merge_planes(zbuf).permute(0, 3, 1, 2), z, 2 * RTOL)
    check("r*h", merge_planes(rh).permute(0, 3, 1, 2), r * h0, 2 * RTOL)
    # the q convolution sees the r*h the device produced (22-bit planes): use those values for the reference
    rh_val = merge_planes(rh.cpu()).double().permute(0, 3, 1, 2)
    z_val = merge_planes(zbuf.cpu()).double().permute(0, 3, 1, 2)
    qv = torch.tanh(F.conv2d(torch.cat([rh_val, inp, mf], 1), eff(wq), bq.double(), 1, pad))
    q_it([rh, mfd], out=net, act="tanh", preadd=pq, fuse=dict(kind="gru_h", h=net, z=zbuf))
    torch.cuda.synchronize()
    check("h_new", merge_planes(net).permute(0, 3, 1, 2), (1 - z_val) * h0 + z_val * qv, 2 * RTOL)


def test_split_plane_batched_gemm(dev):
    """All-pairs correlation volume (RAFT/corr.py:52-60) from split-plane feature maps."""
    from propainter_amd.conv import batched_gemm_nt_split
    g = torch.Generator().manual_seed(3)
    B, M, K = 3, 23 * 31, 256
    a = split_planes(torch.randn(B, M, K, generator=g) * 3).cuda()
    b = split_planes(torch.randn(B, M, K, generator=g) * 3).cuda()
    out = batched_gemm_nt_split(a, b, out_scale=1.0 / 16.0)
    torch.cuda.synchronize()
    av, bv = merge_planes(a.cpu()).double(), merge_planes(b.cpu()).double()
    check("volume", out, torch.matmul(av, bv.transpose(1, 2)) / 16.0)


@pytest.mark.parametrize("impl", [0, 12, 13])
def test_split_plane_correlation_pyramid_from_pooled_features(dev, impl):
    """Levels 1..3 of the correlation pyramid (RAFT/corr.py:21-27: avg_pool2d of the volume, floor sizes) as GEMMs of f1 with the
    pooled split-plane features (pp_corr_feature_pyramid_split; pooling is linear) against fp64 pooling of the fp64 volume --
    level sizes that are no multiples of 16 rows (22 x 30 -> 11 x 15 = 165, 5 x 7 = 35, 2 x 3 = 6: padded weight rows, ragged
    cout tiles, rows of 4-byte alignment only) through every tile configuration of the volume GEMM."""
    from propainter_amd import hip
    from propainter_amd.conv import batched_gemm_nt_split
    g = torch.Generator().manual_seed(11)
    P, h, w = 2, 22, 30
    f1 = split_planes(torch.randn(P, h, w, 256, generator=g) * 2).cuda()
    f2 = split_planes(torch.randn(P, h, w, 256, generator=g) * 2).cuda()
    lv = hip.corr_feature_pyramid_split(f2)
    f1v = merge_planes(f1.cpu()).double().view(P, h * w, 256)
    ref = (torch.matmul(f1v, merge_planes(f2.cpu()).double().view(P, h * w, 256).transpose(1, 2)) / 16.0).view(P * h * w, 1, h, w)
    vol = batched_gemm_nt_split(f1.view(P, h * w, 512), f2.view(P, h * w, 512), out_scale=1.0 / 16.0, impl=impl)
    torch.cuda.synchronize()
    check("level0", vol.view(P * h * w, 1, h, w), ref)
    for l, fl in enumerate(lv, start=1):
        ref = F.avg_pool2d(ref, 2, 2)
        assert fl.shape == (P, h >> l, w >> l, 512)
        got = batched_gemm_nt_split(f1.view(P, h * w, 512), fl.view(P, -1, 512), out_scale=1.0 / 16.0, impl=impl)
        torch.cuda.synchronize()
        check(f"level{l}", got.view(P * h * w, 1, h >> l, w >> l), ref, 2 * RTOL)


def test_split_plane_aux_ops(dev):
    """Split-plane outputs of the correlation lookup, the flow-tap gather, the NCHW packer and the fused InstanceNorm tail
    against their fp32 forms / torch."""
    from propainter_amd import hip
    g = torch.Generator().manual_seed(9)
    P, h, w = 2, 16, 24
    # ---- correlation lookup: split output == fp32 output to 2^-22
    f1, f2 = torch.randn(P, h * w, 256, generator=g), torch.randn(P, h * w, 256, generator=g)
    vol = (torch.matmul(f1, f2.transpose(1, 2)) / 16.0).cuda()
    levels = [vol.view(P * h * w, h, w)]
    hh, ww = h, w
    for _ in range(3):
        levels.append(hip.corr_avgpool(levels[-1], P * h * w, hh, ww))
        hh, ww = hh // 2, ww // 2
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None] + torch.randn(P, h, w, 2, generator=g) * 3).contiguous().cuda()
    ref32 = hip.corr_lookup(levels, coords, torch.empty((P, h, w, 328), dtype=torch.float32, device=dev))
    sp = hip.corr_lookup(levels, coords, torch.full((P, h, w, 656), 9.0, dtype=torch.float16, device=dev), split=True)
    torch.cuda.synchronize()
    check("corr_lookup", merge_planes(sp, 0, 324), ref32[..., :324], 1e-6)
    assert (sp[..., 324:328] == 0).all() and (sp[..., 328 + 324:] == 0).all()
    # ---- flow taps
    c0 = torch.stack([xs, ys], -1)[None].expand(P, h, w, 2).contiguous().cuda()
    rows32 = hip.raft_flow_taps(coords, c0, torch.empty((P, h, w, 16), dtype=torch.float32, device=dev))
    xbuf = torch.full((P, h, w, 256), 5.0, dtype=torch.float16, device=dev)
    rows = hip.raft_flow_taps(coords, c0, torch.empty((P, h, w, 32), dtype=torch.float16, device=dev), flow_out=xbuf, flow_choff=126, split=True)
    torch.cuda.synchronize()
    check("flow_taps", merge_planes(rows), rows32, 1e-6)
    check("flow_window", merge_planes(xbuf, 126, 2), (coords - c0), 1e-6)
    assert (xbuf[..., :126] == 5).all() and (xbuf[..., 128:254] == 5).all()
    # ---- NCHW packer
    img = torch.rand(3, 3, 40, 56, generator=g) * 2 - 1
    x = hip.nchw_to_nhwc(img.cuda(), cpad=8, split=True)
    torch.cuda.synchronize()
    assert x.shape == (3, 40, 56, 16) and (x[..., 3:8] == 0).all() and (x[..., 11:] == 0).all()
    check("nchw_to_nhwc", merge_planes(x, 0, 3).permute(0, 3, 1, 2), img, 1e-6)
    # ---- InstanceNorm + residual tail
    v = torch.randn(2, 33, 47, 64, generator=g) * 2 + 0.5
    res_t, res_v = planes(torch.randn(2, 64, 33, 47, generator=g))
    y = hip.instance_norm_split(v.cuda(), relu=True, residual=res_t, relu2=True)
    y0 = hip.instance_norm_split(v.cuda(), relu=False)
    torch.cuda.synchronize()
    inorm = F.instance_norm(v.double().permute(0, 3, 1, 2), eps=1e-5)
    check("instance_norm_split", merge_planes(y0).permute(0, 3, 1, 2), inorm, 2e-6)
    check("instance_norm_split + residual", merge_planes(y).permute(0, 3, 1, 2), torch.relu(torch.relu(inorm) + res_v), 2e-6)


def test_split_plane_raft_matches_reference_golden():
    """RAFT_bi(precision="f16x3") end to end against the REAL reference's golden flows (tests/golden/raft_128x192.npz,
    oracle/make_golden.py): fp32-class end-point error."""
    from tests.helpers import load_golden, seeded_models
    raft = seeded_models("cuda")[0]
    g = load_golden("raft_128x192.npz")
    fr = torch.from_numpy(g["frames_u8"]).permute(0, 3, 1, 2).float().div(255)[None] * 2 - 1
    raft.precision = "f16x3"
    ff, fb = raft(fr.cuda(), iters=int(g["iters"]))
    ff2, fb2 = raft(fr.cuda(), iters=int(g["iters"]), streams=2)
    torch.cuda.synchronize()
    assert torch.equal(ff, ff2) and torch.equal(fb, fb2), "pair groups on two streams must give identical flows"
    ef = (ff[0].cpu() - torch.from_numpy(g["flows_f"])).pow(2).sum(1).sqrt()
    eb = (fb[0].cpu() - torch.from_numpy(g["flows_b"])).pow(2).sum(1).sqrt()
    print(f"SPLIT_PARITY raft golden EPE: fw mean {ef.mean():.2e} max {ef.max():.2e}, bw mean {eb.mean():.2e} max {eb.max():.2e}")
    assert ef.mean() < 5e-5 and ef.max() < 2e-3 and eb.mean() < 5e-5 and eb.max() < 2e-3


@pytest.mark.parametrize("split", [False, True], ids=["f16", "split"])
@pytest.mark.parametrize("cin,cout,act,out_f32", [(256, 2, None, True), (64, 3, "tanh", False)], ids=["flow_head", "rgb_head"])
def test_streaming_head_kernel(dev, cin, cout, act, out_f32, split):
    """conv_head.hip (impl 110; opt-in: measured neutral against the 16-cout MFMA tiles, profiles/r3l_head_kernel_ab.txt): 3x3 heads with
    <= 4 couts as v_dot2 dot products over an LDS halo patch, plain fp16 and split-plane tri-product form, ragged tiles (37 x 45 map)."""
    from propainter_amd.conv import ConvLayer, pad8
    if split and not out_f32:
        pytest.skip("split-plane heads write plain fp32")
    g = torch.Generator().manual_seed(21)
    N, H, W = 2, 37, 45
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.3
    fn = {None: lambda v: v, "tanh": torch.tanh}[act]
    if split:
        xt, xv = planes(x)
        w_eff = w.half().double() + (w - w.half().float()).half().double()
        ref = fn(F.conv2d(xv, w_eff, b.double(), 1, 1))
        layer = ConvLayer(w, b, padding=1, src_channels=[cin], dtype=torch.float16, device=dev, split=True)
        tol = RTOL
    else:
        xt = x.permute(0, 2, 3, 1).contiguous().half().cuda()
        ref = fn(F.conv2d(xt.float().cpu().permute(0, 3, 1, 2).double(), w.half().double(), b.double(), 1, 1))
        layer = ConvLayer(w, b, padding=1, src_channels=[cin], dtype=torch.float16, device=dev)
        tol = 2e-3
    layer.impl = 110
    out = layer([xt], act=act, out_dtype=torch.float32 if out_f32 else None)
    layer.impl = 0
    base = layer([xt], act=act, out_dtype=torch.float32 if out_f32 else None)
    torch.cuda.synchronize()
    assert out.shape[-1] == pad8(cout) and (out[..., cout:] == 0).all()
    check(f"head_{cin}_{cout}", out[..., :cout].permute(0, 3, 1, 2), ref, tol)
    check(f"head_vs_mfma_{cin}_{cout}", out[..., :cout].permute(0, 3, 1, 2), base[..., :cout].permute(0, 3, 1, 2), tol)


# ---- range stress: the planes' quantisation INCLUDED in the reference -------------------------------
# A split-plane value is hi = fp16(v), lo = fp16(v - hi).  While lo is a NORMAL fp16 number (|v| >= 2^-3) the pair carries 22
# significand bits; below that lo falls into fp16's subnormal range (quantum 2^-24) and the pair has an ABSOLUTE resolution of
# 2^-25; |v| > 65504 overflows the hi plane to inf (the documented failure: non-finite output, caught by RAFT_bi's finite-flow
# guard).  Error model per operand: |dv| <= 2^-22 |v| + 2^-25.  A product drops lo*lo (<= 2^-22 |x||w|) and the K-term sum is
# accumulated in fp32 (sqrt(K) x 2^-24 of the absolute sum; bias add and result: 2^-24 of the output).  BOUND below is that model summed over the taps -- elementwise, against fp64 on the TRUE fp32 inputs and
# weights (not on the values the planes happen to represent, as the tests above do).
def _stress_values(shape, g, lo_exp, hi_exp):
    """sign * 10^U(lo_exp, hi_exp): log-uniform magnitudes."""
    mag = torch.pow(10.0, torch.rand(shape, generator=g) * (hi_exp - lo_exp) + lo_exp)
    return mag * (torch.randint(0, 2, shape, generator=g).float() * 2 - 1)


STRESS_RANGES = {"wide": (-6.0, 4.3), "tiny": (-6.0, -3.0), "huge": (3.0, 4.78)}      # 10^4.78 = 60 256 < 65 504
STRESS_LAYERS = [
    dict(name="halo3x3_c128_tri", cin=[128], cout=128, k=(3, 3), pad=1),
    dict(name="halo1x5_c256_two_src_tri", cin=[128, 128], cout=256, k=(1, 5), pad=(0, 2)),
    dict(name="v2_1x1_c324_tri", cin=[324], cout=256, k=(1, 1), pad=0),
    dict(name="v2_7x7s2_c3_plain", cin=[3], cout=64, k=(7, 7), stride=2, pad=3),
    dict(name="v2_3x3_c128_plain", cin=[128], cout=128, k=(3, 3), pad=1, tri=False),
]


@pytest.mark.parametrize("rng", sorted(STRESS_RANGES))
@pytest.mark.parametrize("case", STRESS_LAYERS, ids=[c["name"] for c in STRESS_LAYERS])
def test_split_plane_range_stress(dev, case, rng):
    from propainter_amd.conv import ConvLayer, pad8
    g = torch.Generator().manual_seed(101)
    N, H, W = 1, 24, 40
    cin, cout, k = case["cin"], case["cout"], case["k"]
    stride, pad = case.get("stride", 1), case["pad"]
    xs = [_stress_values((N, c, H, W), g, *STRESS_RANGES[rng]) for c in cin]
    K = sum(cin) * k[0] * k[1]
    # weights log-uniform over three decades around 1/sqrt(K): small enough that lo planes of most weights are subnormal
    w = _stress_values((cout, sum(cin), *k), g, -2.5, 0.0) / math.sqrt(K)
    b = torch.randn(cout, generator=g) * 0.3
    srcs = [split_planes(x.permute(0, 2, 3, 1).contiguous(), pad8(c)).cuda() for x, c in zip(xs, cin)]
    layer = ConvLayer(w, b, stride=stride, padding=pad, src_channels=cin, dtype=torch.float16, device=dev, split=True, tri=case.get("tri"))
    out = layer(srcs, out_dtype=torch.float32)             # plain fp32 output: the huge range must not overflow an OUTPUT plane
    torch.cuda.synchronize()
    got = out[..., :cout].permute(0, 3, 1, 2).double().cpu()
    x64, w64 = torch.cat(xs, 1).double(), w.double()
    ref = F.conv2d(x64, w64, b.double(), stride, pad)
    e22, e25, e24 = 2.0 ** -22, 2.0 ** -25, 2.0 ** -24
    ax, aw = x64.abs(), w64.abs()
    bound = (F.conv2d(ax * e22 + e25, aw, None, stride, pad) + F.conv2d(ax, aw * e22 + e25, None, stride, pad)
             + e22 * F.conv2d(ax, aw, None, stride, pad) + 2 * e24 * math.sqrt(K) * F.conv2d(ax, aw, None, stride, pad)
             + 2 * e24 * ref.abs())                          # the bias add and the fp32 result itself round at 2^-24 of the output
    err = (got - ref).abs()
    assert torch.isfinite(got).all()
    ratio = (err / bound).max().item()
    rel_rng = err.max().item() / ref.abs().max().item()
    print(f"SPLIT_RANGE_STRESS {case['name']} [{rng}]: max err/bound {ratio:.3f}, max|d| {err.max().item():.3e} = {rel_rng:.2e} of the output range "
          f"(|x| in 1e{STRESS_RANGES[rng][0]:+.1f}..1e{STRESS_RANGES[rng][1]:+.1f})")
    assert ratio <= 1.0, (ratio, rel_rng)
    # the fp32-class claim in the units the other tests use: <= 3e-6 of the output range whenever the INPUT planes are normal numbers
    if rng != "tiny":
        assert rel_rng <= 3e-6, rel_rng


def test_split_plane_overflow_is_loud(dev):
    """|v| > 65504 cannot be represented: the hi plane is inf, every output that touches it is non-finite (never a silently wrong
    finite number), and RAFT_bi's finite-flow guard (flow_comp_raft.py) turns that into an error naming precision='f32'."""
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 16, 24, generator=g)
    x[0, 3, 8, 12] = 7.0e4
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    layer = ConvLayer(w, None, padding=1, src_channels=[64], dtype=torch.float16, device=dev, split=True)
    out = layer([split_planes(x.permute(0, 2, 3, 1).contiguous()).cuda()], out_dtype=torch.float32)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(out[0, :, :, :64]).all(-1).cpu()
    assert bad[7:10, 11:14].all() and bad.sum() == 9, "exactly the 3x3 neighbourhood of the overflowing pixel is non-finite"
    from propainter_amd.model.modules.flow_comp_raft import assert_finite_flows
    from tests.helpers import seeded_models
    raft = seeded_models("cuda")[0]
    raft.precision = "f16x3"
    fr = torch.rand(1, 2, 3, 128, 192) * 2 - 1
    fr[0, 0, 0, 50, 60] = 1e9                       # far outside [-1, 1]: the encoder's first split-plane activation overflows
    raft(fr.cuda(), iters=2)
    with pytest.raises(FloatingPointError, match="f32"):
        assert_finite_flows(raft)
    raft(torch.rand(1, 2, 3, 128, 192).cuda() * 2 - 1, iters=2)
    assert_finite_flows(raft)                        # a clean pass clears the flag


@pytest.mark.parametrize("case", ["smooth", "ragged_oob", "divergent", "chaotic"])
def test_split_plane_corr_lookup_on_the_fly(dev, case):
    """pp_corr_lookup_otf_split (no all-pairs volume, fp32-class: RAFT/corr.py:13-60) against (i) fp64 volume -> avg-pool pyramid ->
    bilinear lookup on the values the feature planes represent and (ii) the volume path of the same engine (batched tri-product GEMMs +
    pp_corr_lookup), whose blend arithmetic it repeats.  Cases as in tests/test_ops_gpu.py: smooth flow (one shared box per 8x8 tile),
    ragged map + coordinates far outside / on exact integers, a flow diverging inside tiles (quadrant fallback), per-pixel random targets
    (single-pixel fallback).  Output layout: 88 channels per level and plane (81 taps + 7 zeros)."""
    from propainter_amd import hip
    from propainter_amd.conv import batched_gemm_nt_split
    g = torch.Generator().manual_seed({"smooth": 1, "ragged_oob": 2, "divergent": 3, "chaotic": 4}[case])
    P, h, w = (2, 24, 40) if case != "ragged_oob" else (2, 19, 29)
    f1 = split_planes(torch.randn(P, h, w, 256, generator=g) * 2).cuda()
    f2 = split_planes(torch.randn(P, h, w, 256, generator=g) * 2).cuda()
    base = O.coords_grid(P, h, w)
    if case == "smooth":
        coords = base + torch.tensor([1.7, -2.3]).view(1, 2, 1, 1) + 0.3 * torch.randn(P, 2, h, w, generator=g)
    elif case == "ragged_oob":
        coords = base + torch.randn(P, 2, h, w, generator=g) * 1.5
        coords[0, :, 0, 0] = torch.tensor([-30.3, 2.2]); coords[0, :, 0, 1] = torch.tensor([300.0, 20.5])
        coords[0, :, 1, :] = base[0, :, 1, :] + 3.0                      # exact integers: 1-ulp round-trip effects
        coords[1, :, 5, 5] = torch.tensor([-4.0, -4.0]); coords[1, :, 6, 6] = torch.tensor([w + 3.5, h + 3.5])
        coords[1, :, 7, 7] = torch.tensor([-5.5, 3.0])
    elif case == "divergent":
        coords = base * 1.6 - 4.0 + torch.randn(P, 2, h, w, generator=g)
    else:
        coords = torch.rand(P, 2, h, w, generator=g) * torch.tensor([w * 1.2, h * 1.2]).view(1, 2, 1, 1) - 2.0
    cd = coords.permute(0, 2, 3, 1).contiguous().cuda()
    lv = [f2] + hip.corr_feature_pyramid_split(f2)
    LV = hip.OTF_SPLIT_LEVEL_CHANNELS
    out = torch.full((P, h, w, 8 * LV), 7.0, dtype=torch.float16, device=dev)
    hip.corr_lookup_otf_split(f1, lv, cd, out)
    again = torch.empty_like(out)
    hip.corr_lookup_otf_split(f1, lv, cd, again)
    torch.cuda.synchronize()
    assert torch.equal(out, again), "the on-the-fly lookup is not run-to-run deterministic"
    pl = out.view(P, h, w, 2, 4, LV)
    assert (pl[..., 81:] == 0).all(), "pad channels of every level group must be zero"
    got = (pl[..., :81].float().sum(3)).reshape(P, h, w, 324)            # hi + lo, channel l*81 + a*9 + b
    # (i) fp64 on the represented values (the fp32 oracle's lookup on fp64 tensors: its coordinate arithmetic then runs in fp64 -- the
    #     per-tap fp32 round trip is what (ii) checks)
    f1v = merge_planes(f1.cpu()).double().permute(0, 3, 1, 2)
    f2v = merge_planes(f2.cpu()).double().permute(0, 3, 1, 2)
    ref = O.corr_lookup(O.corr_pyramid(f1v, f2v), coords.double())
    check(f"otf_split_vs_fp64[{case}]", got.permute(0, 3, 1, 2), ref, 2e-5)    # coordinate rounding (fp32 vs fp64 taps) dominates: ~1e-6 px x gradient
    # (ii) the volume path of the same engine: same products (tri-product, fp32 accumulation in a different order), same blend arithmetic
    n8 = h * w
    gemm = lambda b_: batched_gemm_nt_split(f1.view(P, n8, 512), b_.view(P, -1, 512), out_scale=1.0 / 16.0)
    levels = [gemm(t).view(P * n8, t.shape[1], t.shape[2]) for t in lv]
    vol = hip.corr_lookup(levels, cd, torch.empty((P, h, w, 656), dtype=torch.float16, device=dev), split=True)
    torch.cuda.synchronize()
    check(f"otf_split_vs_volume[{case}]", got, merge_planes(vol, 0, 324), 2e-6)
    # batch invariance: pair 1 alone gives the same bytes as inside the batch of 2
    solo = torch.empty((1, h, w, 8 * LV), dtype=torch.float16, device=dev)
    hip.corr_lookup_otf_split(f1[1:].contiguous(), [t[1:].contiguous() for t in lv], cd[1:].contiguous(), solo)
    torch.cuda.synchronize()
    assert torch.equal(solo[0], out[1])


def test_split_plane_raft_volume_free_matches_the_volume_engine():
    """RAFT_bi(precision="f16x3") with the volume-free correlation (default) against the same engine on the fp32 all-pairs volume
    (PP_RAFT_SPLIT_VOLUME=1: the reference's own call pattern, RAFT/corr.py:13-60) at 128x192, 6 iterations: same flows to fp32-class
    end-point error."""
    import os
    from tests.helpers import load_golden, seeded_models
    g = load_golden("raft_128x192.npz")
    fr = (torch.from_numpy(g["frames_u8"]).permute(0, 3, 1, 2).float().div(255)[None] * 2 - 1).cuda()
    flows = {}
    for vol in ("0", "1"):
        os.environ["PP_RAFT_SPLIT_VOLUME"] = vol
        try:
            raft = seeded_models("cuda", raft_precision="f16x3")[0]
            eng = raft._get_engine("f16x3", fr.device)
            assert eng.corr_otf == (vol == "0")
            flows[vol] = raft(fr, iters=int(g["iters"]))
        finally:
            os.environ.pop("PP_RAFT_SPLIT_VOLUME", None)
    torch.cuda.synchronize()
    for d in (0, 1):
        e = (flows["0"][d] - flows["1"][d]).pow(2).sum(2).sqrt()
        print(f"SPLIT_PARITY raft otf vs volume dir {d}: EPE mean {e.mean():.2e} max {e.max():.2e}")
        assert e.mean() < 2e-5 and e.max() < 1e-3
    ef = (flows["0"][0][0].cpu() - torch.from_numpy(g["flows_f"])).pow(2).sum(1).sqrt()
    assert ef.mean() < 5e-5 and ef.max() < 2e-3, (ef.mean(), ef.max())
