"""Parity of every HIP kernel (called through the C-ABI) against plain PyTorch fp32 references / the CPU oracle.
Tolerances: fp32 kernels 1e-4 relative to the reference's max magnitude (fp32 MFMA == fmaf chain); fp16 kernels
(fp16 storage, fp32 accumulation) 1e-2 relative unless stated."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import propainter_oracle as O
from oracle.deform_conv_ref import deform_conv2d
from tests.helpers import report

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.float16]


def tol(dt, scale=1.0):
    return (2e-4 if dt == torch.float32 else 1e-2) * scale


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from propainter_amd import hip
    hip.lib()
    return torch.device("cuda:0")


def nhwc(x, dt, cpad=None):
    """NCHW fp32 CPU -> NHWC device tensor with zero channel padding."""
    n, c, h, w = x.shape
    cp = cpad or (c + 7) // 8 * 8
    out = torch.zeros(n, h, w, cp, dtype=dt, device="cuda")
    out[..., :c] = x.permute(0, 2, 3, 1).to("cuda", dt)
    return out


def check(name, got, ref, rtol):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    lim = rtol * max(ref.abs().max().item(), 1e-3)
    assert math.isfinite(err) and err <= lim, report(name, got, ref) + f" limit {lim:.3e}"


CONV_CASES = [
    dict(name="3x3", cin=[128], cout=128, k=(3, 3), stride=1, pad=1),
    dict(name="7x7s2_c3", cin=[3], cout=64, k=(7, 7), stride=2, pad=3),
    dict(name="1x1_c324", cin=[324], cout=256, k=(1, 1), stride=1, pad=0),
    dict(name="two_src", cin=[192, 64], cout=126, k=(3, 3), stride=1, pad=1, act="relu"),
    dict(name="three_src_small", cin=[128, 128, 5], cout=128, k=(3, 3), stride=1, pad=1, act="lrelu"),
    dict(name="grouped", cin=[32, 48], cout=256, k=(3, 3), stride=1, pad=1, groups=8, act="lrelu"),
    dict(name="dilated", cin=[128], cout=128, k=(3, 3), stride=1, pad=3, dil=3),
    dict(name="replicate5x5s2", cin=[3], cout=32, k=(5, 5), stride=2, pad=2, pad_mode="replicate"),
    dict(name="cout2_f32out", cin=[256], cout=2, k=(3, 3), stride=1, pad=1, out_f32=True),
    dict(name="cout3_tanh", cin=[64], cout=3, k=(3, 3), stride=1, pad=1, act="tanh"),
    dict(name="1x5", cin=[128, 256], cout=256, k=(1, 5), stride=1, pad=(0, 2), act="sigmoid"),
    dict(name="temporal3x1", cin=[64], cout=64, k=(3, 1), stride=1, pad=(2, 0), dil=(2, 1)),
    dict(name="residual_relu2", cin=[96], cout=96, k=(3, 3), stride=1, pad=1, act="relu", residual=True, act2="relu"),
    dict(name="7x7s3_c40", cin=[40], cout=512, k=(7, 7), stride=3, pad=3, residual=True),
    dict(name="cout48", cin=[64], cout=48, k=(3, 3), stride=2, pad=1),
    dict(name="cout432", cin=[128], cout=432, k=(3, 3), stride=1, pad=1),
    dict(name="5x1_tanh", cin=[128, 256], cout=128, k=(5, 1), stride=1, pad=(2, 0), act="tanh"),          # halo-tile kernel, 16x8 tiles
    dict(name="3x3_res_halo", cin=[128], cout=128, k=(3, 3), stride=1, pad=1, act="lrelu", residual=True),   # fp32 staging epilogue
    dict(name="3x3_cout64_halo", cin=[64, 64], cout=64, k=(3, 3), stride=1, pad=1, act="relu"),
]


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "f16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c["name"] for c in CONV_CASES])
def test_conv2d(dev, case, dt):
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(100 + [c["name"] for c in CONV_CASES].index(case["name"]))
    groups = case.get("groups", 1)
    cin = case["cin"]
    kh, kw = case["k"]
    N, H, W = 2, 19, 23
    w = torch.randn(case["cout"], sum(cin), kh, kw, generator=g) / math.sqrt(sum(cin) * kh * kw)
    b = torch.randn(case["cout"], generator=g) * 0.1
    srcs_nchw = [torch.randn(N, groups * c, H, W, generator=g) for c in cin]
    # grouped reference input: per group, concatenate the group's slice of every source
    x = torch.cat([torch.cat([s[:, gi * c:(gi + 1) * c] for s, c in zip(srcs_nchw, cin)], 1) for gi in range(groups)], 1)
    dil = case.get("dil", 1)
    if case.get("pad_mode") == "replicate":
        p = case["pad"]
        ref = F.conv2d(F.pad(x, (p, p, p, p), mode="replicate"), w, b, case["stride"], 0, dil, groups)
    else:
        ref = F.conv2d(x, w, b, case["stride"], case["pad"], dil, groups)
    act = case.get("act")
    if act == "relu": ref = F.relu(ref)
    elif act == "lrelu": ref = F.leaky_relu(ref, 0.1)
    elif act == "sigmoid": ref = torch.sigmoid(ref)
    elif act == "tanh": ref = torch.tanh(ref)
    res = None
    if case.get("residual"):
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res
        if case.get("act2") == "relu":
            ref = F.relu(ref)
    layer = ConvLayer(w, b, stride=case["stride"], padding=case["pad"], dilation=dil, groups=groups, src_channels=cin,
                      pad_mode=case.get("pad_mode", "zeros"), dtype=dt, device=dev)
    srcs = [nhwc(s, dt) for s in srcs_nchw]
    out = layer(srcs, act=act, act_param=0.1, residual=None if res is None else nhwc(res, dt), act2=case.get("act2"),
                out_dtype=torch.float32 if case.get("out_f32") else None)
    torch.cuda.synchronize()
    assert out.dtype == (torch.float32 if case.get("out_f32") else dt)
    check(case["name"], out[..., :case["cout"]].permute(0, 3, 1, 2), ref, tol(dt))
    if out.shape[-1] > case["cout"]:
        assert (out[..., case["cout"]:] == 0).all(), "channel padding must stay zero"


@pytest.mark.parametrize("mode", ["f16_halo", "f16_v2", "f16_v1", "f32", "f32_split3"])
@pytest.mark.parametrize("k,pad", [((1, 5), (0, 2)), ((5, 1), (2, 0))], ids=["1x5", "5x1"])
def test_conv_fused_gru_epilogue(dev, k, pad, mode):
    """SepConvGRU half step (RAFT/update.py:45-60) as two convolutions with the fused epilogue: the iteration-invariant
    part of the input enters as a pre-activation addend, z / r*h and (1-z)*h + z*q come out of the epilogues -- against
    the plain formulation in torch, through every kernel family that can serve the layer."""
    from propainter_amd.conv import ConvLayer
    dt = torch.float16 if mode.startswith("f16") else torch.float32
    g = torch.Generator().manual_seed(55)
    N, H, W, C = 2, 19, 27, 128
    h0, inp, mf = (torch.randn(N, C, H, W, generator=g) * 0.7 for _ in range(3))
    q8 = lambda t: t.to(dt).float()
    h0, inp, mf = q8(h0), q8(inp), q8(mf)
    wz, wr, wq = (torch.randn(C, 3 * C, *k, generator=g) / math.sqrt(3 * C * 5) for _ in range(3))
    bz, br, bq = (torch.randn(C, generator=g) * 0.1 for _ in range(3))
    hx = torch.cat([h0, inp, mf], 1)
    z = torch.sigmoid(F.conv2d(hx, wz, bz, 1, pad))
    r = torch.sigmoid(F.conv2d(hx, wr, br, 1, pad))
    qv = torch.tanh(F.conv2d(torch.cat([r * h0, inp, mf], 1), wq, bq, 1, pad))
    href = (1 - z) * h0 + z * qv
    mk = lambda w, b, sc: ConvLayer(w, b, padding=pad, src_channels=sc, dtype=dt, device=dev, split3=(mode == "f32_split3"))
    wzr, bzr = torch.cat([wz, wr], 0), torch.cat([bz, br], 0)
    it = lambda w: torch.cat([w[:, :C], w[:, 2 * C:]], 1)
    zr_pre, q_pre = mk(wzr[:, C:2 * C], bzr, [C]), mk(wq[:, C:2 * C], bq, [C])
    zr_it, q_it = mk(it(wzr), None, [C, C]), mk(it(wq), None, [C, C])
    impl = {"f16_halo": 70, "f16_v2": 12, "f16_v1": 1}.get(mode, 0)
    zr_it.impl = q_it.impl = impl
    hd, inpd, mfd = nhwc(h0, dt), nhwc(inp, dt), nhwc(mf, dt)
    pzr, pq = zr_pre([inpd]), q_pre([inpd])
    zbuf = torch.full((N, H, W, C + 8), 5.0, dtype=dt, device=dev)
    rh = torch.empty((N, H, W, C), dtype=dt, device=dev)
    net = hd.clone()
    zr_it([net, mfd], out=zbuf, out_choff=8, act="sigmoid", preadd=pzr, fuse=dict(kind="gru_zr", h=net, out2=rh, split=C))
    q_it([rh, mfd], out=net, act="tanh", preadd=pq, fuse=dict(kind="gru_h", h=net, z=(zbuf, 8)))
    torch.cuda.synchronize()
    t = tol(dt, 2.0)
    assert (zbuf[..., :8] == 5).all()
    check("z", zbuf[..., 8:].permute(0, 3, 1, 2), z, t)
    check("r*h", rh.permute(0, 3, 1, 2), r * h0, t)
    check("h_new", net.permute(0, 3, 1, 2), href, t)


@pytest.mark.parametrize("impl", [0, 109, 12], ids=["auto", "halo_epilogue_reads", "v2"])
@pytest.mark.parametrize("cout,hw,k,pad", [(128, (40, 72), (3, 3), 1), (64, (40, 72), (3, 3), 1), (128, (19, 27), (3, 3), 1), (256, (33, 50), (1, 5), (0, 2))],
                         ids=["c128", "c64", "c128_small_grid", "c256_1x5"])
@pytest.mark.parametrize("act2", [None, "relu"])
def test_conv_linear_residual_and_preadd_through_the_matrix_cores(dev, cout, hw, k, pad, act2, impl):
    """A residual added to a LINEAR convolution (out = act2(conv + bias + residual): the backbone convolutions of both propagation
    modules) and a pre-activation addend (the SepConvGRU partial sums) enter the halo kernel as one more K block with identity
    weights (exact: fp16 x 1.0 into the fp32 accumulators) -- against torch, and against the epilogue-read form of the same kernel
    (impl 109 exists in diagnostic builds only; skipped otherwise) and the v2 kernel."""
    from propainter_amd import hip
    from propainter_amd.conv import ConvLayer
    dt = torch.float16
    g = torch.Generator().manual_seed(123)
    N, (H, W), C = 2, hw, 128
    x = (torch.randn(N, C, H, W, generator=g) * 0.8).to(dt).float()
    w = torch.randn(cout, C, *k, generator=g) / math.sqrt(C * k[0] * k[1])
    b = torch.randn(cout, generator=g) * 0.1
    res = (torch.randn(N, cout, H, W, generator=g) * 2).to(dt).float()
    y = F.conv2d(x, w.to(dt).float(), b, 1, pad)
    ref_res = y + res
    if act2 == "relu":
        ref_res = F.relu(ref_res)
    ref_pre = torch.tanh(y + res)
    layer = ConvLayer(w, b, padding=pad, src_channels=[C], dtype=dt, device=dev)
    layer.impl = impl
    xd, rd = nhwc(x, dt), nhwc(res, dt)
    try:
        out = layer([xd], residual=rd, act2=act2)
        out_pre = layer([xd], preadd=rd, act="tanh")
        torch.cuda.synchronize()
    except RuntimeError as e:
        if impl == 109 and "not available" in str(e):
            pytest.skip("impl 109 is compiled in PP_DIAG builds only")
        raise
    check("conv + residual", out[..., :cout].permute(0, 3, 1, 2), ref_res, tol(dt, 2.0))
    check("tanh(conv + preadd)", out_pre[..., :cout].permute(0, 3, 1, 2), ref_pre, tol(dt, 2.0))


@pytest.mark.parametrize("dt", [torch.float16, torch.float32], ids=["f16", "f32"])
def test_raft_flow_taps_7x1_equals_7x7(dev, dt):
    """Motion encoder convf1 = Conv2d(2, 128, 7, padding=3) (RAFT/update.py:85,92) as pp_raft_flow_taps + a 7x1 convolution over
    the 16 gathered channels: same terms as the 7x7 window; the flow side output feeds the GRU input window."""
    from propainter_amd import hip
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(91)
    P, h, w = 3, 13, 22
    wf = torch.randn(128, 2, 7, 7, generator=g) / math.sqrt(98)
    bf = torch.randn(128, generator=g) * 0.1
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    c0 = torch.stack([xs, ys], -1)[None].expand(P, h, w, 2).contiguous()
    c1 = c0 + torch.randn(P, h, w, 2, generator=g) * 6
    flow = (c1 - c0).to(dt).float()
    ref = F.relu(F.conv2d(flow.permute(0, 3, 1, 2), wf.to(dt).float(), bf, 1, 3))
    wrow = torch.zeros(128, 16, 7, 1)
    for kx in range(7):
        wrow[:, 2 * kx:2 * kx + 2, :, 0] = wf[:, :, :, kx]
    layer = ConvLayer(wrow, bf, padding=(3, 0), src_channels=[16], dtype=dt, device=dev)
    rows = torch.full((P, h, w, 16), 7.0, dtype=dt, device=dev)
    xbuf = torch.full((P, h, w, 16), 5.0, dtype=dt, device=dev)
    hip.raft_flow_taps(c1.to(dev), c0.to(dev), rows, flow_out=xbuf, flow_choff=6)
    out = layer([rows], act="relu")
    torch.cuda.synchronize()
    assert torch.equal(xbuf[..., 6:8].float().cpu(), flow) and (xbuf[..., :6] == 5).all() and (xbuf[..., 8:] == 5).all()
    assert (rows[..., 14:] == 0).all() and torch.equal(rows[..., 6:8].float().cpu(), flow)
    check("convf1 as taps + 7x1", out.permute(0, 3, 1, 2), ref, tol(dt, 2.0))


@pytest.mark.parametrize("mode", ["f16_halo", "f16_v2", "f16_v1", "f32"])
@pytest.mark.parametrize("with_flow", [True, False], ids=["flow", "noflow"])
def test_conv_fused_dcn_offset_mask_head(dev, mode, with_flow):
    """conv_offset head of the deformable alignment (model/propainter.py:57-65, recurrent_flow_completion.py:31-40):
    mag * tanh(offsets) (+ flow flipped to (y, x) per tap) | sigmoid(masks) fused into the convolution epilogue, against
    torch; for fp32 also bit-identical with the plain convolution followed by pp_dcn_offset_mask_act."""
    from propainter_amd import hip
    from propainter_amd.conv import ConvLayer
    dt = torch.float16 if mode.startswith("f16") else torch.float32
    g = torch.Generator().manual_seed(77)
    N, H, W, C, mag = 2, 21, 30, 128, 3.0
    x = (torch.randn(N, C, H, W, generator=g) * 0.8).to(dt).float()
    w = torch.randn(432, C, 3, 3, generator=g) / math.sqrt(C * 9) * 2
    b = torch.randn(432, generator=g) * 0.1
    flow = (torch.randn(N, 2, H, W, generator=g) * 4).to(dt).float()
    y = F.conv2d(x, w.to(dt).float(), b, 1, 1)
    off = mag * torch.tanh(y[:, :288])
    if with_flow:
        off = off + flow.flip(1).repeat(1, 144, 1, 1)
    ref = torch.cat([off, torch.sigmoid(y[:, 288:])], 1)
    layer = ConvLayer(w, b, padding=1, src_channels=[C], dtype=dt, device=dev)
    layer.impl = {"f16_halo": 70, "f16_v2": 12, "f16_v1": 1}.get(mode, 0)
    xd = nhwc(x, dt)
    aux = torch.zeros((N, H, W, 8), dtype=dt, device=dev)
    aux[..., :2] = nhwc(flow, dt)[..., :2]
    fl = aux if with_flow else None
    out = layer([xd], fuse=dict(kind="dcn_om", mag=mag, flow=fl))
    torch.cuda.synchronize()
    check("fused offset/mask head", out.permute(0, 3, 1, 2), ref, tol(dt, 3.0))
    if dt == torch.float32:
        plain = layer([xd])
        hip.dcn_offset_mask_act(plain, mag, flow=fl)
        torch.cuda.synchronize()
        assert torch.equal(plain, out), "fused head must equal conv + pp_dcn_offset_mask_act bit for bit in fp32"


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c["name"] in ("3x3", "7x7s2_c3", "1x1_c324", "two_src", "1x5", "cout2_f32out", "residual_relu2")],
                         ids=lambda c: c["name"])
def test_conv2d_split3_is_fp32_class(dev, case):
    """fp32 tensors with every product as three fp16 MFMAs (hi*hi + hi*lo + lo*hi, pp_conv_args_t.impl 3): the error
    against an fp64 reference must be fp32-class (1e-5 of the range; the exact fp32 kernel measures ~2e-6, the fp16
    engine ~1e-3)."""
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(300 + [c["name"] for c in CONV_CASES].index(case["name"]))
    cin, (kh, kw) = case["cin"], case["k"]
    N, H, W = 2, 19, 23
    w = torch.randn(case["cout"], sum(cin), kh, kw, generator=g) / math.sqrt(sum(cin) * kh * kw)
    b = torch.randn(case["cout"], generator=g) * 0.1
    srcs_nchw = [torch.randn(N, c, H, W, generator=g) * 3 for c in cin]
    ref = F.conv2d(torch.cat(srcs_nchw, 1).double(), w.double(), b.double(), case["stride"], case["pad"])
    act = case.get("act")
    if act == "relu": ref = F.relu(ref)
    elif act == "sigmoid": ref = torch.sigmoid(ref)
    res = None
    if case.get("residual"):
        res = torch.randn(ref.shape, generator=g)
        ref = F.relu(ref + res.double()) if case.get("act2") == "relu" else ref + res.double()
    errs = {}
    for split3 in (True, False):
        layer = ConvLayer(w, b, stride=case["stride"], padding=case["pad"], src_channels=cin, dtype=torch.float32, device=dev,
                          split3=split3)
        out = layer([nhwc(s_, torch.float32) for s_ in srcs_nchw], act=act, residual=None if res is None else nhwc(res, torch.float32),
                    act2=case.get("act2"))
        torch.cuda.synchronize()
        errs[split3] = ((out[..., :case["cout"]].permute(0, 3, 1, 2).double().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert errs[True] < 1e-5, f"split3 relative error {errs[True]:.2e} (exact fp32 kernel {errs[False]:.2e})"


@pytest.mark.parametrize("k,pad", [((3, 3), 1), ((1, 5), (0, 2)), ((5, 1), (2, 0))], ids=["3x3", "1x5", "5x1"])
def test_conv_kernel_families_are_bit_identical(dev, k, pad):
    """The register-staged kernel (impl 1), the LDS-DMA kernel (impl 12) and the halo-tile kernel (impl 70) walk K in the
    same (table-defined) order with the same MFMA, so on an fp16 layer they must agree bit for bit -- on a ragged map
    (edge tiles partly outside), with two sources, bias, activation and a channel-window output."""
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(21)
    N, H, W = 2, 21, 37
    cin, cout = [64, 128], 192
    w = torch.randn(cout, sum(cin), *k, generator=g) / math.sqrt(sum(cin) * k[0] * k[1])
    b = torch.randn(cout, generator=g) * 0.1
    layer = ConvLayer(w, b, padding=pad, src_channels=cin, dtype=torch.float16, device=dev)
    srcs = [nhwc(torch.randn(N, c, H, W, generator=g), torch.float16) for c in cin]
    outs = {}
    for impl in (1, 12, 70):
        layer.impl = impl
        buf = torch.full((N, H, W, 256), 3.0, dtype=torch.float16, device=dev)
        layer(srcs, out=buf, out_choff=64, act="lrelu", act_param=0.2)
        torch.cuda.synchronize()
        outs[impl] = buf
        assert (buf[..., :64] == 3).all()
    assert torch.equal(outs[1], outs[12]) and torch.equal(outs[1], outs[70])
    ref = F.leaky_relu(F.conv2d(torch.cat([s[..., :c].permute(0, 3, 1, 2).float().cpu() for s, c in zip(srcs, cin)], 1), w, b, 1, pad), 0.2)
    check("families", outs[70][..., 64:].permute(0, 3, 1, 2), ref, 1e-2)


@pytest.mark.parametrize("residual", [False, True], ids=["plain", "residual"])
def test_a_stationary_gemm_is_bit_identical_to_the_tiled_kernel(dev, residual):
    """conv_gemm_ast.hip (impl 80: activations register-resident, weights streamed over all couts) against the tiled LDS-DMA
    kernel (impl 12) on a Linear 512 -> 1960 with a ragged row count, bias, activation and (optionally) a residual."""
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(33)
    M, K, N = 3001, 512, 1960
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    x = torch.randn(1, 1, M, K, generator=g).to(dev, torch.float16)
    res = torch.randn(1, 1, M, N, generator=g).to(dev, torch.float16) if residual else None
    layer = ConvLayer(w, b, dtype=torch.float16, device=dev)
    outs = {}
    for impl in (12, 80):
        layer.impl = impl
        outs[impl] = layer([x], act="lrelu", act_param=0.2, residual=res).clone()
        torch.cuda.synchronize()
    assert torch.equal(outs[12], outs[80])
    ref = F.leaky_relu(x[0, 0].float().cpu() @ w.t() + b, 0.2) + (res[0, 0].float().cpu() if residual else 0)
    check("ast", outs[80][0, 0], ref, 1e-2)


@pytest.mark.parametrize("k,cin,cout", [(1, 512, 1960), (7, 40, 512)], ids=["linear_512_1960", "7x7s3_c40_512"])
def test_256x256_tile_is_bit_identical_to_the_256x128_tile(dev, k, cin, cout):
    """conv_v2_dispatch picks the 256 x 256 block tile (impl 18: 64 x 128 wave tiles, epilogue in two 64-cout halves) for cout >= 512
    once a launch has 256 blocks; same K order as the 256 x 128 tile (impl 13), so the outputs must be equal -- ragged row count,
    cout not a multiple of 256, bias, activation, residual, output written into a channel window of a wider buffer."""
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(41)
    if k == 1:
        N, H, W, stride, pad = 1, 1, 9001, 1, 0
    else:
        N, H, W, stride, pad = 2, 96, 141, 3, 3
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(N, cin, H, W, generator=g)
    layer = ConvLayer(w, b, stride=stride, padding=pad, dtype=torch.float16, device=dev)
    OH, OW = layer.out_hw(H, W)
    res = torch.randn(N, OH, OW, cout, generator=g).to(dev, torch.float16)
    outs = {}
    for impl in (13, 18):
        layer.impl = impl
        buf = torch.full((N, OH, OW, cout + 72), 3.0, dtype=torch.float16, device=dev)
        layer([nhwc(x, torch.float16)], out=buf, out_choff=8, act="lrelu", act_param=0.2, residual=res)
        torch.cuda.synchronize()
        outs[impl] = buf
    assert torch.equal(outs[13], outs[18])
    assert (outs[18][..., :8] == 3.0).all() and (outs[18][..., 8 + cout:] == 3.0).all()
    ref = F.leaky_relu(F.conv2d(x.half().float(), w.half().float(), b, stride, pad), 0.2) + res.float().cpu().permute(0, 3, 1, 2)
    check("tile256", outs[18][..., 8:8 + cout].permute(0, 3, 1, 2), ref, 1e-2)


@pytest.mark.parametrize("dt", [torch.float16, torch.float32], ids=["f16", "f32"])
def test_pack_nhwc8_equals_three_window_writes(dev, dt):
    """pp_pack_nhwc8 (the encoder input cat(frame, mask, updated mask), model/propainter.py:334-336, as one launch of whole 16-byte rows)
    against three pp_nchw_to_nhwc channel-window writes into a zeroed buffer: identical bytes, odd sizes, one and two sources as well."""
    from propainter_amd import hip
    g = torch.Generator().manual_seed(9)
    N, H, W = 3, 37, 53
    a, b, c = (torch.randn(N, k, H, W, generator=g).to(dev, dt) for k in (3, 1, 1))
    ref = torch.zeros(N, H, W, 8, dtype=dt, device=dev)
    hip.nchw_to_nhwc(a, out=ref, out_choff=0)
    hip.nchw_to_nhwc(b, out=ref, out_choff=3)
    hip.nchw_to_nhwc(c, out=ref, out_choff=4)
    out = torch.full((N, H, W, 8), 7.0, dtype=dt, device=dev)
    assert hip.pack_nhwc8([a, b, c], out=out) is out
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    one = hip.pack_nhwc8([a])
    two = hip.pack_nhwc8([b, a])
    torch.cuda.synchronize()
    assert torch.equal(one[..., :3], ref[..., :3]) and (one[..., 3:] == 0).all()
    assert torch.equal(two[..., 0], ref[..., 3]) and torch.equal(two[..., 1:4], ref[..., :3]) and (two[..., 4:] == 0).all()
    with pytest.raises(AssertionError):
        hip.pack_nhwc8([torch.zeros(N, 5, H, W, dtype=dt, device=dev), torch.zeros(N, 4, H, W, dtype=dt, device=dev)])


def test_pack_nhwc8_split_plane_equals_the_split_window_write(dev):
    """pp_pack_nhwc8 with dtype PP_F16S (fp32 frames -> [8 hi | 8 lo] fp16 rows, the RAFT encoders' input of the f16x3 engine) against
    pp_nchw_to_nhwc's split-plane window write into a zeroed buffer: identical bytes, odd sizes; hi + lo restores 22 bits of the input."""
    from propainter_amd import hip
    g = torch.Generator().manual_seed(10)
    N, H, W = 2, 41, 67
    a = (torch.randn(N, 3, H, W, generator=g) * 3).to(dev)
    b = torch.rand(N, 1, H, W, generator=g).to(dev)
    ref = hip.nchw_to_nhwc(a, cpad=8, split=True)
    hip.nchw_to_nhwc(b, out=ref, out_choff=3, split=True)
    out = torch.full((N, H, W, 16), 7.0, dtype=torch.float16, device=dev)
    assert hip.pack_nhwc8([a, b], out=out, split=True) is out
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    back = (out[..., :4].float() + out[..., 8:12].float()).permute(0, 3, 1, 2)
    assert (back - torch.cat([a, b], 1)).abs().max().item() <= 3 * 4 * 2.0 ** -22
    with pytest.raises(AssertionError):
        hip.pack_nhwc8([a.half()], split=True)


def test_conv2d_output_window_and_large_m(dev):
    """writes into a channel window of a wider buffer; M not a multiple of the tile; asymmetric data (transposes)."""
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 64, 37, 41, generator=g)
    w = torch.randn(126, 64, 3, 3, generator=g) / 24
    layer = ConvLayer(w, None, padding=1, dtype=torch.float16, device=dev)
    buf = torch.full((1, 37, 41, 256), 7.0, dtype=torch.float16, device=dev)
    layer([nhwc(x, torch.float16)], out=buf, out_choff=128)
    torch.cuda.synchronize()
    check("window", buf[..., 128:254].permute(0, 3, 1, 2), F.conv2d(x, w, None, 1, 1), 1e-2)
    assert (buf[..., :128] == 7).all() and (buf[..., 254:] == 7).all()


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "f16"])
def test_batched_gemm_nt(dev, dt):
    from propainter_amd.conv import batched_gemm_nt
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(3, 150, 256, generator=g), torch.randn(3, 150, 256, generator=g)
    out = batched_gemm_nt(a.to(dev, dt), b.to(dev, dt), out_scale=1 / 16)
    torch.cuda.synchronize()
    ref = torch.matmul(a.to(dt).float(), b.to(dt).float().transpose(1, 2)) / 16
    check("gemm", out, ref, 2e-4 if dt == torch.float32 else 2e-3)


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "f16"])
@pytest.mark.parametrize("cin", [[128], [128, 128]], ids=["gen", "fc"])
def test_deform_conv(dev, dt, cin):
    """offset/mask head activation vs torch, then the deformable implicit GEMM vs the oracle restatement fed with
    the device-produced (dtype-rounded) offsets and masks."""
    from propainter_amd import hip
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(5)
    N, H, W = 1, 17, 21
    ctot = sum(cin)
    x = torch.randn(N, ctot, H, W, generator=g)
    w = torch.randn(128, ctot, 3, 3, generator=g) / math.sqrt(ctot * 9)
    b = torch.randn(128, generator=g) * 0.1
    raw = torch.randn(N, 432, H, W, generator=g)
    gen = len(cin) == 1
    mag = 3.0 if gen else 5.0
    flow = None
    if gen:
        flow = torch.randn(N, 2, H, W, generator=g) * 2
        flow[:, :, 0, 0] = 40.0                    # whole neighbourhood far outside -> zero columns
    else:
        raw[:, :288, 0, 1] = 100.0                 # saturated tanh: offsets exactly +5
    rq = raw.to(dt).float()
    offset = mag * torch.tanh(rq[:, :288])
    if gen:
        offset = offset + flow.to(dt).float().flip(1).repeat(1, 144, 1, 1)
    mask = torch.sigmoid(rq[:, 288:])
    om = nhwc(raw, dt)
    hip.dcn_offset_mask_act(om, mag, flow=nhwc(flow, dt) if gen else None)
    torch.cuda.synchronize()
    check("offmask_act_off", om[..., :288].permute(0, 3, 1, 2), offset, tol(dt, 2))
    check("offmask_act_mask", om[..., 288:432].permute(0, 3, 1, 2), mask, tol(dt, 2))
    omc = om.float().cpu().permute(0, 3, 1, 2)
    ref = deform_conv2d(x.to(dt).float(), omc[:, :288].contiguous(), w.to(dt).float(), b, 1, 1, 1, omc[:, 288:432].contiguous())
    layer = ConvLayer(w, b, padding=1, src_channels=cin, dcn_groups=16, dtype=dt, device=dev)
    srcs, off = [], 0
    for c in cin:
        srcs.append(nhwc(x[:, off:off + c], dt))
        off += c
    out = layer(srcs, dcn_offmask=om)
    torch.cuda.synchronize()
    check("deform", out.permute(0, 3, 1, 2), ref, tol(dt, 2))


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "f16"])
@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
@pytest.mark.parametrize("c", [128, 1, 2])
def test_flow_warp(dev, dt, mode, c):
    from propainter_amd import hip
    g = torch.Generator().manual_seed(9)
    N, H, W = 2, 24, 33
    x = torch.randn(N, c, H, W, generator=g)
    flow = torch.randn(N, H, W, 2, generator=g) * 3
    flow[0, 0, :5] = torch.tensor([0.5, 0.0]); flow[0, 1, :5] = torch.tensor([1.5, 2.5])      # ties
    flow[0, 2, :5] = torch.tensor([-40.0, 3.0])                                              # out of bounds
    xq, fq = x.to(dt).float(), flow.to(dt).float()
    ref = O.flow_warp(xq, fq, mode)
    out = hip.flow_warp(nhwc(x, dt) if c % 8 == 0 else x.permute(0, 2, 3, 1).contiguous().to(dev, dt),
                        flow.to(dev, dt).contiguous(), mode=mode)
    torch.cuda.synchronize()
    check(f"warp_{mode}_{c}", out.permute(0, 3, 1, 2), ref, tol(dt))


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "f16"])
def test_fb_check(dev, dt):
    from propainter_amd import hip
    g = torch.Generator().manual_seed(10)
    N, H, W = 2, 30, 40
    fw = F.interpolate(torch.randn(N, 2, 4, 5, generator=g) * 3, size=(H, W), mode="bilinear", align_corners=True)
    bw = -fw + torch.randn(N, 2, H, W, generator=g) * 0.45        # smooth flow, noisy inverse: ~half the pixels valid
    fwq, bwq = fw.to(dt).float(), bw.to(dt).float()
    ref = O.fb_consistency_check(fwq, bwq)
    aux = torch.zeros(N, H, W, 8, dtype=dt, device=dev)
    aux[..., :2] = fw.permute(0, 2, 3, 1).to(dev, dt)
    hip.fb_check(aux, bw.permute(0, 2, 3, 1).contiguous().to(dev, dt), out=aux, out_choff=2)
    torch.cuda.synchronize()
    got = aux[..., 2].float().cpu()
    mism = (got != ref[:, 0]).float().mean().item()
    assert mism <= (0.0 if dt == torch.float32 else 5e-3) + 2e-3, f"fb_check mismatch fraction {mism}"
    assert 0.2 < ref.mean() < 0.98


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "f16"])
def test_corr_pyramid_and_lookup(dev, dt):
    from propainter_amd import hip
    from propainter_amd.conv import batched_gemm_nt
    g = torch.Generator().manual_seed(12)
    P, h, w = 2, 16, 24
    f1, f2 = torch.randn(P, 256, h, w, generator=g), torch.randn(P, 256, h, w, generator=g)
    f1q, f2q = f1.to(dt).float(), f2.to(dt).float()
    pyr = O.corr_pyramid(f1q, f2q)
    n8 = h * w
    vol = batched_gemm_nt(f1.permute(0, 2, 3, 1).reshape(P, n8, 256).to(dev, dt).contiguous(),
                          f2.permute(0, 2, 3, 1).reshape(P, n8, 256).to(dev, dt).contiguous(), out_scale=1 / 16)
    levels = [vol.view(P * n8, h, w)]
    hh, ww = h, w
    for _ in range(3):
        levels.append(hip.corr_avgpool(levels[-1], P * n8, hh, ww))
        hh, ww = hh // 2, ww // 2
    torch.cuda.synchronize()
    for l in range(4):
        check(f"pyr{l}", levels[l], pyr[l][:, 0], 3e-4 if dt == torch.float32 else 3e-3)
    coords = O.coords_grid(P, h, w) + torch.randn(P, 2, h, w, generator=g) * 4
    coords[0, :, 0, 0] = torch.tensor([-3.3, 2.2]); coords[0, :, 0, 1] = torch.tensor([30.0, 20.5])
    ref = O.corr_lookup(pyr, coords)
    out = torch.empty(P, h, w, 328, dtype=dt, device=dev)
    hip.corr_lookup(levels, coords.permute(0, 2, 3, 1).contiguous().to(dev), out)
    torch.cuda.synchronize()
    check("lookup", out[..., :324].permute(0, 3, 1, 2), ref, 3e-4 if dt == torch.float32 else 5e-3)
    assert (out[..., 324:] == 0).all()


@pytest.mark.parametrize("case", ["smooth", "ragged_oob", "divergent", "chaotic"])
def test_corr_lookup_on_the_fly_matches_the_volume_pyramid(dev, case):
    """pp_corr_feature_pyramid + pp_corr_lookup_otf (no all-pairs volume) against the oracle's volume -> avg-pool
    pyramid -> bilinear lookup on the same fp16-rounded features.  Cases: a smooth flow (one shared box per 8x8 tile),
    a map with ragged tiles + coordinates far outside the map / on exact integers, a flow that diverges inside tiles
    (quadrant fallback) and per-pixel random targets (single-pixel fallback)."""
    from propainter_amd import hip
    g = torch.Generator().manual_seed({"smooth": 1, "ragged_oob": 2, "divergent": 3, "chaotic": 4}[case])
    P, h, w = (2, 24, 40) if case != "ragged_oob" else (2, 19, 29)
    f1, f2 = torch.randn(P, 256, h, w, generator=g), torch.randn(P, 256, h, w, generator=g)
    f1q, f2q = f1.half().float(), f2.half().float()
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
    ref = O.corr_lookup(O.corr_pyramid(f1q, f2q), coords)
    f1d, f2d = nhwc(f1, torch.float16), nhwc(f2, torch.float16)
    lv = hip.corr_feature_pyramid(f2d)
    assert [tuple(t.shape) for t in lv] == [(P, h >> l, w >> l, 256) for l in range(4)]
    pooled = F.avg_pool2d(f2q[:, :, :(h >> 2) * 4, :(w >> 2) * 4], 4, 4)
    check("f2_level2", lv[2].permute(0, 3, 1, 2), pooled, 2e-3)
    out = torch.full((P, h, w, 328), 7.0, dtype=torch.float16, device=dev)
    hip.corr_lookup_otf(f1d, lv, coords.permute(0, 2, 3, 1).contiguous().to(dev), out)
    again = torch.empty_like(out)
    hip.corr_lookup_otf(f1d, lv, coords.permute(0, 2, 3, 1).contiguous().to(dev), again)
    torch.cuda.synchronize()
    check("lookup_otf", out[..., :324].permute(0, 3, 1, 2), ref, 5e-3)
    assert (out[..., 324:] == 0).all()
    assert torch.equal(out, again), "the on-the-fly lookup is not run-to-run deterministic"
    # batch invariance: pair 1 alone gives the same bytes as inside the batch of 2
    solo = torch.empty((1, h, w, 328), dtype=torch.float16, device=dev)
    hip.corr_lookup_otf(f1d[1:].contiguous(), [t[1:].contiguous() for t in lv], coords[1:].permute(0, 2, 3, 1).contiguous().to(dev), solo)
    torch.cuda.synchronize()
    assert torch.equal(solo[0], out[1])


@pytest.mark.parametrize("case", ["gen_flow_shift", "gen_wild_offsets", "fc_two_src", "gen_channel_window"])
def test_deform_conv_patch_staged_kernel(dev, case):
    """The patch-staged fp16 deformable kernel (conv_dcn.hip, impl 90) against the oracle restatement and bit-compared
    with the register-staged gather (impl 1) on the same offsets: a common flow displacement of (+7.6, -9.3) px that the
    tile-mean patch shift must absorb, offsets of +-20 px per sample (every corner from the global fallback, many outside
    the image), the two-source 256-channel flow-completion layer, and channel windows / ragged tiles (33 x 45 pixels)."""
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed({"gen_flow_shift": 41, "gen_wild_offsets": 42, "fc_two_src": 43, "gen_channel_window": 44}[case])
    dt = torch.float16
    cin = [128, 128] if case == "fc_two_src" else [128]
    N, H, W = 2, 33, 45
    ctot = sum(cin)
    x = torch.randn(N, ctot, H, W, generator=g)
    w = torch.randn(128, ctot, 3, 3, generator=g) / math.sqrt(ctot * 9)
    b = torch.randn(128, generator=g) * 0.1
    off = torch.randn(N, 288, H, W, generator=g) * 1.5
    if case == "gen_flow_shift":
        off[:, 0::2] += 7.6
        off[:, 1::2] -= 9.3
    elif case == "gen_wild_offsets":
        off = (torch.rand(N, 288, H, W, generator=g) * 2 - 1) * 20
    elif case == "fc_two_src":
        off = 5 * torch.tanh(off)
        off[:, :, 3, 4] = 5.0
    msk = torch.rand(N, 144, H, W, generator=g)
    om = nhwc(torch.cat([off, msk], 1), dt)
    omc = om.float().cpu().permute(0, 3, 1, 2)
    ref = deform_conv2d(x.to(dt).float(), omc[:, :288].contiguous(), w.to(dt).float(), b, 1, 1, 1, omc[:, 288:432].contiguous())
    layer = ConvLayer(w, b, padding=1, src_channels=cin, dcn_groups=16, dtype=dt, device=dev)
    if case == "gen_channel_window":         # source and output as channel windows of wider buffers
        wide = torch.zeros(N, H, W, 160, dtype=dt, device=dev)
        wide[..., 16:144] = x.permute(0, 2, 3, 1).to(dev, dt)
        srcs, kw = [(wide, 16)], dict(out=torch.full((N, H, W, 144), 2.0, dtype=dt, device=dev), out_choff=8)
    else:
        srcs, o_, kw = [], 0, {}
        for c in cin:
            srcs.append(nhwc(x[:, o_:o_ + c], dt))
            o_ += c
    outs = {}
    for impl in (90, 106, 122, 1):     # 90: tile rows by launch size; 106 / 122: 128- / 64-pixel tiles forced; 1: register-staged gather
        layer.impl = impl
        if "out" in kw:
            kw["out"] = torch.full((N, H, W, 144), 2.0, dtype=dt, device=dev)
        o = layer(srcs, dcn_offmask=om, **kw)
        torch.cuda.synchronize()
        outs[impl] = o
    # the two tile heights sample the same corners with the same arithmetic (only the staged patch differs): identical bytes
    assert torch.equal(outs[106], outs[122]) and torch.equal(outs[90], outs[122])
    if "out" in kw:
        assert (outs[90][..., :8] == 2).all() and (outs[90][..., 136:] == 2).all()
        got90, got1 = outs[90][..., 8:136], outs[1][..., 8:136]
    else:
        got90, got1 = outs[90], outs[1]
    check("deform_patch", got90.permute(0, 3, 1, 2), ref, tol(dt, 2))
    d = (got90.float() - got1.float()).abs().max().item()
    assert d <= 2e-2 * ref.abs().max().item(), f"patch-staged vs register-staged kernel: max |d| {d}"
    again = layer(srcs, dcn_offmask=om, **({} if "out" not in kw else dict(out=torch.full((N, H, W, 144), 2.0, dtype=dt, device=dev), out_choff=8)))
    torch.cuda.synchronize()
    layer.impl = 90
    a2 = layer(srcs, dcn_offmask=om, **({} if "out" not in kw else dict(out=torch.full((N, H, W, 144), 2.0, dtype=dt, device=dev), out_choff=8)))
    torch.cuda.synchronize()
    assert torch.equal(a2, outs[90]), "patch-staged deformable kernel is not run-to-run deterministic"


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "f16"])
def test_convex_upsample(dev, dt):
    from propainter_amd import hip
    g = torch.Generator().manual_seed(13)
    B, h, w = 2, 9, 11
    flow, mask = torch.randn(B, 2, h, w, generator=g) * 3, torch.randn(B, 576, h, w, generator=g)
    ref = O.convex_upsample(flow, mask.to(dt).float())
    out = hip.convex_upsample(flow.permute(0, 2, 3, 1).contiguous().to(dev), nhwc(mask, dt))
    torch.cuda.synchronize()
    check("convex", out, ref, 2e-4)


def _attention_reference(q, k, v, pk, pv, own, rolled, tind, wmask, heads=4):
    """torch fp32 restatement on explicit index sets (same math as oracle.sparse_window_attention)."""
    B, T, Hp, Wp, C = q.shape
    ch = C // heads
    qf, kf, vf = (a.reshape(B, T, Hp * Wp, heads, ch) for a in (q, k, v))
    out = torch.zeros_like(qf)
    for b in range(B):
        for w in range(own.shape[0]):
            qi = qf[b][:, own[w]]
            if wmask[b, w] > 0:
                idx = torch.cat([own[w], rolled[w]])
                kk = torch.cat([kf[b][tind][:, idx], pk[b][tind].reshape(len(tind), -1, heads, ch)], 1).reshape(-1, heads, ch)
                vv = torch.cat([vf[b][tind][:, idx], pv[b][tind].reshape(len(tind), -1, heads, ch)], 1).reshape(-1, heads, ch)
                a = torch.softmax(torch.einsum("qhc,khc->hqk", qi.reshape(-1, heads, ch), kk) / math.sqrt(ch), -1)
                y = torch.einsum("hqk,khc->qhc", a, vv).reshape(T, -1, heads, ch)
            else:
                a = torch.softmax(torch.einsum("tqhc,tkhc->thqk", qi, kf[b][:, own[w]]) / math.sqrt(ch), -1)
                y = torch.einsum("thqk,tkhc->tqhc", a, vf[b][:, own[w]])
            out[b][:, own[w]] = y
    return out.reshape(B, T, Hp, Wp, C)


@pytest.mark.parametrize("pattern", ["none", "all", "batch2"])
def test_sparse_window_attention_mask_patterns(dev, pattern):
    """Persistent masked-window launch: empty list (no masked window), full list, and two batch entries with different
    masked sets (the compacted list spans the batch)."""
    from propainter_amd import hip
    dt = torch.float16
    g = torch.Generator().manual_seed(15)
    B = 2 if pattern == "batch2" else 1
    T, Hp, Wp, C = 4, 10, 18, 512
    q, k, v = (torch.randn(B, T, Hp, Wp, C, generator=g) for _ in range(3))
    P = (Hp // 4) * (Wp // 4)
    pk, pv = torch.randn(B, T, P, C, generator=g), torch.randn(B, T, P, C, generator=g)
    own_np, rolled_np = hip.window_tables(Hp, Wp)
    own, rolled = torch.from_numpy(own_np).long(), torch.from_numpy(rolled_np).long()
    wmask = {"none": torch.zeros(1, 4), "all": torch.ones(1, 4), "batch2": torch.tensor([[1.0, 0, 0, 1], [0, 0, 3.0, 0]])}[pattern]
    tind = torch.tensor([0, 2])
    cast = lambda a: a.to(dt).float()
    ref = _attention_reference(cast(q), cast(k), cast(v), cast(pk), cast(pv), own, rolled, tind, wmask)
    out = hip.sparse_window_attention(q.to(dev, dt), k.to(dev, dt), v.to(dev, dt), pk.to(dev, dt), pv.to(dev, dt),
                                      torch.from_numpy(own_np).to(dev), torch.from_numpy(rolled_np).to(dev),
                                      tind.to(dev, torch.int32), wmask.to(dev))
    torch.cuda.synchronize()
    check(pattern, out, ref, 6e-3)


def _attention_by_rolling_tensors(q, k, v, pk, pv, tind, wmask, heads=4, ws=(5, 9)):
    """SparseWindowAttention.forward on given q / k / v / pooled k, v WITHOUT any index table: window_partition (sparse_transformer.py:104-115),
    the four torch.roll shifts and the valid_ind_rolled selection (:140-155,182-205), masked windows attending to own + rolled + pooled keys
    of the T_ind frames, unmasked ones to their own frame's window (:227-269) -- the reference's own tensor operations, restated."""
    B, T, Hp, Wp, C = q.shape
    wh, ww = ws
    ch = C // heads
    nwh, nww = Hp // wh, Wp // ww

    def part(x):          # (B, T, H, W, C) -> (B, n_windows, heads, T, wh * ww, ch)
        x = x.view(B, T, nwh, wh, nww, ww, heads, ch).permute(0, 2, 4, 6, 1, 3, 5, 7).contiguous()
        return x.view(B, nwh * nww, heads, T, wh * ww, ch)
    eh, ew = (wh + 1) // 2, (ww + 1) // 2
    m_tl, m_tr, m_bl, m_br = (torch.ones(wh, ww) for _ in range(4))
    m_tl[:-eh, :-ew] = 0
    m_tr[:-eh, ew:] = 0
    m_bl[eh:, :-ew] = 0
    m_br[eh:, ew:] = 0
    valid = torch.stack((m_tl, m_tr, m_bl, m_br), 0).flatten(0).nonzero(as_tuple=False).view(-1)
    rolled = lambda a: torch.cat([part(torch.roll(a, shifts=sh, dims=(2, 3))) for sh in ((-eh, -ew), (-eh, ew), (eh, -ew), (eh, ew))], 4)[:, :, :, :, valid]
    wq, wk, wv, rk, rv = part(q), part(k), part(v), rolled(k), rolled(v)
    P = pk.shape[2]
    pool = lambda a: a.view(B, T, P, heads, ch).permute(0, 3, 1, 2, 4)[:, None].expand(B, nwh * nww, heads, T, P, ch)
    pkw, pvw = pool(pk), pool(pv)
    out = torch.zeros_like(wq)
    for b in range(B):
        for w in range(nwh * nww):
            if wmask[b, w] > 0:
                kk = torch.cat([wk[b, w][:, tind], rk[b, w][:, tind], pkw[b, w][:, tind]], 2).reshape(heads, -1, ch)
                vv = torch.cat([wv[b, w][:, tind], rv[b, w][:, tind], pvw[b, w][:, tind]], 2).reshape(heads, -1, ch)
                a = torch.softmax(wq[b, w].reshape(heads, -1, ch) @ kk.transpose(-2, -1) / math.sqrt(ch), -1)
                out[b, w] = (a @ vv).view(heads, T, wh * ww, ch)
            else:
                a = torch.softmax(wq[b, w] @ wk[b, w].transpose(-2, -1) / math.sqrt(ch), -1)
                out[b, w] = a @ wv[b, w]
    out = out.view(B, nwh, nww, heads, T, wh, ww, ch).permute(0, 4, 1, 5, 2, 6, 3, 7).contiguous()
    return out.view(B, T, Hp, Wp, C)


@pytest.mark.parametrize("grid", [(10, 18), (15, 27)], ids=["2x2 windows", "3x3 windows"])
def test_sparse_window_attention_against_rolled_tensors(dev, grid):
    """The op with the PRODUCT's index tables (hip.window_tables) against a reference that never sees an index set: it rolls and
    partitions the key / value TENSORS like sparse_transformer.py:174-205 (the other op tests take their
    index sets from the product, so a wrong roll table would pass them).  Circular wrap repeats tokens on small grids: multiplicity
    matters, and the 3 x 3 grid has an interior window whose rolled neighbourhood does not wrap."""
    from propainter_amd import hip
    Hp, Wp = grid
    g = torch.Generator().manual_seed(150 + Hp)
    B, T, C = 1, 4, 512
    q, k, v = (torch.randn(B, T, Hp, Wp, C, generator=g) for _ in range(3))
    P = (Hp // 4) * (Wp // 4)
    pk, pv = torch.randn(B, T, P, C, generator=g), torch.randn(B, T, P, C, generator=g)
    nw = (Hp // 5) * (Wp // 9)
    wmask = (torch.arange(nw) % 3 != 1).float()[None]              # masked and unmasked windows
    tind = torch.tensor([1, 3])
    for dt, lim in ((torch.float32, 2e-4), (torch.float16, 6e-3)):
        cast = lambda a: a.to(dt).float()
        ref = _attention_by_rolling_tensors(cast(q), cast(k), cast(v), cast(pk), cast(pv), tind, wmask)
        own_np, rolled_np = hip.window_tables(Hp, Wp)
        out = hip.sparse_window_attention(q.to(dev, dt), k.to(dev, dt), v.to(dev, dt), pk.to(dev, dt), pv.to(dev, dt),
                                          torch.from_numpy(own_np).to(dev), torch.from_numpy(rolled_np).to(dev),
                                          tind.to(dev, torch.int32), wmask.to(dev), impl=0 if dt == torch.float16 else 1)
        torch.cuda.synchronize()
        check(f"attention_vs_rolled_tensors_{Hp}x{Wp}_{dt}", out, ref, lim)


def test_sparse_window_attention_with_more_than_64_key_frames(dev):
    """A long clip's window: 140 frames, temporal dilation 2 -> 70 key frames in a masked window's key set (the reference has no limit:
    sparse_transformer.py:337-342 builds T_ind for any t; rounds 1-5 stopped at 64, the kernel's key-frame table is 256 entries now).
    Against the tensor-rolling reference, both kernels."""
    from propainter_amd import hip
    Hp, Wp = 10, 18
    g = torch.Generator().manual_seed(164)
    B, T, C = 1, 140, 512
    q, k, v = (torch.randn(B, T, Hp, Wp, C, generator=g) for _ in range(3))
    P = (Hp // 4) * (Wp // 4)
    pk, pv = torch.randn(B, T, P, C, generator=g), torch.randn(B, T, P, C, generator=g)
    wmask = torch.tensor([[1.0, 0.0, 0.0, 3.0]])
    tind = torch.arange(1, T, 2)
    assert tind.numel() == 70
    for dt, lim in ((torch.float32, 2e-4), (torch.float16, 6e-3)):
        cast = lambda a: a.to(dt).float()
        ref = _attention_by_rolling_tensors(cast(q), cast(k), cast(v), cast(pk), cast(pv), tind, wmask)
        own_np, rolled_np = hip.window_tables(Hp, Wp)
        out = hip.sparse_window_attention(q.to(dev, dt), k.to(dev, dt), v.to(dev, dt), pk.to(dev, dt), pv.to(dev, dt),
                                          torch.from_numpy(own_np).to(dev), torch.from_numpy(rolled_np).to(dev),
                                          tind.to(dev, torch.int32), wmask.to(dev), impl=0 if dt == torch.float16 else 1)
        torch.cuda.synchronize()
        check(f"attention_70_key_frames_{dt}", out, ref, lim)


@pytest.mark.parametrize("variant", ["ref_f32", "ref_f16", "mfma_f16"])
def test_sparse_window_attention(dev, variant):
    from propainter_amd import hip
    dt = torch.float32 if variant == "ref_f32" else torch.float16
    g = torch.Generator().manual_seed(14)
    B, T, Hp, Wp, C = 1, 5, 10, 18, 512
    q, k, v = (torch.randn(B, T, Hp, Wp, C, generator=g) for _ in range(3))
    q = q * 2.0
    P = (Hp // 4) * (Wp // 4)
    pk, pv = torch.randn(B, T, P, C, generator=g), torch.randn(B, T, P, C, generator=g)
    own_np, rolled_np = hip.window_tables(Hp, Wp)
    own, rolled = torch.from_numpy(own_np).long(), torch.from_numpy(rolled_np).long()
    wmask = torch.tensor([[0.0, 2.0, 1.0, 0.0]])
    tind = torch.tensor([1, 3])
    cast = lambda a: a.to(dt).float()
    ref = _attention_reference(cast(q), cast(k), cast(v), cast(pk), cast(pv), own, rolled, tind, wmask)
    # fused buffers exercise the cstride path
    qkv = torch.cat([q, k, v], -1).to(dev, dt).contiguous()
    pkv = torch.cat([pk, pv], -1).to(dev, dt).contiguous()
    out = hip.sparse_window_attention(qkv, qkv[..., C:], qkv[..., 2 * C:], pkv, pkv[..., C:], torch.from_numpy(own_np).to(dev),
                                      torch.from_numpy(rolled_np).to(dev), tind.to(dev, torch.int32), wmask.to(dev),
                                      qkv_cstride=3 * C, pkv_cstride=2 * C, C_=C, impl=0 if variant == "mfma_f16" else 1)
    torch.cuda.synchronize()
    check(variant, out, ref, 2e-4 if dt == torch.float32 else 6e-3)


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "f16"])
def test_token_ops(dev, dt):
    from propainter_amd import hip
    g = torch.Generator().manual_seed(15)
    BT, H, W, C = 2, 16, 24, 40
    fh, fw = O.token_grid(H), O.token_grid(W)
    tok = torch.randn(BT, fh * fw, C * 49, generator=g)
    tq = tok.to(dt).float()
    folded = F.fold(tq.permute(0, 2, 1), (H, W), 7, 1, 3, 3)
    norm = F.fold(torch.ones_like(tq).permute(0, 2, 1), (H, W), 7, 1, 3, 3)
    # the device op takes tap-major features ((ky*7+kx)*C + c); the reference (F.fold) order is c*49 + ky*7 + kx
    tok_dev = tok.view(BT, fh * fw, C, 49).transpose(2, 3).contiguous().view(BT, fh * fw, 49 * C).to(dev, dt)
    out = hip.fold_tokens(tok_dev, BT, fh, fw, C, H, W, normalize=True, act=hip.ACT_GELU)
    out2 = hip.fold_tokens(tok_dev, BT, fh, fw, C, H, W, normalize=False)
    torch.cuda.synchronize()
    check("fold_norm_gelu", out.permute(0, 3, 1, 2), F.gelu(folded / norm), tol(dt))
    check("fold", out2.permute(0, 3, 1, 2), folded, tol(dt))
    x = torch.randn(3, 7, 512, generator=g) * 2 + 0.5
    gam, bet = torch.rand(512, generator=g) + 0.5, torch.randn(512, generator=g)
    ln = hip.layernorm(x.to(dev, dt), gam.to(dev), bet.to(dev))
    torch.cuda.synchronize()
    check("layernorm", ln, F.layer_norm(x.to(dt).float(), (512,), gam, bet), tol(dt))
    xp = torch.randn(2, 512, 8, 12, generator=g)
    pw, pb = torch.randn(512, 1, 4, 4, generator=g) / 16, torch.randn(512, generator=g) * 0.1
    dp = hip.depthwise_pool(nhwc(xp, dt), pw.view(512, 4, 4).to(dev).contiguous(), pb.to(dev), 4)
    torch.cuda.synchronize()
    check("pool", dp.permute(0, 3, 1, 2), F.conv2d(xp.to(dt).float(), pw, pb, 4, 0, 1, 512), tol(dt))


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "f16"])
def test_elementwise_ops(dev, dt):
    from propainter_amd import hip
    g = torch.Generator().manual_seed(16)
    x = torch.randn(2, 96, 13, 17, generator=g) * 2 + 0.3
    xq = x.to(dt).float()
    out = hip.instance_norm(nhwc(x, dt), relu=True)
    torch.cuda.synchronize()
    check("instance_norm", out.permute(0, 3, 1, 2), F.relu(F.instance_norm(xq)), tol(dt, 2))
    up = hip.upsample2x(nhwc(x, dt))
    torch.cuda.synchronize()
    check("upsample2x", up.permute(0, 3, 1, 2), F.interpolate(xq, scale_factor=2, mode="bilinear", align_corners=True), tol(dt))
    zr = torch.rand(2, 256, 5, 6, generator=g)
    h, q = torch.randn(2, 128, 5, 6, generator=g), torch.randn(2, 128, 5, 6, generator=g)
    rh = torch.empty(2, 5, 6, 128, dtype=dt, device=dev)
    hip.gru_gate(nhwc(zr, dt), nhwc(h, dt), 0, 128, rh, 0)
    hn = nhwc(h, dt)
    hip.gru_gate(nhwc(zr, dt), hn, 0, 128, hn, 0, q=nhwc(q, dt))
    torch.cuda.synchronize()
    zq, hq, qq = zr.to(dt).float(), h.to(dt).float(), q.to(dt).float()
    check("gru_rh", rh.permute(0, 3, 1, 2), zq[:, 128:] * hq, tol(dt))
    check("gru_h", hn.permute(0, 3, 1, 2), (1 - zq[:, :128]) * hq + zq[:, :128] * qq, tol(dt))
    y = torch.randn(2, 3, 9, 10, generator=g)
    buf = torch.zeros(2, 9, 10, 8, dtype=dt, device=dev)
    hip.nchw_to_nhwc(y.to(dev), out=buf, out_choff=2)
    back = hip.nhwc_to_nchw(buf, 3, choff=2, out_dtype=torch.float32)
    torch.cuda.synchronize()
    check("layout_roundtrip", back, y, tol(dt))
    assert (buf[..., :2] == 0).all() and (buf[..., 5:] == 0).all()
    mask = (torch.rand(1, 3, 10, 18, generator=g) > 0.97).float()
    wm = hip.window_mask(mask.to(dev, dt))
    torch.cuda.synchronize()
    ref = F.max_pool2d(mask, (5, 9), (5, 9)).view(1, 3, -1).sum(1)
    assert torch.equal(wm.cpu(), ref)
    big = (torch.rand(2, 3, 20, 36, generator=g) > 0.99).float()          # windows of 10 x 12 = 120 positions (> one wave: the generic path)
    wm2 = hip.window_mask(big.to(dev, dt), 10, 12)
    torch.cuda.synchronize()
    assert torch.equal(wm2.cpu(), F.max_pool2d(big, (10, 12), (10, 12)).view(2, 3, -1).sum(1))


@pytest.mark.parametrize("k", [0, 1, 4, 7])
def test_binary_dilate_is_bit_identical_to_scipy(dev, k):
    """pp_binary_dilate == scipy.ndimage.binary_dilation(mask, iterations=k) (cross element, zero border) -- the mask
    pre-processing of the driver (inference_propainter.py:96,105); blobs touching every border, isolated pixels, empty frame."""
    import scipy.ndimage
    from propainter_amd import hip
    rng = np.random.RandomState(3 + k)
    m = (rng.rand(3, 37, 53) > 0.985).astype(np.uint8) * 200
    m[0, :3, :5] = 255; m[0, -1, -1] = 1; m[1, 17:22, 0] = 9; m[2] = 0
    out = hip.binary_dilate(torch.from_numpy(m).to(dev), k).cpu().numpy()
    for i in range(3):
        want = (scipy.ndimage.binary_dilation(m[i], iterations=k) if k > 0 else m[i] > 0).astype(np.uint8) * 255
        assert np.array_equal(out[i], want), (k, i, int((out[i] != want).sum()))


@pytest.mark.parametrize("size", [(216, 120), (300, 200), (864, 480), (431, 239), (432, 240)], ids=lambda s_: f"{s_[0]}x{s_[1]}")
def test_resize_bilinear_u8_is_the_cv2_arithmetic(dev, size):
    """pp_resize_bilinear_u8 (the final cv2.resize(f, out_size) of the driver, inference_propainter.py:469-470, on the device) against the
    numpy restatement of OpenCV's fixed-point INTER_LINEAR (video_io.resize_u8_linear): byte for byte -- exact 2:1 (the INTER_AREA
    shortcut), fractional down- and up-scaling, identity -- and against float bilinear interpolation (align_corners=False, no
    antialiasing) within one level."""
    from propainter_amd import hip, video_io
    rng = np.random.RandomState(5)
    a = rng.randint(0, 256, (3, 240, 432, 3)).astype(np.uint8)
    out = hip.resize_bilinear_u8(torch.from_numpy(a).to(dev), size)
    torch.cuda.synchronize()
    ref = np.stack([video_io.resize_u8_linear(f, size) for f in a])
    assert out.shape == (3, size[1], size[0], 3) and np.array_equal(out.cpu().numpy(), ref)
    fl = F.interpolate(torch.from_numpy(a).permute(0, 3, 1, 2).float(), size=(size[1], size[0]), mode="bilinear", align_corners=False)
    assert (out.cpu().float() - fl.permute(0, 2, 3, 1)).abs().max().item() <= 1.0
