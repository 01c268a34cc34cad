"""The ping-pong halo kernel (csrc/conv_halo8.h, impl 82 / 83: 256-pixel tiles, eight waves in two groups running in opposite phases,
weights two steps ahead in a ring of three LDS stages, counted vmcnt waits) against the 128-pixel halo kernel (impl 71 / 72) it is
derived from: same K order, same MFMA sequence per accumulator, same epilogue -> BIT-IDENTICAL outputs.  Ragged maps (edge tiles
partly outside the image, a tile row that is half empty), two sources, every epilogue form, fp16 and split-plane; every comparison is
repeated (a racing LDS-DMA schedule is wrong only now and then)."""
import math

import pytest
import torch

from tests.cpu_emulation import split_planes

pytestmark = pytest.mark.gpu
REPEATS = 4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from propainter_amd import hip
    hip.lib()
    return torch.device("cuda:0")


def nhwc(x, dt):
    return x.permute(0, 2, 3, 1).contiguous().to("cuda", dt)


def _inputs(g, N, cin, H, W, split):
    from propainter_amd.conv import pad8
    if split:
        return [split_planes((torch.randn(N, c, H, W, generator=g) * (1.0 + i)).permute(0, 2, 3, 1).contiguous(), pad8(c)).cuda() for i, c in enumerate(cin)]
    return [nhwc(torch.randn(N, c, H, W, generator=g), torch.float16) for c in cin]


def _like_out(g, N, C, H, W, split, scale=1.0):
    t = torch.randn(N, C, H, W, generator=g) * scale
    return split_planes(t.permute(0, 2, 3, 1).contiguous()).cuda() if split else nhwc(t, torch.float16)


@pytest.mark.parametrize("split", [False, True], ids=["f16", "f16x3"])
@pytest.mark.parametrize("k,pad", [((3, 3), 1), ((1, 5), (0, 2)), ((5, 1), (2, 0))], ids=["3x3", "1x5", "5x1"])
@pytest.mark.parametrize("cout,bn", [(256, 128), (128, 128), (192, 64), (64, 64)], ids=["c256", "c128", "c192_bn64", "c64_bn64"])
@pytest.mark.parametrize("epi", ["act", "preadd", "linear_residual", "residual_after_act"])
def test_ping_pong_kernel_is_bit_identical_to_the_128_pixel_kernel(dev, split, k, pad, cout, bn, epi):
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(77)
    N, H, W = 3, 41, 53                      # 3 x 4 tiles of 16 x 16 per image: bottom row 9 of 16 rows, right column 5 of 16 columns
    cin = [128, 64]
    w = torch.randn(cout, sum(cin), *k, generator=g) / math.sqrt(sum(cin) * k[0] * k[1])
    b = torch.randn(cout, generator=g) * 0.1
    layer = ConvLayer(w, b, padding=pad, src_channels=cin, dtype=torch.float16, device=dev, split=split)
    srcs = _inputs(g, N, cin, H, W, split)
    kw = {}
    if epi == "act":
        kw = dict(act="lrelu", act_param=0.2)
    elif epi == "preadd":
        kw = dict(act="tanh", preadd=_like_out(g, N, cout, H, W, split, 2.0))
    elif epi == "linear_residual":
        kw = dict(residual=_like_out(g, N, cout, H, W, split, 2.0), act2="relu")
    else:
        kw = dict(act="relu", residual=_like_out(g, N, cout, H, W, split, 2.0), act2="relu")
    layer.impl = 71 if bn == 128 else 72
    ref = layer(srcs, **kw).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(ref.float()).all() and float(ref.float().abs().max()) > 0.1
    layer.impl = 82 if bn == 128 else 83
    for i in range(REPEATS):
        got = layer(srcs, **kw)
        torch.cuda.synchronize()
        ne = got != ref
        assert not bool(ne.any()), (f"run {i}: {int(ne.sum())} of {ne.numel()} values differ, first at "
                                    f"{tuple(int(v) for v in ne.nonzero()[0])}, max |d| {float((got.float() - ref.float()).abs().max()):.3e}")


@pytest.mark.parametrize("split", [False, True], ids=["f16", "f16x3"])
@pytest.mark.parametrize("k,pad", [((1, 5), (0, 2)), ((5, 1), (2, 0))], ids=["1x5", "5x1"])
def test_ping_pong_kernel_fused_gru_epilogues(dev, split, k, pad):
    """SepConvGRU half step (RAFT/update.py:45-60) with the fused z | r gate (256 couts, r*h as second output) and candidate gate
    epilogues, the partial sums as pre-activation addends: ping-pong kernel vs the 128-pixel kernel, bit for bit, h carried over 3 steps."""
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(55)
    N, H, W, C = 5, 45, 80, 128
    wzr = torch.randn(2 * C, 2 * C, *k, generator=g) / math.sqrt(3 * C * 5)
    wq = torch.randn(C, 2 * C, *k, generator=g) / math.sqrt(3 * C * 5)
    mk = lambda w: ConvLayer(w, None, padding=pad, src_channels=[C, C], dtype=torch.float16, device=dev, split=split)
    zr_it, q_it = mk(wzr), mk(wq)
    h0, mf = _like_out(g, N, C, H, W, split, 0.7), _like_out(g, N, C, H, W, split, 0.7)
    pzr, pq = _like_out(g, N, 2 * C, H, W, split, 0.5), _like_out(g, N, C, H, W, split, 0.5)
    CW = 2 * C if split else C

    def run(impl):
        zr_it.impl = q_it.impl = impl
        net = h0.clone()
        zbuf = torch.empty((N, H, W, CW), dtype=torch.float16, device=dev)
        rh = torch.empty((N, H, W, CW), dtype=torch.float16, device=dev)
        for _ in range(3):
            zr_it([net, mf], out=zbuf, act="sigmoid", preadd=pzr, fuse=dict(kind="gru_zr", h=net, out2=rh, split=C))
            q_it([rh, mf], out=net, act="tanh", preadd=pq, fuse=dict(kind="gru_h", h=net, z=zbuf))
        torch.cuda.synchronize()
        return net, zbuf.clone(), rh.clone()

    ref = run(71)
    assert torch.isfinite(ref[0].float()).all()
    for i in range(REPEATS):
        got = run(82)
        for name, a, b_ in zip(("h", "z", "r*h"), got, ref):
            assert torch.equal(a, b_), f"run {i}: {name} differs in {int((a != b_).sum())} values"


def test_automatic_dispatch_agrees_with_both_kernels(dev):
    """impl 0 takes the 128-pixel kernel (the ping-pong form measured 3-9 % slower: profiles/r6_halo8_pingpong.txt; PP_HALO8=1, read once
    per process, would let launches of >= 256 blocks take it) -- whichever it takes, it must agree with both explicit forms bit for bit,
    on a launch that cannot fill the chip with 256-pixel tiles and on one that can."""
    from propainter_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(5)
    w = torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(128 * 9)
    layer = ConvLayer(w, None, padding=1, src_channels=[128], dtype=torch.float16, device=dev)
    for N, H, W in ((1, 40, 72), (20, 90, 160)):
        x = nhwc(torch.randn(N, 128, H, W, generator=g), torch.float16)
        outs = {}
        for impl in (0, 71, 82):
            layer.impl = impl
            outs[impl] = layer([x], act="relu").clone()
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[71]) and torch.equal(outs[0], outs[82])
