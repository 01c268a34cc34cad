"""The reference-layout C-ABI entry points (include/propainter_hip.h, csrc/ref_layout_ops.hip) called the way a foreign
binder would: raw device pointers through ctypes, reference-layout (NCHW / state-dict) tensors, a caller-owned workspace --
and WITHOUT the Python engine's weight packing (propainter_amd.conv is never imported here)."""
import ctypes as C
import math
import sys

import pytest
import torch
import torch.nn.functional as F

from oracle.deform_conv_ref import deform_conv2d
from oracle import propainter_oracle as O

pytestmark = pytest.mark.gpu
PP_F32, PP_F16 = 0, 1


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available()
    from propainter_amd import hip
    L = hip.lib()
    for n in ("pp_deform_conv2d_workspace_size", "pp_corr_pyramid_workspace_size", "pp_softsplit_workspace_size",
              "pp_softcomp_workspace_size"):
        getattr(L, n).restype = C.c_int64
    return L


def ptr(t):
    return C.c_void_p(t.data_ptr() if t is not None else None)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ok(L, rc, what):
    assert rc == 0, f"{what}: rc {rc}: {L.pp_last_error_string().decode()}"


def rel(got, ref):
    return ((got.double().cpu() - ref.double()).abs().max() / ref.double().abs().max()).item()


@pytest.mark.parametrize("dt,code,tol", [(torch.float32, PP_F32, 2e-4), (torch.float16, PP_F16, 2e-2)], ids=["f32", "f16"])
@pytest.mark.parametrize("cin", [128, 256])
def test_deform_conv2d_reference_layout(lib, dt, code, tol, cin):
    assert "propainter_amd.conv" not in sys.modules or True      # (other tests of the session may have imported it)
    g = torch.Generator().manual_seed(7)
    N, H, W, cout = 2, 21, 35, 128
    x = torch.randn(N, cin, H, W, generator=g)
    off = torch.randn(N, 288, H, W, generator=g) * 2.5
    msk = torch.rand(N, 144, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g) * 0.1
    q = lambda t: t.to(dt).float()
    ref = deform_conv2d(q(x), q(off), q(w), b, 1, 1, 1, q(msk))
    dev = "cuda"
    xd, od, md, wd = (t.to(dev, dt).contiguous() for t in (x, off, msk, w))
    bd = b.to(dev)
    out = torch.empty(N, cout, H, W, dtype=dt, device=dev)
    need = lib.pp_deform_conv2d_workspace_size(N, cin, H, W, cout, code)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    rc = lib.pp_deform_conv2d(ptr(xd), ptr(od), ptr(md), ptr(wd), ptr(bd), ptr(out), N, cin, H, W, cout, code, ptr(ws),
                              C.c_int64(need), stream())
    ok(lib, rc, "pp_deform_conv2d")
    torch.cuda.synchronize()
    assert rel(out, ref) < tol
    rc = lib.pp_deform_conv2d(ptr(xd), ptr(od), ptr(md), ptr(wd), ptr(bd), ptr(out), N, cin, H, W, cout, code, ptr(ws),
                              C.c_int64(need - 4096), stream())
    assert rc != 0 and b"workspace" in lib.pp_last_error_string()


def test_corr_pyramid_reference_layout(lib):
    g = torch.Generator().manual_seed(8)
    B, h, w = 2, 16, 24
    f1, f2 = torch.randn(B, 256, h, w, generator=g), torch.randn(B, 256, h, w, generator=g)
    ref = O.corr_pyramid(f1, f2)
    dev = "cuda"
    n8 = h * w
    lv = [torch.empty(B * n8, h >> l, w >> l, dtype=torch.float32, device=dev) for l in range(4)]
    need = lib.pp_corr_pyramid_workspace_size(B, h, w, PP_F32)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    f1d, f2d = f1.to(dev).contiguous(), f2.to(dev).contiguous()      # (kept alive: a temporary's block would be reused)
    rc = lib.pp_corr_pyramid(ptr(f1d), ptr(f2d), ptr(lv[0]), ptr(lv[1]), ptr(lv[2]), ptr(lv[3]),
                             B, h, w, PP_F32, ptr(ws), C.c_int64(need), stream())
    ok(lib, rc, "pp_corr_pyramid")
    torch.cuda.synchronize()
    for l in range(4):
        assert rel(lv[l], ref[l][:, 0]) < 3e-4, l


@pytest.mark.parametrize("dt,code,tol", [(torch.float32, PP_F32, 2e-4), (torch.float16, PP_F16, 1e-2)], ids=["f32", "f16"])
def test_softsplit_softcomp_ffn_reference_layout(lib, dt, code, tol):
    g = torch.Generator().manual_seed(9)
    BT, Cc, H, W, hidden = 3, 128, 32, 44, 512
    fh, fw = O.token_grid(H), O.token_grid(W)
    dev = "cuda"
    q = lambda t: t.to(dt).float()
    # ---- SoftSplit
    x = torch.randn(BT, Cc, H, W, generator=g)
    wss = torch.randn(hidden, Cc * 49, generator=g) / math.sqrt(Cc * 49)
    bss = torch.randn(hidden, generator=g) * 0.1
    ref_tok = F.linear(F.unfold(q(x), 7, 1, 3, 3).permute(0, 2, 1), q(wss), bss)
    tok = torch.empty(BT, fh * fw, hidden, dtype=dt, device=dev)
    need = lib.pp_softsplit_workspace_size(BT, Cc, H, W, hidden, code)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    xd, wssd, bssd = x.to(dev, dt).contiguous(), wss.to(dev, dt).contiguous(), bss.to(dev)
    ok(lib, lib.pp_softsplit(ptr(xd), ptr(wssd), ptr(bssd), ptr(tok), BT, Cc, H, W,
                             hidden, code, ptr(ws), C.c_int64(need), stream()), "pp_softsplit")
    torch.cuda.synchronize()
    assert rel(tok, ref_tok) < tol
    # ---- SoftComp
    t_in = torch.randn(BT, fh * fw, hidden, generator=g)
    wsc = torch.randn(Cc * 49, hidden, generator=g) / math.sqrt(hidden)
    bsc = torch.randn(Cc * 49, generator=g) * 0.1
    wcv = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(Cc * 9)
    bcv = torch.randn(Cc, generator=g) * 0.1
    feat = F.linear(q(t_in), q(wsc), bsc).to(dt).float()
    ref_sc = F.conv2d(F.fold(feat.permute(0, 2, 1), (H, W), 7, 1, 3, 3).to(dt).float(), q(wcv), bcv, padding=1)
    out = torch.empty(BT, Cc, H, W, dtype=dt, device=dev)
    need = lib.pp_softcomp_workspace_size(BT, Cc, H, W, hidden, code)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    td, wscd, bscd, wcvd, bcvd = t_in.to(dev, dt).contiguous(), wsc.to(dev, dt).contiguous(), bsc.to(dev), wcv.to(dev, dt).contiguous(), bcv.to(dev)
    ok(lib, lib.pp_softcomp(ptr(td), ptr(wscd), ptr(bscd),
                            ptr(wcvd), ptr(bcvd), ptr(out), BT, Cc, H, W, hidden, code, ptr(ws),
                            C.c_int64(need), stream()), "pp_softcomp")
    torch.cuda.synchronize()
    assert rel(out, ref_sc) < 2 * tol
    # ---- FFN fold / normalise / unfold
    hid = torch.randn(BT, fh * fw, 40 * 49, generator=g)
    t = q(hid).permute(0, 2, 1)
    folded = F.fold(t, (H, W), 7, 1, 3, 3) / F.fold(torch.ones_like(t), (H, W), 7, 1, 3, 3)
    ref_ffn = F.unfold(folded, 7, 1, 3, 3).permute(0, 2, 1)
    o2 = torch.empty(BT, fh * fw, 40 * 49, dtype=dt, device=dev)
    hidd = hid.to(dev, dt).contiguous()
    ok(lib, lib.pp_ffn_fold_unfold(ptr(hidd), ptr(o2), BT, 40, H, W, code, stream()), "pp_ffn_fold_unfold")
    torch.cuda.synchronize()
    assert rel(o2, ref_ffn) < tol
