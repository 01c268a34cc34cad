"""CPU restatement of ``torchvision.ops.deform_conv2d`` (modulated, v2).

Test infrastructure only.  torchvision is an un-vendored third-party dependency
of the reference (``requirements.txt:11`` pins only ``torchvision>=0.8.2``) and is
absent from this image, so its published algorithm is restated here; call sites:
``model/propainter.py:67-69``, ``model/recurrent_flow_completion.py:42-44``.

Semantics (torchvision/csrc/ops/cpu/deform_conv2d_kernel.cpp, ``deformable_im2col``
+ ``bilinear_interpolate``):
  * ``offset`` is ``[N, 2*G*K, Ho, Wo]`` with channel ``2*(g*K+k)`` = dy and
    ``2*(g*K+k)+1`` = dx for offset-group ``g`` and tap ``k = ky*kw + kx``;
  * ``mask`` is ``[N, G*K, Ho, Wo]``;
  * input channel ``c`` belongs to offset-group ``c // (Cin // G)``;
  * sample position ``(oy*s - p + ky*d + dy, ox*s - p + kx*d + dx)``, bilinear with
    every out-of-image corner contributing 0 (whole sample 0 if y<=-1, y>=H, x<=-1, x>=W);
  * columns ``[Cin*K, Ho*Wo]`` (times mask) x weight ``[Cout, Cin*K]`` + bias.
"parity unpinned" against torchvision itself; pinned by known answers in
``tests/test_oracle_cpu.py`` (zero offset == conv2d, integer offset == shifted conv).
"""
import torch
import torch.nn.functional as F


def bilinear_zeros(img, py, px):
    """img [N,C,H,W]; py,px [N,...] pixel coordinates -> [N,C,...]; zero outside."""
    N, C, H, W = img.shape
    shp = py.shape[1:]
    py = py.reshape(N, -1)
    px = px.reshape(N, -1)
    y0 = torch.floor(py)
    x0 = torch.floor(px)
    ly = py - y0
    lx = px - x0
    y0 = y0.long()
    x0 = x0.long()
    flat = img.reshape(N, C, H * W)
    out = torch.zeros(N, C, py.shape[1], dtype=img.dtype)
    for dy, wy in ((0, 1 - ly), (1, ly)):
        for dx, wx in ((0, 1 - lx), (1, lx)):
            yy = y0 + dy
            xx = x0 + dx
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1))
            v = torch.gather(flat, 2, idx[:, None, :].expand(N, C, -1))
            out = out + v * (wy * wx * ok.to(img.dtype))[:, None, :]
    return out.reshape(N, C, *shp)


def deform_conv2d(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0),
                  dilation=(1, 1), mask=None):
    def _pair(v):
        return (v, v) if isinstance(v, int) else tuple(v)
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    N, Cin, H, W = input.shape
    Cout, Cin_g, kh, kw = weight.shape
    assert Cin_g == Cin, "weight groups != 1 not used by the reference"
    K = kh * kw
    G = offset.shape[1] // (2 * K)
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    dt = input.dtype
    oy = torch.arange(Ho, dtype=dt).view(1, 1, 1, Ho, 1) * sh - ph
    ox = torch.arange(Wo, dtype=dt).view(1, 1, 1, 1, Wo) * sw - pw
    ky = (torch.arange(K) // kw).to(dt).view(1, 1, K, 1, 1) * dh
    kx = (torch.arange(K) % kw).to(dt).view(1, 1, K, 1, 1) * dw
    off = offset.view(N, G, K, 2, Ho, Wo)
    py = oy + ky + off[:, :, :, 0]          # [N,G,K,Ho,Wo]
    px = ox + kx + off[:, :, :, 1]
    cg = Cin // G
    xg = input.view(N * G, cg, H, W)
    samp = bilinear_zeros(xg, py.reshape(N * G, K, Ho, Wo), px.reshape(N * G, K, Ho, Wo))
    samp = samp.view(N, G, cg, K, Ho, Wo)
    if mask is not None:
        samp = samp * mask.view(N, G, 1, K, Ho, Wo)
    cols = samp.reshape(N, Cin * K, Ho * Wo)       # channel-major, tap-minor == weight.view(Cout, Cin*K)
    out = torch.matmul(weight.view(1, Cout, Cin * K), cols).view(N, Cout, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
