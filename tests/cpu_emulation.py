"""TEST-ONLY emulation of the libpropainter_hip device ops with plain PyTorch on the CPU.

Purpose: validate the *host-side engine graphs* (channel windows, multi-source convolutions, packed weights and
K-chunk tables, fused q/k/v GEMMs, fold rewrites, scheduling) against the reference goldens without a GPU.
Every emulated op consumes exactly the arguments the HIP kernel would get (same packed weights, same tables), so
only the kernels themselves remain to be proven on the GPU.  Never imported by the product.
"""
import contextlib
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import propainter_oracle as O
from oracle.deform_conv_ref import bilinear_zeros
from propainter_amd import conv as pconv
from propainter_amd import hip

_ACT = {hip.ACT_NONE: lambda v, p: v, hip.ACT_RELU: lambda v, p: F.relu(v), hip.ACT_LRELU: lambda v, p: F.leaky_relu(v, p),
        hip.ACT_SIGMOID: lambda v, p: torch.sigmoid(v), hip.ACT_TANH: lambda v, p: torch.tanh(v),
        hip.ACT_GELU: lambda v, p: F.gelu(v)}


def split_planes(x, cpad=None):
    """fp32 [..., C] -> split-plane fp16 [..., 2 * cpad]: hi = fp16(v) | lo = fp16(v - hi) (what the SPLIT epilogues / packers write)."""
    C_ = x.shape[-1]
    cpad = cpad or C_
    hi = x.float().half()
    lo = (x.float() - hi.float()).half()
    out = torch.zeros(x.shape[:-1] + (2 * cpad,), dtype=torch.float16, device=x.device)
    out[..., :C_] = hi
    out[..., cpad:cpad + C_] = lo
    return out


def merge_planes(t, choff=0, C_=None):
    """split-plane fp16 [..., 2 * Cp] -> fp32 [..., C_] = hi + lo of the channel window [choff, choff + C_)."""
    lo = t.shape[-1] // 2
    C_ = lo - choff if C_ is None else C_
    return t[..., choff:choff + C_].float() + t[..., lo + choff:lo + choff + C_].float()


def _put(t, choff, y, split):
    """writes fp32 y into the channel window of t starting at choff (both planes when split)"""
    C_ = y.shape[-1]
    tv = t.view(y.shape[:-1] + (t.shape[-1],))
    if split and t.dtype == torch.float16:
        lo = t.shape[-1] // 2
        hi = y.half()
        tv[..., choff:choff + C_] = hi
        tv[..., lo + choff:lo + choff + C_] = (y - hi.float()).half()
    else:
        tv[..., choff:choff + C_] = y.to(t.dtype)


def _get(t, choff, C_, split):
    return merge_planes(t, choff, C_) if split else t.float()[..., choff:choff + C_]


def _conv_call(self, srcs, out=None, out_choff=0, act=None, act_param=0.0, out_scale=1.0, residual=None, res_choff=0,
               act2=None, out_dtype=None, dcn_offmask=None, out_hw=None, preadd=None, fuse=None):
    srcs = [(s, 0) if torch.is_tensor(s) else s for s in srcs]
    x0 = srcs[0][0]
    N, H, W = x0.shape[:3]
    OH, OW = out_hw if out_hw is not None else self.out_hw(H, W)
    sp = bool(getattr(self, "split", False))
    if out is not None:
        pconv.check_inplace(self, srcs, out, out_choff, fuse, dcn_offmask)      # the intra-launch hazard check sees every emulated launch
    if out is None:
        cp = pconv.pad8(self.cout)
        odt = out_dtype or self.dtype
        out = torch.zeros((N, OH, OW, 2 * cp if (sp and odt == torch.float16) else cp), dtype=odt)
    kt = self.ktable.cpu().numpy()[:self.kchunks]      # the last row is the kernels' 16-byte zero page
    oy = torch.arange(OH).view(1, OH, 1) * self.stride[0] - self.padding[0]
    ox = torch.arange(OW).view(1, 1, OW) * self.stride[1] - self.padding[1]
    nidx = torch.arange(N).view(N, 1, 1)
    for g in range(self.groups):
        A = torch.zeros(N, OH, OW, self.K)
        for kc, (dy, dx, code, choff) in enumerate(kt):
            s = int(code) & 0xff
            if s == 255:
                continue
            src, so = srcs[s]
            cb = so + (g * self.src_channels[s] if self.groups > 1 else 0) + int(choff)
            srcf = src.float()
            if dcn_offmask is None:
                iy, ix = oy + int(dy), ox + int(dx)
                if self.pad_mode == 1:
                    ok = torch.ones(1, OH, OW, dtype=torch.bool)
                    iy, ix = iy.clamp(0, H - 1), ix.clamp(0, W - 1)
                else:
                    ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
                v = srcf[nidx, iy.clamp(0, H - 1).expand(N, OH, OW), ix.clamp(0, W - 1).expand(N, OH, OW), cb:cb + 8]
                A[..., kc * 8:kc * 8 + 8] = v * ok[..., None]
            else:
                grp, tap = (int(code) >> 8) & 0xff, (int(code) >> 16) & 0xff
                om = dcn_offmask.float()
                py = (oy + int(dy)).float() + om[..., 2 * (grp * 9 + tap)]
                px = (ox + int(dx)).float() + om[..., 2 * (grp * 9 + tap) + 1]
                smp = bilinear_zeros(srcf[..., cb:cb + 8].permute(0, 3, 1, 2).contiguous(), py, px)   # [N,8,OH,OW]
                A[..., kc * 8:kc * 8 + 8] = smp.permute(0, 2, 3, 1) * om[..., 288 + grp * 9 + tap][..., None]
        if self.dtype == torch.float16:
            A = A.half().float()
        Wg = self.weight[g, :self.cout_g].float()
        if getattr(self, "tri", False):
            # tri-product format: per (block, tap) the K range holds [32 ch hi | 32 ch lo] on the A side and [W_hi | W_lo] on the weight
            # side; the kernel computes W_hi x A_hi + W_hi x A_lo + W_lo x A_hi (NOT the plain K walk hi x W_hi + lo x W_lo)
            lo = np.repeat((kt[:, 2].astype(np.int64) & pconv.KT_PLANE_LO) != 0, 8)
            hi_c, lo_c = torch.from_numpy(np.nonzero(~lo)[0]), torch.from_numpy(np.nonzero(lo)[0])
            y = A[..., hi_c] @ Wg[:, hi_c].t() + A[..., lo_c] @ Wg[:, hi_c].t() + A[..., hi_c] @ Wg[:, lo_c].t()
        else:
            y = A @ Wg.t()
        if self.bias is not None:
            y = y + self.bias[g * self.cout_g:(g + 1) * self.cout_g]
        y = y * out_scale
        win = lambda x: (x, 0) if torch.is_tensor(x) else x
        if preadd is not None:
            pt, pc = win(preadd)
            y = y + _get(pt, pc, self.cout_g, sp).reshape(y.shape)
        y = _ACT[pconv.ACTS[act]](y, act_param)
        if fuse is not None and fuse["kind"] == "dcn_om":
            Cs = int(fuse.get("split", 288))
            off = float(fuse["mag"]) * torch.tanh(y[..., :Cs])
            if fuse.get("flow") is not None:
                ft, fc = win(fuse["flow"])
                fl = ft.float()[..., fc:fc + 2].reshape(y.shape[:-1] + (2,))
                odd = (torch.arange(Cs) % 2).bool()                 # even channels (dy) += flow_y, odd channels (dx) += flow_x
                off = off + torch.where(odd, fl[..., 0:1], fl[..., 1:2])
            y = torch.cat([off, torch.sigmoid(y[..., Cs:])], -1)
        elif fuse is not None:
            ht, hc = win(fuse["h"])
            if fuse["kind"] == "gru_zr":
                Cs = int(fuse["split"])
                ot, oc = win(fuse["out2"])
                hv = _get(ht, hc, self.cout_g - Cs, sp)
                _put(ot, oc, y[..., Cs:] * hv.reshape(y[..., Cs:].shape), sp)
                _put(out, out_choff, y[..., :Cs], sp)
                continue
            zt, zc = win(fuse["z"])
            hv = _get(ht, hc, self.cout_g, sp).reshape(y.shape)
            zv = _get(zt, zc, self.cout_g, sp).reshape(y.shape)
            y = (1 - zv) * hv + zv * y
        c0 = out_choff + g * self.cout_g
        if residual is not None:
            y = y + _get(residual, res_choff + g * self.cout_g, self.cout_g, sp).reshape(y.shape)
        if pconv.ACTS[act2] == hip.ACT_RELU:
            y = F.relu(y)
        _put(out, c0, y, sp)
    return out


def _batched_gemm_nt_split(a, b, out_scale=1.0, impl=0):
    K = a.shape[-1] // 2
    ah, al, bh, bl = a[..., :K].float(), a[..., K:].float(), b[..., :K].float(), b[..., K:].float()
    return (torch.matmul(ah, bh.transpose(1, 2)) + torch.matmul(al, bh.transpose(1, 2)) + torch.matmul(ah, bl.transpose(1, 2))) * out_scale


def _batched_gemm_nt(a, bt, out_scale=1.0, split3=False):
    return torch.matmul(a.float(), bt.float().transpose(1, 2)) * out_scale


def _flow_warp(x, flow, out=None, mode="bilinear", x_choff=0, C_=None, fl_choff=0, out_choff=0):
    C_ = x.shape[-1] if C_ is None else C_
    xs = x[..., x_choff:x_choff + C_].permute(0, 3, 1, 2).float()
    y = O.flow_warp(xs, flow[..., fl_choff:fl_choff + 2].float(), mode).permute(0, 2, 3, 1)
    if out is None:
        return y.to(x.dtype).contiguous()
    out[..., out_choff:out_choff + C_] = y.to(out.dtype)
    return out


def _fb_check(fw, bw, out=None, out_choff=0):
    v = O.fb_consistency_check(fw[..., :2].permute(0, 3, 1, 2).float(), bw[..., :2].permute(0, 3, 1, 2).float())[:, 0]
    if out is None:
        return v[..., None].to(fw.dtype)
    out[..., out_choff] = v.to(out.dtype)
    return out


def _img_prop_step(x_prop, m_prop, x_cur, m_cur, f_prop, f_chk, x_out, m_out, mode="nearest"):
    fp = f_prop.float()
    valid = O.fb_consistency_check(fp, f_chk.float())
    warped = O.flow_warp(x_prop.float(), fp.permute(0, 2, 3, 1), mode)
    mpv = O.binarize(O.flow_warp(m_prop.float(), fp.permute(0, 2, 3, 1)))
    u = O.binarize(m_cur.float() * valid * (1 - mpv))
    x_out.copy_((u * warped + (1 - u) * x_cur.float()).to(x_out.dtype))
    m_out.copy_(O.binarize(m_cur.float() * (1 - valid * (1 - mpv))).to(m_out.dtype))


def _corr_avgpool(x, M, H, W):
    return F.avg_pool2d(x.view(M, 1, H, W), 2, 2)[:, 0].contiguous()


def _corr_lookup(levels, coords, out, split=False):
    ref = O.corr_lookup([l[:, None] for l in levels], coords.permute(0, 3, 1, 2))
    out.zero_()
    _put(out, 0, ref.permute(0, 2, 3, 1).contiguous(), split)
    return out


def _corr_feature_pyramid(f2):
    lv, x = [f2], f2.float().permute(0, 3, 1, 2)
    P, _, h, w = x.shape
    for l in (1, 2, 3):
        s_ = 1 << l
        y = F.avg_pool2d(x[:, :, :(h >> l) * s_, :(w >> l) * s_], s_, s_)
        lv.append(y.permute(0, 2, 3, 1).contiguous().to(f2.dtype))
    return lv


def _corr_feature_pyramid_split(f2):
    """split-plane features [P,h,w,512]: fp32 means of hi + lo over the 2^l x 2^l blocks, stored as split planes again"""
    x = merge_planes(f2).permute(0, 3, 1, 2)
    P, _, h, w = x.shape
    lv = []
    for l in (1, 2, 3):
        s_ = 1 << l
        y = F.avg_pool2d(x[:, :, :(h >> l) * s_, :(w >> l) * s_], s_, s_)
        lv.append(split_planes(y.permute(0, 2, 3, 1).contiguous()))
    return lv


def _corr_lookup_otf(f1, f2_levels, coords, out):
    P, h, w, _ = f1.shape
    pyr = []
    a = f1.float().reshape(P, h * w, 256)
    for f2 in f2_levels:
        hl, wl = f2.shape[1], f2.shape[2]
        vol = torch.matmul(a, f2.float().reshape(P, hl * wl, 256).transpose(1, 2)) / 16.0
        pyr.append(vol.reshape(P * h * w, 1, hl, wl))
    ref = O.corr_lookup(pyr, coords.permute(0, 3, 1, 2))
    out[..., :324] = ref.permute(0, 2, 3, 1).to(out.dtype)
    out[..., 324:] = 0
    return out


def _corr_lookup_otf_split(f1, f2_levels, coords, out):
    """split-plane features -> split-plane lookup tile with 88 channels per level and plane (81 taps + 7 zeros)"""
    P, h, w, _ = f1.shape
    LV = hip.OTF_SPLIT_LEVEL_CHANNELS
    a = merge_planes(f1).reshape(P, h * w, 256)
    pyr = []
    for f2 in f2_levels:
        hl, wl = f2.shape[1], f2.shape[2]
        vol = torch.matmul(a, merge_planes(f2).reshape(P, hl * wl, 256).transpose(1, 2)) / 16.0
        pyr.append(vol.reshape(P * h * w, 1, hl, wl))
    ref = O.corr_lookup(pyr, coords.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)       # [P,h,w,324]
    out.zero_()
    for l in range(4):
        v = ref[..., l * 81:(l + 1) * 81]
        hi = v.half()
        out[..., l * LV:l * LV + 81] = hi
        out[..., 4 * LV + l * LV:4 * LV + l * LV + 81] = (v - hi.float()).half()
    return out


def _convex_upsample(flow, mask):
    return O.convex_upsample(flow.permute(0, 3, 1, 2), mask[..., :576].permute(0, 3, 1, 2).float())


def _window_mask(mask, wh=5, ww=9):
    B, Lt = mask.shape[:2]
    return F.max_pool2d(mask.float(), (wh, ww), (wh, ww)).view(B, Lt, -1).sum(1)


def _attention(q, k, v, pk, pv, own, rolled, tind, wmask, heads=4, wh=5, ww=9, qkv_cstride=None, pkv_cstride=None,
               C_=None, impl=0, out_hw=None):
    from tests.test_ops_gpu import _attention_reference
    C_ = q.shape[-1] if C_ is None else C_
    out = _attention_reference(q[..., :C_].float(), k[..., :C_].float(), v[..., :C_].float(), pk[..., :C_].float(),
                               pv[..., :C_].float(), own.long(), rolled.long(), tind.long(), wmask, heads)
    if out_hw is not None:
        out = out[:, :, :out_hw[0], :out_hw[1]].contiguous()
    return out.to(q.dtype)


def _fold_tokens(tokens, BT, fh, fw, Cc, H, W, normalize=False, act=hip.ACT_NONE):
    # device layout is tap-major ((ky*7+kx)*C + c); F.fold wants c*49 + ky*7 + kx
    t = tokens.float().view(BT, fh * fw, 49, Cc).permute(0, 3, 2, 1).reshape(BT, Cc * 49, fh * fw)
    y = F.fold(t, (H, W), 7, 1, 3, 3)
    if normalize:
        y = y / F.fold(torch.ones_like(t), (H, W), 7, 1, 3, 3)
    return _ACT[act](y, 0.0).permute(0, 2, 3, 1).contiguous().to(tokens.dtype)


def _layernorm(x, gamma, beta, eps=1e-5):
    return F.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps).to(x.dtype)


def _layernorm_grid(x, gamma, beta, out, eps=1e-5):
    out[:, :x.shape[1], :x.shape[2]] = F.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps).to(x.dtype)
    return out


def _depthwise_pool(x, weight, bias, k=4):
    y = F.conv2d(x.float().permute(0, 3, 1, 2), weight[:, None], bias, k, 0, 1, x.shape[-1])
    return y.permute(0, 2, 3, 1).contiguous().to(x.dtype)


def _instance_norm_split(x, relu=False, eps=1e-5, residual=None, res_choff=0, relu2=False):
    y = F.instance_norm(x.float().permute(0, 3, 1, 2), eps=eps).permute(0, 2, 3, 1)
    y = F.relu(y) if relu else y
    if residual is not None:
        y = y + merge_planes(residual, res_choff, x.shape[-1])
    return split_planes(F.relu(y) if relu2 else y)


def _instance_norm(x, relu=False, eps=1e-5, out=None):
    y = F.instance_norm(x.float().permute(0, 3, 1, 2), eps=eps)
    y = (F.relu(y) if relu else y).permute(0, 2, 3, 1).contiguous().to(x.dtype)
    return y


def _upsample2x(x):
    y = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
    return y.permute(0, 2, 3, 1).contiguous().to(x.dtype)


def _dcn_act(om, mag, flow=None, fl_choff=0):
    v = om.float()
    off = mag * torch.tanh(v[..., :288])
    if flow is not None:
        f = flow.float()
        off[..., 0::2] += f[..., fl_choff + 1:fl_choff + 2]
        off[..., 1::2] += f[..., fl_choff:fl_choff + 1]
    om[..., :288] = off.to(om.dtype)
    om[..., 288:432] = torch.sigmoid(v[..., 288:432]).to(om.dtype)
    return om


def _raft_flow_taps(coords1, coords0, rows, flow_out=None, flow_choff=0, split=False):
    flow = coords1 - coords0                                               # fp32 [P,h,w,2]
    fp = F.pad(flow, (0, 0, 3, 3))                                         # zero columns left / right
    w = flow.shape[2]
    rows.zero_()
    for kx in range(7):
        _put(rows, 2 * kx, fp[:, :, kx:kx + w], split)
    if flow_out is not None:
        _put(flow_out, flow_choff, flow, split)
    return rows


def _gru_gate(zr, h, h_choff, Cc, out, out_choff, q=None):
    hv = h[..., h_choff:h_choff + Cc].float()
    if q is None:
        y = zr[..., Cc:2 * Cc].float() * hv
    else:
        z = zr[..., :Cc].float()
        y = (1 - z) * hv + z * q[..., :Cc].float()
    out[..., out_choff:out_choff + Cc] = y.to(out.dtype)
    return out


def _pack_nhwc8(srcs, out=None, split=False):
    x = torch.cat(list(srcs), 1)
    N, c, H, W = x.shape
    if out is None:
        out = torch.empty((N, H, W, 16 if split else 8), dtype=torch.float16 if split else x.dtype, device=x.device)
    out.zero_()
    _put(out, 0, x.float().permute(0, 2, 3, 1), split) if split else out[..., :c].copy_(x.permute(0, 2, 3, 1))
    return out


def _nchw_to_nhwc(x, out=None, out_choff=0, out_dtype=None, cpad=None, scale=1.0, split=False):
    N, Cc, H, W = x.shape
    if out is None:
        cp = cpad or (Cc + 7) // 8 * 8
        out = torch.zeros((N, H, W, 2 * cp if split else cp), dtype=torch.float16 if split else (out_dtype or x.dtype))
    _put(out, out_choff, (x.float() * scale).permute(0, 2, 3, 1), split)
    return out


def _nhwc_to_nchw(x, Cc=None, choff=0, out_dtype=None, act=hip.ACT_NONE):
    Cc = x.shape[-1] if Cc is None else Cc
    return _ACT[act](x[..., choff:choff + Cc].float(), 0.0).permute(0, 3, 1, 2).contiguous().to(out_dtype or x.dtype)


@contextlib.contextmanager
def emulated_device_ops():
    """Patches propainter_amd.hip / ConvLayer with the CPU emulations for the duration of the block."""
    patches = {
        "flow_warp": _flow_warp, "fb_check": _fb_check, "img_prop_step": _img_prop_step, "corr_avgpool": _corr_avgpool,
        "corr_lookup": _corr_lookup, "corr_feature_pyramid": _corr_feature_pyramid, "corr_lookup_otf": _corr_lookup_otf,
        "corr_feature_pyramid_split": _corr_feature_pyramid_split, "corr_lookup_otf_split": _corr_lookup_otf_split,
        "convex_upsample": _convex_upsample, "window_mask": _window_mask, "raft_flow_taps": _raft_flow_taps,
        "sparse_window_attention": _attention, "fold_tokens": _fold_tokens, "layernorm": _layernorm,
        "depthwise_pool": _depthwise_pool, "instance_norm": _instance_norm, "upsample2x": _upsample2x,
        "dcn_offset_mask_act": _dcn_act, "gru_gate": _gru_gate, "nchw_to_nhwc": _nchw_to_nhwc, "nhwc_to_nchw": _nhwc_to_nchw, "pack_nhwc8": _pack_nhwc8,
        "instance_norm_split": _instance_norm_split, "layernorm_grid": _layernorm_grid,
    }
    patches["require_gpu"] = lambda t, who: None
    saved = {k: getattr(hip, k) for k in patches}
    saved_call, saved_gemm = pconv.ConvLayer.__call__, pconv.batched_gemm_nt
    import propainter_amd.model.modules.flow_comp_raft as fr
    saved_fr_gemm, saved_fr_gemm_s, saved_gemm_s = fr.batched_gemm_nt, fr.batched_gemm_nt_split, pconv.batched_gemm_nt_split
    try:
        for k, f in patches.items():
            setattr(hip, k, f)
        pconv.ConvLayer.__call__ = _conv_call
        pconv.batched_gemm_nt = _batched_gemm_nt
        fr.batched_gemm_nt = _batched_gemm_nt
        fr.batched_gemm_nt_split = pconv.batched_gemm_nt_split = _batched_gemm_nt_split
        yield
    finally:
        for k, f in saved.items():
            setattr(hip, k, f)
        pconv.ConvLayer.__call__ = saved_call
        pconv.batched_gemm_nt = saved_gemm
        fr.batched_gemm_nt = saved_fr_gemm
        fr.batched_gemm_nt_split, pconv.batched_gemm_nt_split = saved_fr_gemm_s, saved_gemm_s
