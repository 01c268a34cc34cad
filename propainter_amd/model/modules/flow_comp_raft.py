"""RAFT optical flow on the MI355X engine, behind the reference's ``RAFT_bi`` interface.

Drop-in for ``model/modules/flow_comp_raft.py:27-55`` (reference): ``RAFT_bi(model_path, device)`` and
``forward(gt_local_frames[b,l_t,3,h,w], iters=20) -> (flows_f, flows_b)`` each ``[b,l_t-1,2,h,w]``; state-dict keys
are the reference's (``fix_raft.fnet.conv1.weight`` ...; the checkpoint file carries a ``module.`` prefix,
``flow_comp_raft.py:18-20``).  Only the configuration the reference instantiates is built
(small=False, alternate_corr=False; RAFT/raft.py:36-56).

Exact-math savings over the reference call pattern (results unchanged): the feature / context encoders run once
per frame instead of 3x per pair-direction; the convex-upsampling mask head runs only on the last iteration
(RAFT/raft.py:143-144 discards the rest); all pair-directions of a clip advance through the GRU as one batch.
"""
import os

import torch
import torch.nn as nn

from ... import hip
from ...conv import ConvLayer, batched_gemm_nt, batched_gemm_nt_split, fold_batchnorm
from ...param_tree import ParamTree, conv_entries, norm_entries, populate


def _encoder_schema(prefix, out_dim, batchnorm):
    e = conv_entries(f"{prefix}.conv1", 64, 3, 7)
    if batchnorm:
        e += norm_entries(f"{prefix}.norm1", 64, running=True)
    cin = 64
    for li, (dim, stride) in enumerate(((64, 1), (96, 2), (128, 2)), start=1):
        for bi in range(2):
            p = f"{prefix}.layer{li}.{bi}"
            s = stride if bi == 0 else 1
            e += conv_entries(f"{p}.conv1", dim, cin, 3) + conv_entries(f"{p}.conv2", dim, dim, 3)
            if batchnorm:
                e += norm_entries(f"{p}.norm1", dim, True) + norm_entries(f"{p}.norm2", dim, True)
            if s != 1:
                if batchnorm:
                    e += norm_entries(f"{p}.norm3", dim, True)
                e += conv_entries(f"{p}.downsample.0", dim, cin, 1)
                if batchnorm:
                    e += norm_entries(f"{p}.downsample.1", dim, True)
            cin = dim
    e += conv_entries(f"{prefix}.conv2", out_dim, 128, 1)
    return e


def raft_schema(prefix=""):
    """Key/shape list of the reference RAFT (RAFT/raft.py:36-56, extractor.py:118-166, update.py:79-125)."""
    p = prefix
    e = _encoder_schema(f"{p}fnet", 256, False) + _encoder_schema(f"{p}cnet", 256, True)
    u = f"{p}update_block"
    e += conv_entries(f"{u}.encoder.convc1", 256, 324, 1) + conv_entries(f"{u}.encoder.convc2", 192, 256, 3)
    e += conv_entries(f"{u}.encoder.convf1", 128, 2, 7) + conv_entries(f"{u}.encoder.convf2", 64, 128, 3)
    e += conv_entries(f"{u}.encoder.conv", 126, 256, 3)
    for g in "zrq":
        e += conv_entries(f"{u}.gru.conv{g}1", 128, 384, 1, 5)
    for g in "zrq":
        e += conv_entries(f"{u}.gru.conv{g}2", 128, 384, 5, 1)
    e += conv_entries(f"{u}.flow_head.conv1", 256, 128, 3) + conv_entries(f"{u}.flow_head.conv2", 2, 256, 3)
    e += conv_entries(f"{u}.mask.0", 256, 128, 3) + conv_entries(f"{u}.mask.2", 576, 256, 1)
    return e


class _RaftEngine:
    """Packed layers for one (dtype, device)."""

    split = False                 # split-plane activations (_RaftEngineSplit)

    def __init__(self, sd, dtype, device, split3=False):
        self.dtype, self.device, self.split3 = dtype, device, split3
        self.corr_otf = dtype == torch.float16       # on-the-fly correlation (fp16 MFMA); fp32 modes keep the exact volume
        self._build(sd, lambda w, b, **kw: ConvLayer(w, b, dtype=dtype, device=device, split3=split3, **kw))

    def _build(self, sd, mk):

        def enc(prefix, bn):
            def cv(name, norm=None, **kw):
                w, b = sd[f"{prefix}.{name}.weight"], sd[f"{prefix}.{name}.bias"]
                if bn and norm is not None:
                    n = f"{prefix}.{norm}"
                    w, b = fold_batchnorm(w, b, sd[n + ".weight"], sd[n + ".bias"], sd[n + ".running_mean"],
                                          sd[n + ".running_var"])
                return mk(w, b, **kw)
            L = {"conv1": cv("conv1", "norm1", stride=2, padding=3, src_channels=[3])}
            for li, stride in ((1, 1), (2, 2), (3, 2)):
                for bi in range(2):
                    p = f"layer{li}.{bi}"
                    s = stride if bi == 0 else 1
                    L[p + ".conv1"] = cv(p + ".conv1", p + ".norm1", stride=s, padding=1)
                    L[p + ".conv2"] = cv(p + ".conv2", p + ".norm2", padding=1)
                    if s != 1:
                        L[p + ".down"] = cv(p + ".downsample.0", p + ".downsample.1", stride=s)
            L["conv2"] = cv("conv2")
            return L

        self.fnet = enc("fnet", False)
        self.cnet = enc("cnet", True)
        u = "update_block."
        g = lambda n: (sd[u + n + ".weight"], sd[u + n + ".bias"])
        self.convc1 = mk(*g("encoder.convc1"), src_channels=[324])
        self.convc2 = mk(*g("encoder.convc2"), padding=1)
        # convf1 = Conv2d(2, 128, 7, padding=3) (RAFT/update.py:85,92).  As a 7x7 window over a 2-channel map the implicit GEMM
        # pads every tap to a 16-byte chunk (K = 448 for 98 real terms); the engine instead gathers the 7 horizontal taps of
        # each pixel into 16 channels (pp_raft_flow_taps) and runs a 7x1 convolution over them: K = 7 x 16, same terms, same order
        wf, bf = g("encoder.convf1")
        wrow = torch.zeros((wf.shape[0], 16, 7, 1), dtype=wf.dtype)
        for kx in range(7):
            wrow[:, 2 * kx:2 * kx + 2, :, 0] = wf[:, :, :, kx]
        self.convf1 = mk(wrow, bf, padding=(3, 0), src_channels=[16])
        self.convf2 = mk(*g("encoder.convf2"), padding=1)
        self.convm = mk(*g("encoder.conv"), padding=1, src_channels=[192, 64])
        # SepConvGRU (RAFT/update.py:45-60).  Every gate convolution reads cat[h, x], x = cat[inp, motion, flow]; inp (the
        # context features) does not change over the iterations, so its share of each convolution (1/3 of the K range) is
        # computed ONCE per pair (`*_pre`, bias included) and enters the per-iteration convolution over [h | motion, flow]
        # as a pre-activation addend.  The gate arithmetic (r * h, (1 - z) * h + z * q) runs in the epilogues.
        self.gru = []
        for s, pad in (("1", (0, 2)), ("2", (2, 0))):
            wz, bz = g("gru.convz" + s)
            wr, br = g("gru.convr" + s)
            wq, bq = g("gru.convq" + s)
            wzr, bzr = torch.cat([wz, wr], 0), torch.cat([bz, br], 0)
            it = lambda w: torch.cat([w[:, :128], w[:, 256:]], 1)
            self.gru.append(dict(
                zr_pre=mk(wzr[:, 128:256], bzr, padding=pad, src_channels=[128]),
                q_pre=mk(wq[:, 128:256], bq, padding=pad, src_channels=[128]),
                zr=mk(it(wzr), None, padding=pad, src_channels=[128, 128]),
                q=mk(it(wq), None, padding=pad, src_channels=[128, 128])))
        self.fh1 = mk(*g("flow_head.conv1"), padding=1)
        self.fh2 = mk(*g("flow_head.conv2"), padding=1)
        self.mask0 = mk(*g("mask.0"), padding=1)
        self.mask2 = mk(*g("mask.2"))

    # ---- encoders (RAFT/extractor.py:168-192); x NHWC [n,H,W,8] -> [n,H/8,W/8,256]
    def encode(self, L, x, instance_norm):
        def block(x, p, down):
            if instance_norm:
                y = hip.instance_norm(L[p + ".conv1"]([x]), relu=True)
                y = hip.instance_norm(L[p + ".conv2"]([y]), relu=True)
                if down:
                    x = hip.instance_norm(L[p + ".down"]([x]), relu=False)
                return torch.relu_(y.add_(x))
            y = L[p + ".conv1"]([x], act="relu")
            if down:
                x = L[p + ".down"]([x])
            return L[p + ".conv2"]([y], act="relu", residual=x, act2="relu")
        if instance_norm:
            x = hip.instance_norm(L["conv1"]([x]), relu=True)
        else:
            x = L["conv1"]([x], act="relu")
        for li in (1, 2, 3):
            x = block(x, f"layer{li}.0", li > 1)
            x = block(x, f"layer{li}.1", False)
        return L["conv2"]([x])

    # ---- iterative update for a batch of pair-directions
    def refine(self, f1, f2, ctx, iters):
        """f1, f2, ctx: NHWC [P,h,w,256].  Returns fp32 flow_up [P,2,8h,8w]."""
        P, h, w, _ = f1.shape
        dev, dt = f1.device, self.dtype
        n8 = h * w
        if self.corr_otf:
            # fp16 engine: no all-pairs volume.  avg_pool(f1 . f2) = f1 . avg_pool(f2), so f2 is pooled once per pair and
            # every iteration computes exactly the dot products its 9x9x4 windows touch (csrc/raft_corr_otf.hip)
            f2_levels = hip.corr_feature_pyramid(f2)
            lookup = lambda coords, out: hip.corr_lookup_otf(f1, f2_levels, coords, out)
        else:
            # all-pairs correlation volume + pyramid (RAFT/corr.py:13-27,52-60), fp32
            vol = batched_gemm_nt(f1.view(P, n8, 256), f2.view(P, n8, 256), out_scale=1.0 / 16.0, split3=self.split3)
            levels = [vol.view(P * n8, h, w)]
            hh, ww = h, w
            for _ in range(3):
                levels.append(hip.corr_avgpool(levels[-1], P * n8, hh, ww))
                hh, ww = hh // 2, ww // 2
            lookup = lambda coords, out: hip.corr_lookup(levels, coords, out)
        net = torch.tanh(ctx[..., :128]).contiguous()
        inp = torch.relu(ctx[..., 128:]).contiguous()
        pre = [(G["zr_pre"]([inp]), G["q_pre"]([inp])) for G in self.gru]     # iteration-invariant partial sums
        xbuf = torch.empty((P, h, w, 128), dtype=dt, device=dev)            # [motion(126) | flow(2)]
        ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32),
                                torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
        coords0 = torch.stack([xs, ys], -1)[None].expand(P, h, w, 2).contiguous()
        coords1 = coords0.clone()
        corr = torch.empty((P, h, w, 328), dtype=dt, device=dev)
        frow = torch.empty((P, h, w, 16), dtype=dt, device=dev)            # 7 horizontal flow taps per pixel (convf1 input)
        zbuf = torch.empty((P, h, w, 128), dtype=dt, device=dev)
        rh = torch.empty((P, h, w, 128), dtype=dt, device=dev)
        delta = torch.zeros((P, h, w, 8), dtype=torch.float32, device=dev)
        for it in range(iters):
            lookup(coords1, corr)
            hip.raft_flow_taps(coords1, coords0, frow, flow_out=xbuf, flow_choff=126)       # flow = coords1 - coords0
            cor = self.convc2([self.convc1([corr], act="relu")], act="relu")
            flo = self.convf2([self.convf1([frow], act="relu")], act="relu")
            self.convm([cor, flo], out=xbuf, out_choff=0, act="relu")
            for G, (pzr, pq) in zip(self.gru, pre):
                G["zr"]([net, xbuf], out=zbuf, act="sigmoid", preadd=pzr,
                        fuse=dict(kind="gru_zr", h=net, out2=rh, split=128))          # zbuf <- z, rh <- r * h
                G["q"]([rh, xbuf], out=net, act="tanh", preadd=pq,
                       fuse=dict(kind="gru_h", h=net, z=zbuf))                       # net <- (1 - z) * h + z * q
            self.fh2([self.fh1([net], act="relu")], out=delta, out_dtype=torch.float32)
            coords1 = coords1 + delta[..., :2]
        mask = self.mask2([self.mask0([net], act="relu")], out_scale=0.25)
        return hip.convex_upsample((coords1 - coords0).contiguous(), mask)


class _RaftEngineSplit(_RaftEngine):
    """Precision "f16x3": RAFT at the reference's precision class (it keeps RAFT fp32 even under --fp16,
    inference_propainter.py:311) on the fp16 matrix cores and the LDS-DMA kernels.

    Every activation is a SPLIT-PLANE tensor: fp16 [..., 2C] whose first C channels hold hi = fp16(v) and whose last C
    channels hold lo = fp16(v - hi) of the fp32 value v (22 significand bits); weights are split the same way when they
    are packed.  A convolution computes hi*W_hi + lo*W_hi + hi*W_lo with fp32 accumulation (the dropped lo*W_lo term is
    < 2^-22 relative): its K table walks every 64-channel block three times (conv.split_ktable), so the halo-tile /
    LDS-DMA kernels of the fp16 engine run unchanged; only their epilogues read operands as hi + lo and write two planes
    (csrc/conv_epilogue.h, SPLIT).  Coordinates, flow, correlation values and the up-sampling mask stay fp32.  The
    all-pairs correlation volume is kept (fp32, RAFT/corr.py:13-27): its GEMM runs on the same kernels
    (conv.batched_gemm_nt_split)."""

    def __init__(self, sd, device):
        self.dtype, self.device, self.split3, self.split = torch.float16, device, False, True
        self.corr_otf = False
        mk = lambda w, b, **kw: ConvLayer(w, b, dtype=torch.float16, device=device, split=True, **kw)
        self._build(sd, mk)
        # the context encoder's output convolution feeds tanh (GRU state) and relu (context input) halves
        # (RAFT/raft.py:113-116): two convolutions with the activation in the epilogue instead of a split + two passes
        w2, b2 = sd["cnet.conv2.weight"], sd["cnet.conv2.bias"]
        self.cnet["conv2_net"] = mk(w2[:128], b2[:128])
        self.cnet["conv2_inp"] = mk(w2[128:], b2[128:])
        # Correlation WITHOUT the all-pairs volume (round 4; csrc/raft_corr_otf.hip, corr_otf_split_kernel): per iteration the tri-product
        # dot products of the 4 x (10 x 10) neighbourhoods, blended as pp_corr_lookup does.  Its output planes carry 88 channels per
        # level (81 taps + 7 zeros), so convc1's weight columns are spread accordingly: 11 full 32-channel blocks per plane.
        # PP_RAFT_SPLIT_VOLUME=1 keeps the fp32 volume + pyramid GEMMs + pp_corr_lookup (A/B, and the reference's own call pattern).
        self.corr_otf = os.environ.get("PP_RAFT_SPLIT_VOLUME", "0") != "1"
        LV = hip.OTF_SPLIT_LEVEL_CHANNELS
        wc1, bc1 = sd["update_block.encoder.convc1.weight"], sd["update_block.encoder.convc1.bias"]
        wsp = torch.zeros((wc1.shape[0], 4 * LV, 1, 1), dtype=wc1.dtype)
        for l in range(4):
            wsp[:, l * LV:l * LV + 81] = wc1[:, l * 81:(l + 1) * 81]
        self.convc1_otf = mk(wsp, bc1, src_channels=[4 * LV])
        self.pool_volume = os.environ.get("PP_RAFT_POOL_VOLUME", "0") == "1"      # levels 1..3 by pooling the level-0 volume (A/B)
        self.volume_impl = int(os.environ.get("PP_RAFT_VOLUME_IMPL", "0"))        # tile configuration of the volume GEMMs (0 = auto)

    def encode(self, L, x, instance_norm):
        """x split-plane NHWC [n,H,W,16] -> split-plane [n,H/8,W/8,512]; for the context encoder (instance_norm False) the
        pair (tanh half, relu half), each split-plane [n,H/8,W/8,256]."""
        f32 = torch.float32
        if instance_norm:
            def block(x, p, down):
                y = hip.instance_norm_split(L[p + ".conv1"]([x], out_dtype=f32), relu=True)
                if down:
                    x = hip.instance_norm_split(L[p + ".down"]([x], out_dtype=f32), relu=False)
                return hip.instance_norm_split(L[p + ".conv2"]([y], out_dtype=f32), relu=True, residual=x, relu2=True)
            x = hip.instance_norm_split(L["conv1"]([x], out_dtype=f32), relu=True)
        else:
            def block(x, p, down):
                y = L[p + ".conv1"]([x], act="relu")
                if down:
                    x = L[p + ".down"]([x])
                return L[p + ".conv2"]([y], act="relu", residual=x, act2="relu")
            x = L["conv1"]([x], act="relu")
        for li in (1, 2, 3):
            x = block(x, f"layer{li}.0", li > 1)
            x = block(x, f"layer{li}.1", False)
        if instance_norm:
            return L["conv2"]([x])
        return L["conv2_net"]([x], act="tanh"), L["conv2_inp"]([x], act="relu")

    def refine(self, f1, f2, ctx, iters):
        """f1, f2 split-plane [P,h,w,512]; ctx = (net0, inp) split-plane [P,h,w,256] each.  Returns fp32 flow_up [P,2,8h,8w]."""
        net0, inp = ctx
        P, h, w, _ = f1.shape
        dev, dt = f1.device, torch.float16
        n8 = h * w
        if self.corr_otf:
            # no all-pairs volume: avg_pool(f1 . f2) = f1 . avg_pool(f2), so f2 is pooled once per pair (fp32 means, stored as planes) and
            # every iteration computes exactly the dot products its 9 x 9 x 4 windows touch, three fp16 products per product
            f2c = f2.contiguous()
            f2_levels = [f2c] + hip.corr_feature_pyramid_split(f2c)
            f1c = f1.contiguous()
            corr = torch.empty((P, h, w, 8 * hip.OTF_SPLIT_LEVEL_CHANNELS), dtype=dt, device=dev)
            lookup = lambda coords: hip.corr_lookup_otf_split(f1c, f2_levels, coords, corr)
            convc1 = self.convc1_otf
        else:
            # all-pairs correlation pyramid (RAFT/corr.py:13-27,52-60), fp32.  Levels 1..3 are GEMMs of f1 with the pooled split-plane
            # features (fp32 means, 3 fp16 products per product like level 0) instead of three pooling passes that re-read the level-0
            # volume (829 MB per pair-direction at 720p); PP_RAFT_POOL_VOLUME=1 keeps the pooling passes
            gemm = lambda b_: batched_gemm_nt_split(f1.view(P, n8, 512), b_.view(P, -1, 512), out_scale=1.0 / 16.0, impl=self.volume_impl)
            levels = [gemm(f2).view(P * n8, h, w)]
            if self.pool_volume:
                hh, ww = h, w
                for _ in range(3):
                    levels.append(hip.corr_avgpool(levels[-1], P * n8, hh, ww))
                    hh, ww = hh // 2, ww // 2
            else:
                levels += [gemm(fl).view(P * n8, fl.shape[1], fl.shape[2]) for fl in hip.corr_feature_pyramid_split(f2.contiguous())]
            corr = torch.empty((P, h, w, 656), dtype=dt, device=dev)
            lookup = lambda coords: hip.corr_lookup(levels, coords, corr, split=True)
            convc1 = self.convc1
        net = net0.clone()
        pre = [(G["zr_pre"]([inp]), G["q_pre"]([inp])) for G in self.gru]     # iteration-invariant partial sums
        xbuf = torch.empty((P, h, w, 256), dtype=dt, device=dev)            # [motion(126) | flow(2)] x (hi, lo)
        ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32),
                                torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
        coords0 = torch.stack([xs, ys], -1)[None].expand(P, h, w, 2).contiguous()
        coords1 = coords0.clone()
        frow = torch.empty((P, h, w, 32), dtype=dt, device=dev)
        zbuf = torch.empty((P, h, w, 256), dtype=dt, device=dev)
        rh = torch.empty((P, h, w, 256), dtype=dt, device=dev)
        delta = torch.zeros((P, h, w, 8), dtype=torch.float32, device=dev)
        for it in range(iters):
            lookup(coords1)
            hip.raft_flow_taps(coords1, coords0, frow, flow_out=xbuf, flow_choff=126, split=True)
            cor = self.convc2([convc1([corr], act="relu")], act="relu")
            flo = self.convf2([self.convf1([frow], act="relu")], act="relu")
            self.convm([cor, flo], out=xbuf, out_choff=0, act="relu")
            for G, (pzr, pq) in zip(self.gru, pre):
                G["zr"]([net, xbuf], out=zbuf, act="sigmoid", preadd=pzr, fuse=dict(kind="gru_zr", h=net, out2=rh, split=128))
                G["q"]([rh, xbuf], out=net, act="tanh", preadd=pq, fuse=dict(kind="gru_h", h=net, z=zbuf))
            self.fh2([self.fh1([net], act="relu")], out=delta, out_dtype=torch.float32)
            coords1 = coords1 + delta[..., :2]
        mask = self.mask2([self.mask0([net], act="relu")], out_scale=0.25, out_dtype=torch.float32)
        return hip.convex_upsample((coords1 - coords0).contiguous(), mask)


def assert_finite_flows(raft):
    """Raises FloatingPointError when the last ``RAFT_bi.forward`` produced a non-finite flow.  Host-synchronising (reads a one-element
    device flag): call it where the pass is synchronised anyway (after the D2H of the composited frames).

    Why it exists: the default RAFT precision under ``--fp16`` is "f16x3" -- fp32-class values stored as two fp16 planes (hi = fp16(v),
    lo = fp16(v - hi)).  That format has fp16's EXPONENT range: an activation or weight with |v| > 65504 becomes inf in its hi plane and
    every flow that depends on it NaN (never a silently wrong finite value; tests/test_split_plane_gpu.py::test_split_plane_overflow_is_loud).
    With seeded and released-style weights RAFT's activations stay below ~1e3 (features are instance-normalised, correlations are scaled by
    1/16, flows are pixels), so the guard is a backstop for foreign checkpoints or corrupt inputs; ``precision="f32"`` (exact fp32 matrix
    instructions) has fp32's range."""
    flags = (getattr(raft, "_flows_finite", None) or []) + [(f, False) for f, _ in (getattr(raft, "_flows_finite_eager", None) or [])]
    if flags and flags[0][0].is_cuda:
        torch.cuda.synchronize(flags[0][0].device)     # the flags were written on whatever streams the forwards ran on
    raft._flows_finite_eager = []
    if flags and not bool(torch.stack([f.reshape(()) for f, _ in flags]).all()):
        prec = getattr(raft, "_flows_precision", "?")
        hint = ("a value left fp16's range (|v| > 65504) in the split-plane engine: run RAFT_bi(precision='f32') (CLI: --raft_fp32)"
                if prec == "f16x3" else "check the input frames / checkpoint")
        raise FloatingPointError(f"RAFT produced non-finite flows at precision {prec!r}; {hint}")


class RAFT_bi(nn.Module):
    """Bidirectional RAFT flow of consecutive frame pairs (reference: model/modules/flow_comp_raft.py:27-55)."""

    PRECISIONS = ("f32", "f16x3", "f16")

    def __init__(self, model_path='weights/raft-things.pth', device='cuda', compute_dtype=None, max_pairs=None,
                 precision=None):
        """``precision`` (engine extension; the reference runs RAFT in fp32 even under ``--fp16``,
        inference_propainter.py:311):
          "f32"   exact fp32 products on ``v_mfma_f32_16x16x4_f32`` (157 TFLOP/s peak) -- the default for fp32 input;
          "f16x3" fp32 tensors everywhere, every product on the fp16 matrix cores as hi*hi + hi*lo + lo*hi with fp32
                  accumulation: relative error ~2^-21 per product (fp32: 2^-24) -- reference-class flows at 5x the
                  exact-fp32 matrix rate;
          "f16"   fp16 activations / weights, fp32 accumulation, fp32 correlation values, coordinates and flow (the
                  10-bit mantissa of the TF32 convolutions a CUDA build of the reference runs by default) -- the default
                  for fp16 input.
        ``compute_dtype=torch.float16`` is the older spelling of precision="f16"."""
        super().__init__()
        self.fix_raft = ParamTree()
        populate(self.fix_raft, raft_schema())
        if model_path is not None:
            ckpt = torch.load(model_path, map_location='cpu')
            ckpt = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in ckpt.items()}
            self.fix_raft.load_state_dict(ckpt, strict=True)
        for p in self.parameters():
            p.requires_grad = False
        if precision is not None and precision not in self.PRECISIONS:
            raise ValueError(f"RAFT precision {precision!r} not in {self.PRECISIONS}")
        self.precision = precision             # None: follow compute_dtype / the input dtype
        self.compute_dtype = compute_dtype
        # Every pair is computed independently of its batch neighbours (InstanceNorm per sample, BatchNorm folded), so a
        # clip driver may hand over all frames at once instead of the reference's 12/8/4/2-frame clips
        # (inference_propainter.py:302-330): identical flows, each frame encoded once, larger GEMMs.
        self.batch_invariant = True
        self.supports_streams = True          # forward(..., streams=n): encoders / pair groups on n HIP streams
        self.max_pairs = max_pairs
        # fp32 all-pairs correlation volumes + pyramids of the pair-direction chunks in flight (volume modes "f32" / "f16x3")
        self.volume_budget_bytes = float(os.environ.get("PP_RAFT_VOLUME_GB", "40")) * 1e9
        self._engines = {}
        self._flows_finite = []                # device flags of the forwards recorded under hipGraph capture (refreshed by every replay)
        self._flows_finite_eager = []          # (flag, event) of the eager forwards since the last assert_finite_flows
        self.to(device)
        self.eval()

    def _get_engine(self, precision, device):
        key = (precision, str(device), sum(p._version for p in self.parameters()))
        eng = self._engines.get(precision)
        if eng is None or eng[0] != key:
            sd = {k: v.detach().float().cpu() for k, v in self.fix_raft.state_dict().items()}
            if precision == "f16x3":
                eng = (key, _RaftEngineSplit(sd, device))
            else:
                eng = (key, _RaftEngine(sd, torch.float16 if precision == "f16" else torch.float32, device))
            self._engines[precision] = eng
        return eng[1]

    @hip.on_input_device
    @torch.no_grad()
    def forward(self, gt_local_frames, iters=20, streams=1):
        b, l_t, c, h, w = gt_local_frames.size()
        hip.require_gpu(gt_local_frames, "RAFT_bi")
        if h % 8 or w % 8 or h < 128 or w < 128:
            raise ValueError(f"RAFT needs H, W multiples of 8 and >= 128 (got {h}x{w}; RAFT/utils/utils.py:61-62)")
        prec = self.precision or ("f16" if (self.compute_dtype or gt_local_frames.dtype) == torch.float16 else "f32")
        eng = self._get_engine(prec, gt_local_frames.device)
        dt = eng.dtype
        fr = gt_local_frames.reshape(b * l_t, c, h, w)
        # encoders once per frame, in frame chunks that keep every activation below 2 GiB (32-bit buffer offsets of the
        # LDS-DMA gather; InstanceNorm statistics are per frame, so chunking does not change results)
        fchunk = max(1, (1 << 30) // (h * w * 16 * 4))     # largest activation: [H/2, W/2, 64] per frame, <= 1 GiB in fp32
        # (streams > 1, engine extension: the two encoders of a chunk, and below the pair-direction groups, run on separate HIP
        # streams -- forked from / joined to the current one, parallel branches under hipGraph capture.  Every frame / pair is
        # computed independently of its batch neighbours, so the flows are identical.)
        dev = gt_local_frames.device
        fm, cx_ = [], []
        for s in range(0, b * l_t, fchunk):
            if eng.split:     # split-plane engine: [n, H, W, 8 hi | 8 lo] fp16 planes of the fp32 frames
                x = hip.pack_nhwc8([fr[s:s + fchunk].contiguous().float()], split=True)
            else:
                x = hip.nchw_to_nhwc(fr[s:s + fchunk].contiguous(), out_dtype=dt, cpad=8)
            f_, c_ = hip.fork_join(dev, [lambda: eng.encode(eng.fnet, x, True), lambda: eng.encode(eng.cnet, x, False)], streams)
            fm.append(f_)
            cx_.append(c_ if isinstance(c_, (tuple, list)) else (c_,))      # (the split-plane engine returns the (tanh, relu) halves)
        h8, w8 = h // 8, w // 8
        # pair-directions: forward pairs (i, i + 1) then backward pairs (i + 1, i); the context comes from the first frame of a pair
        def frames_of(parts):
            return parts[0] if len(parts) == 1 else torch.cat(parts, 0)             # [b * l_t, h8, w8, C]
        def pairs(t, first):
            t = t.view(b, l_t, h8, w8, t.shape[-1])
            fw, bw = t[:, :-1].reshape(-1, h8, w8, t.shape[-1]), t[:, 1:].reshape(-1, h8, w8, t.shape[-1])
            return torch.cat([fw, bw], 0) if first else torch.cat([bw, fw], 0)
        tf, tcs = frames_of(fm), [frames_of([c[k] for c in cx_]) for k in range(len(cx_[0]))]
        P = 2 * b * (l_t - 1)
        n8 = h8 * w8
        lanes = streams if (streams > 1 and P >= 2 * streams) else 1
        if eng.corr_otf:      # largest activation of the update block: the [P, h8, w8, 328] (split-plane: 2 x 352) lookup tile, < 2 GiB (32-bit buffer offsets)
            chunk = self.max_pairs or max(1, ((1 << 31) - 1) // (n8 * (8 * hip.OTF_SPLIT_LEVEL_CHANNELS if eng.split else 328) * 2))
        else:                 # fp32 all-pairs pyramid: 1.34 x n8^2 x 4 bytes per pair-direction, 40 GB over the chunks in flight
            per_pair = n8 * n8 * 4 * 1.34
            # (at least 4 pair-directions per chunk: at 1080x1920 a pair's pyramid is 5.6 GB and 40 GB over 3 lanes would leave 2-pair
            #  chunks -- convolution launches over 64 800 pixels that do not fill the chip)
            by_budget = int(self.volume_budget_bytes // lanes // per_pair)
            floor_ = min(4, -(-P // lanes))
            if by_budget < floor_:
                # the floor may exceed the BUDGET (a soft target) but never what the device can actually hold: volumes of every lane in
                # flight + 25 % for the update block's activations must fit the free memory (1440p: a pair's pyramid is ~18 GB)
                free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
                fit = int(free * 0.75 // lanes // per_pair)
                if fit < floor_:
                    floor_ = max(1, min(floor_, fit))
                if floor_ > max(by_budget, 1) and not getattr(self, "_warned_budget", False):
                    self._warned_budget = True
                    import warnings
                    warnings.warn(f"RAFT volume budget {self.volume_budget_bytes / 1e9:.0f} GB < {lanes} lanes x {floor_} pair-directions x "
                                  f"{per_pair / 1e9:.1f} GB = {lanes * floor_ * per_pair / 1e9:.0f} GB: running {floor_}-pair chunks "
                                  f"(PP_RAFT_VOLUME_GB / RAFT_bi.volume_budget_bytes raise the budget, streams=1 lowers the need)")
            chunk = self.max_pairs or max(floor_, by_budget)
            if eng.split:     # ... and the largest split-plane activation (the [P, h8, w8, 2 x 328] lookup tile) below 2 GiB
                chunk = min(chunk, max(1, ((1 << 31) - 1) // (n8 * 656 * 2)))
        if lanes > 1:
            chunk = min(chunk, -(-P // streams))
        if b == 1:
            # one clip: the pairs of a direction are CONSECUTIVE frames, so a chunk's first / second frames and context are plain views of
            # the per-frame maps (no gathered copies of the feature maps: 9 ms and 9 GB per 80-frame 720p clip); chunks do not straddle the
            # two directions and are balanced within one (every pair is computed independently of its chunk neighbours: same flows)
            nd = l_t - 1
            nch = -(-nd // chunk)
            csz = -(-nd // nch)
            parts = []
            for d in (0, 1):                              # forward pairs (i, i + 1), then backward pairs (i + 1, i)
                for s_ in range(0, nd, csz):
                    n_ = min(csz, nd - s_)
                    a0, b0 = s_ + d, s_ + 1 - d
                    parts.append((tf[a0:a0 + n_], tf[b0:b0 + n_], [c[a0:a0 + n_] for c in tcs]))
        else:
            f1, f2 = pairs(tf, True), pairs(tf, False)
            cxs = [pairs(c, True) for c in tcs]
            parts = [(f1[i:i + chunk].contiguous(), f2[i:i + chunk].contiguous(), [c[i:i + chunk].contiguous() for c in cxs])
                     for i in range(0, P, chunk)]
        ups = hip.fork_join(dev, [(lambda a=a, b_=b_, c_=c_: eng.refine(a, b_, c_[0] if len(c_) == 1 else tuple(c_), iters))
                                  for a, b_, c_ in parts], streams)
        up32 = torch.cat(ups, 0)
        up = up32.to(gt_local_frames.dtype)
        # finite-flow guard (device flag, no host sync here: graph-capturable; callers check it after their own synchronisation with
        # assert_finite_flows).  A split-plane value beyond fp16's range (|v| > 65504) puts inf into its hi plane and the flows go NaN.
        # one flag per call (a sharded / streaming pass calls once per rank); flags recorded under hipGraph capture are refreshed by every
        # replay and stay, eager ones are dropped once assert_finite_flows has looked at them
        # (the flag is taken on the fp32 flows BEFORE the cast to the caller's dtype: an fp16 sum over millions of flow values overflows
        #  by itself; eager flags are folded into one running AND, so no call's flag is ever dropped -- a long clip makes many calls)
        captured = up.is_cuda and torch.cuda.is_current_stream_capturing()
        flag = torch.isfinite(up32).all()
        if captured:
            flags = getattr(self, "_flows_finite", None) or []
            flags.append((flag, True))
            self._flows_finite = [fc for fc in flags if not fc[1]] + [fc for fc in flags if fc[1]][-256:]
        else:
            # eager flags: kept one by one WITH an event recorded behind them on the stream that wrote them (another logical rank / lane may
            # call next on another stream: nothing else orders the two); assert_finite_flows stacks them after its synchronisation.  A caller
            # that never asserts (a long-running server) is bounded by folding 256 of them into one, each awaited through its event
            ev = None
            if up.is_cuda:                 # (the CPU emulation of the host-logic tests has no streams)
                ev = torch.cuda.Event()
                ev.record()
            eager = getattr(self, "_flows_finite_eager", None) or []
            eager.append((flag, ev))
            if len(eager) > 256:
                if up.is_cuda:
                    cur = torch.cuda.current_stream(up.device)
                    for _, e in eager:
                        if e is not None:
                            cur.wait_event(e)
                folded = torch.stack([f.reshape(()) for f, _ in eager]).all()
                if up.is_cuda:
                    ev = torch.cuda.Event()
                    ev.record()
                eager = [(folded, ev)]
            self._flows_finite_eager = eager
        self._flows_precision = prec
        half = P // 2
        return up[:half].view(b, l_t - 1, 2, h, w), up[half:].view(b, l_t - 1, 2, h, w)
