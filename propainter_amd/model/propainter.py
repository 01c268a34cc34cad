"""ProPainter generator on the MI355X engine, behind the reference's ``InpaintGenerator`` interface.

Drop-in for ``model/propainter.py:256-372`` (reference): ``InpaintGenerator(init_weights=True, model_path=None)``,
``img_propagation(masked_frames, completed_flows, masks, interpolation='nearest')`` and
``forward(masked_frames, completed_flows, masks_in, masks_updated, num_local_frames, interpolation='bilinear',
t_dilation=2)``; identical state-dict keys (incl. the ``valid_ind_rolled`` buffers).

Engine notes (all algebraically identical to the reference, see oracle/propainter_oracle.py):
  * SoftSplit = unfold(7,3,3) + Linear  ==  one 7x7/stride-3 implicit-GEMM convolution (no 1.45 GB unfold);
  * FusionFeedForward: fc1 GEMM -> fused fold/normalise/GELU gather -> fc2 as a 7x7/stride-3 convolution over the
    40-channel folded map (GELU commutes with the zero-padded unfold);
  * sparse window attention gathers rolled / pooled keys through index tables instead of materialising them, and
    reads the masked-window flags on the device (no nonzero() host sync);
  * q/k/v (and pooled k/v) projections are single fused GEMMs; every torch.cat of the reference is a multi-source
    convolution; forward-backward consistency checks of all propagation steps run as one batched launch.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip
from ..conv import ConvLayer
from ..param_tree import ParamTree, conv_entries, linear_entries, populate

WIN = (5, 9)
HEADS = 4
HIDDEN = 512
DEPTH = 8


def token_grid(n):
    """model/modules/sparse_transformer.py:20-23 (kernel 7, stride 3, padding 3)."""
    return int((n + 2 * 3 - 6 - 1) / 3 + 1)


def generator_schema():
    e = []
    enc = ((0, 64, 5), (2, 64, 64), (4, 128, 64), (6, 256, 128), (8, 384, 256), (10, 512, 640 // 2), (12, 384, 768 // 4),
           (14, 256, 640 // 8), (16, 128, 512))
    for idx, cout, cin in enc:
        e += conv_entries(f"encoder.layers.{idx}", cout, cin, 3)
    e += conv_entries("decoder.0.conv", 128, 128, 3) + conv_entries("decoder.2", 64, 128, 3)
    e += conv_entries("decoder.4.conv", 64, 64, 3) + conv_entries("decoder.6", 3, 64, 3)
    e += linear_entries("ss.embedding", HIDDEN, 128 * 49)
    e += linear_entries("sc.embedding", 128 * 49, HIDDEN) + conv_entries("sc.bias_conv", 128, 128, 3)
    fp = "feat_prop_module"
    for m in ("backward_1", "forward_1"):
        d = f"{fp}.deform_align.{m}"
        e += [(f"{d}.weight", (128, 128, 3, 3), "w"), (f"{d}.bias", (128,), "b")]
        e += conv_entries(f"{d}.conv_offset.0", 128, 2 * 128 + 2 + 1 + 2, 3)
        e += conv_entries(f"{d}.conv_offset.2", 128, 128, 3) + conv_entries(f"{d}.conv_offset.4", 128, 128, 3)
        e += conv_entries(f"{d}.conv_offset.6", 432, 128, 3)
        e += conv_entries(f"{fp}.backbone.{m}.0", 128, 258, 3) + conv_entries(f"{fp}.backbone.{m}.2", 128, 128, 3)
    e += conv_entries(f"{fp}.fuse.0", 128, 258, 3) + conv_entries(f"{fp}.fuse.2", 128, 128, 3)
    for i in range(DEPTH):
        t = f"transformers.transformer.{i}"
        for n in ("key", "query", "value", "proj"):
            e += linear_entries(f"{t}.attention.{n}", HIDDEN, HIDDEN)
        e += [(f"{t}.attention.pool_layer.weight", (HIDDEN, 1, 4, 4), "w"), (f"{t}.attention.pool_layer.bias", (HIDDEN,), "b")]
        e += [(f"{t}.norm1.weight", (HIDDEN,), "one"), (f"{t}.norm1.bias", (HIDDEN,), "zero"),
              (f"{t}.norm2.weight", (HIDDEN,), "one"), (f"{t}.norm2.bias", (HIDDEN,), "zero")]
        e += linear_entries(f"{t}.mlp.fc1.0", 1960, HIDDEN) + linear_entries(f"{t}.mlp.fc2.1", HIDDEN, 1960)
    return e


class _GenEngine:
    def __init__(self, sd, dtype, device):
        self.dtype, self.device = dtype, device
        mk = lambda w, b, **kw: ConvLayer(w, b, dtype=dtype, device=device, **kw)
        g = lambda n: (sd[n + ".weight"], sd[n + ".bias"])
        E = "encoder.layers."
        self.enc = [mk(*g(E + "0"), stride=2, padding=1, src_channels=[5]), mk(*g(E + "2"), padding=1),
                    mk(*g(E + "4"), stride=2, padding=1), mk(*g(E + "6"), padding=1), mk(*g(E + "8"), padding=1)]
        # grouped layers consume the group-wise interleave of x0 (256 ch) and the running feature (:226-230)
        self.enc_g = []
        prev_c = 384
        for idx, groups in ((10, 2), (12, 4), (14, 8), (16, 1)):
            w, b = g(E + str(idx))
            self.enc_g.append(mk(w, b, padding=1, groups=groups, src_channels=[256 // groups, prev_c // groups]))
            prev_c = w.shape[0]
        self.dec = [mk(*g("decoder.0.conv"), padding=1), mk(*g("decoder.2"), padding=1),
                    mk(*g("decoder.4.conv"), padding=1), mk(*g("decoder.6"), padding=1)]
        self.ss = mk(sd["ss.embedding.weight"].view(HIDDEN, 128, 7, 7), sd["ss.embedding.bias"], stride=3, padding=3)
        # fold kernels read tap-major token features (ky*7+kx)*C + c: permute the producing Linear's rows (exact)
        tap_major = lambda w, b, c: (w.view(c, 49, -1).permute(1, 0, 2).reshape(c * 49, -1), b.view(c, 49).t().reshape(-1))
        self.sc_embed = mk(*tap_major(*g("sc.embedding"), 128))
        self.sc_bias = mk(*g("sc.bias_conv"), padding=1)
        fp = "feat_prop_module."
        self.prop = {}
        for m in ("backward_1", "forward_1"):
            d = fp + f"deform_align.{m}"
            self.prop[m] = dict(
                off0=mk(*g(d + ".conv_offset.0"), padding=1, src_channels=[128, 128, 5]),
                off2=mk(*g(d + ".conv_offset.2"), padding=1),
                off4=mk(*g(d + ".conv_offset.4"), padding=1),
                off6=mk(*g(d + ".conv_offset.6"), padding=1),
                dcn=mk(sd[d + ".weight"], sd[d + ".bias"], padding=1, dcn_groups=16),
                bb0=mk(*g(fp + f"backbone.{m}.0"), padding=1, src_channels=[128, 128, 2]),
                bb2=mk(*g(fp + f"backbone.{m}.2"), padding=1))
        self.fuse0 = mk(*g(fp + "fuse.0"), padding=1, src_channels=[128, 128, 2])
        self.fuse2 = mk(*g(fp + "fuse.2"), padding=1)
        self.blocks = []
        f32 = lambda t: t.float().to(device).contiguous()
        for i in range(DEPTH):
            t = f"transformers.transformer.{i}."
            a = t + "attention."
            qkv_w = torch.cat([sd[a + "query.weight"], sd[a + "key.weight"], sd[a + "value.weight"]], 0)
            qkv_b = torch.cat([sd[a + "query.bias"], sd[a + "key.bias"], sd[a + "value.bias"]], 0)
            kv_w = torch.cat([sd[a + "key.weight"], sd[a + "value.weight"]], 0)
            kv_b = torch.cat([sd[a + "key.bias"], sd[a + "value.bias"]], 0)
            self.blocks.append(dict(
                n1=(f32(sd[t + "norm1.weight"]), f32(sd[t + "norm1.bias"])),
                n2=(f32(sd[t + "norm2.weight"]), f32(sd[t + "norm2.bias"])),
                qkv=mk(qkv_w, qkv_b), kv=mk(kv_w, kv_b), proj=mk(*g(a + "proj")),
                pool_w=f32(sd[a + "pool_layer.weight"].view(HIDDEN, 4, 4)), pool_b=f32(sd[a + "pool_layer.bias"]),
                fc1=mk(*tap_major(*g(t + "mlp.fc1.0"), 40)),
                fc2=mk(sd[t + "mlp.fc2.1.weight"].view(HIDDEN, 40, 7, 7), sd[t + "mlp.fc2.1.bias"], stride=3, padding=3)))
        self._win_cache = {}
        self._tind_cache = {}
        self.prop_batch_bytes = 2.5e9        # propagate_windows: bytes of one step-major batch of window frames

    # ------------------------------------------------------------------ encoder / decoder
    def encode(self, x):
        """Encoder.forward (:218-232); x NHWC [n,H,W,8] -> [n,H/4,W/4,128]."""
        for layer in self.enc[:4]:
            x = layer([x], act="lrelu", act_param=0.2)
        x0 = x
        out = self.enc[4]([x0], act="lrelu", act_param=0.2)
        for layer in self.enc_g:
            out = layer([x0, out], act="lrelu", act_param=0.2)
        return out

    def decode(self, x):
        """decoder (:266-273) + tanh (:370); x [n,h,w,128] -> planar [n,3,4h,4w]."""
        x = self.dec[0]([hip.upsample2x(x)], act="lrelu", act_param=0.2)
        x = self.dec[1]([x], act="lrelu", act_param=0.2)
        x = self.dec[2]([hip.upsample2x(x)], act="lrelu", act_param=0.2)
        x = self.dec[3]([x], act="tanh")
        return hip.nhwc_to_nchw(x, 3)

    # ------------------------------------------------------------------ feature propagation (:104-190, learnable)
    def propagation_rows(self, flows_f, flows_b, mask2):
        """The per-pair side inputs of the propagation steps: flows_* [n-1,h,w,2] NHWC (1/4-res, already /4), mask2 [n,h,w,2] ->
        (aux_b, aux_f, mk8).  aux_b[j] = [flow_f[j] | valid(flow_f[j], flow_b[j]) | mask2[j] | 0 0 0] is what the BACKWARD pass needs when
        it moves from frame j + 1 to frame j, aux_f[j] = [flow_b[j] | valid(flow_b[j], flow_f[j]) | mask2[j + 1] | 0 0 0] what the forward
        pass needs from frame j to j + 1 (model/propainter.py:137-160); mk8 = mask2 padded to 8 channels.  All forward-backward
        checks of a direction run as one launch."""
        n, h, w, _ = mask2.shape
        dev, dt = mask2.device, self.dtype
        mk8 = torch.zeros((n, h, w, 8), dtype=dt, device=dev)
        mk8[..., :2] = mask2
        aux_b = torch.zeros((max(n - 1, 0), h, w, 8), dtype=dt, device=dev)
        aux_f = torch.zeros_like(aux_b)
        if n > 1:
            aux_b[..., :2] = flows_f
            aux_b[..., 3:5] = mask2[:n - 1]
            hip.fb_check(aux_b, flows_b.contiguous(), out=aux_b, out_choff=2)
            aux_f[..., :2] = flows_b
            aux_f[..., 3:5] = mask2[1:]
            hip.fb_check(aux_f, flows_f.contiguous(), out=aux_f, out_choff=2)
        return aux_b, aux_f, mk8

    def feature_propagation(self, x, flows_f, flows_b, mask2, interpolation="bilinear", rows=None, out=None):
        """x [t,h,w,128]; flows_* [t-1,h,w,2] NHWC (1/4-res, already /4); mask2 [t,h,w,2] -> fused [t,h,w,128] (written to `out`
        when given).  rows = propagation_rows(...) of these frames when the caller has them already (per-clip cache).

        BATCHED form (engine extension, propagate_windows): x [t,B,h,w,128] and rows of shapes [t-1,B,h,w,8] / [t,B,h,w,8] hold B
        independent windows of the same length, step-major -- step i of every window is ONE launch per layer over B frames.  A window's
        recurrence only ever reads its own frames (batch items never mix in a convolution, a warp or a deformable sampling), so every
        window's result is bit-identical to its own un-batched pass."""
        batched = x.dim() == 5
        aux_b, aux_f, mk8 = rows if rows is not None else self.propagation_rows(flows_f, flows_b, mask2)
        if not batched:
            x, aux_b, aux_f, mk8 = x.unsqueeze(1), aux_b.unsqueeze(1), aux_f.unsqueeze(1), mk8.unsqueeze(1)
        t, B, h, w, c = x.shape
        dev, dt = x.device, self.dtype
        feats = {"input": x}
        prev_name = "input"
        for name in ("backward_1", "forward_1"):
            L = self.prop[name]
            # step i >= 1 of the backward pass moves from frame idx + 1 to idx = t - 1 - i with the rows of pair idx; the forward pass
            # moves from frame i - 1 to i with the rows of pair i - 1 (plain slices: the whole window is capturable in a hipGraph)
            if name == "backward_1":
                order = list(range(t - 1, -1, -1))
                aux, pair = aux_b, (lambda i, idx: idx)
            else:
                order = list(range(t))
                aux, pair = aux_f, (lambda i, idx: i - 1)
            cur_all = feats[prev_name]
            outs = torch.empty((t, B, h, w, c), dtype=dt, device=dev)
            prop = None
            for i, idx in enumerate(order):
                cur = cur_all[idx]
                if i == 0:
                    prop = cur
                else:
                    ax = aux[pair(i, idx)]
                    warped = hip.flow_warp(prop, ax, mode=interpolation)
                    o = L["off0"]([cur, warped, ax], act="lrelu", act_param=0.1)
                    o = L["off2"]([o], act="lrelu", act_param=0.1)
                    o = L["off4"]([o], act="lrelu", act_param=0.1)
                    om = L["off6"]([o], fuse=dict(kind="dcn_om", mag=3.0, flow=ax))      # 3 * tanh(offsets) + flow | sigmoid(masks)
                    prop = L["dcn"]([prop], dcn_offmask=om)
                y = L["bb0"]([cur, prop, mk8[idx]], act="lrelu", act_param=0.2)
                L["bb2"]([y], out=outs[idx], residual=prop)
                prop = outs[idx]
            feats[name] = outs
            prev_name = name
        if out is None:
            out = torch.empty((t, B, h, w, c), dtype=dt, device=dev)
        out5 = out.view(t, B, h, w, c)
        for j in range(t) if batched else (slice(None),):       # (batched: one step-slab at a time keeps every operand below 2 GiB)
            sl = (lambda v: v[j]) if batched else (lambda v: v.view(t * B, h, w, v.shape[-1]))
            y = self.fuse0([sl(feats["backward_1"]), sl(feats["forward_1"]), sl(mk8)], act="lrelu", act_param=0.2)
            self.fuse2([y], residual=sl(x), out=sl(out5))
        return out5 if batched else out5.view(t, h, w, c)

    def propagate_windows(self, clip, windows):
        """PLANS the feature propagation of all generator windows of a clip, windows of equal length batched (engine extension; the
        reference runs it inside every window's forward, model/propainter.py:345-349).  windows: [(first local frame, l_t), ...].
        A window's propagation is a chain of ~170 launches over ONE 1/4-resolution frame each (57 600 pixels at 720p: ~450 blocks for
        512 slots -- a single generation of blocks, all prologue and epilogue); the windows are independent, so the 14 full-length
        windows of an 80-frame clip run as ONE chain of launches over 14 frames each.  Bit-identical to the per-window chain (see
        feature_propagation); groups whose first frames are no arithmetic progression stay on the per-window path.

        ROLLING (round 5, ADVICE medium): a group of windows is propagated when its first window is about to run (``ensure_propagated``,
        called by the pass on its main stream before the window is handed to a lane; ``forward_window`` does it itself for callers that
        do not) and its fused tensor is dropped when its last window has been consumed (``release_window``) -- the pass holds ONE
        group's results (<= ~2.5 GB + the chain's temporaries) instead of every window's: round 4 kept ~2.2x the encoder cache alive
        until the end of the pass (35 GB or more for a 500-frame 1080p clip)."""
        enc = clip["enc"]
        h, w = enc.shape[1], enc.shape[2]
        by_len = {}
        for first, l_t in windows:
            by_len.setdefault(l_t, []).append(first)
        clip["prop"], clip["prop_plan"], clip["prop_groups"], clip["prop_left"] = {}, {}, [], {}
        for l_t, firsts in by_len.items():
            firsts = sorted(set(firsts))
            step = firsts[1] - firsts[0] if len(firsts) > 1 else 1
            if l_t < 2:
                continue
            # batches of at most `bmax` windows: the step-major gather of a batch stays below ~2.5 GB (five such tensors are alive during
            # the chain; launches over >= 8 frames already run at the large-M rate, profiles/r4_batched_propagation.txt)
            per_window = l_t * h * w * 128 * enc.element_size()
            bmax = max(2, min(len(firsts), int(self.prop_batch_bytes // per_window)))
            if any(b - a != step for a, b in zip(firsts, firsts[1:])):
                bmax = 1                      # no arithmetic progression of first frames: every window a group of its own
            for b0 in range(0, len(firsts), bmax):
                group = firsts[b0:b0 + bmax]
                # (round 6: a single window -- the clip's first / last windows, whose lengths differ from the rest -- is a group of ONE and
                #  goes through the same path: its recurrent chain of ~100 small launches then runs on the pass's main stream in
                #  ensure_propagated, never inside a window LANE.  Captured into a hipGraph, lanes carrying those chains gave ~5 % of the replays
                #  a few hundred wrong bytes in one frame of exactly these windows -- profiles/r6_replay_bytes.txt; same arithmetic either way.)
                if len(group) < 2 and os.environ.get("PP_CHAIN_IN_LANES") == "1":
                    continue                  # [diagnosis] rounds 2-5: a single window propagates inside its lane (the defect of profiles/r6_replay_bytes.txt)
                gid = len(clip["prop_groups"])
                clip["prop_groups"].append((l_t, group, step))
                clip["prop_left"][gid] = len(group)
                for first in group:
                    clip["prop_plan"][(first, l_t)] = gid
        return clip

    def ensure_propagated(self, clip, first, l_t):
        """Runs the batched propagation of the group the window (first, l_t) belongs to, on the CURRENT stream, unless it is there already."""
        gid = clip.get("prop_plan", {}).get((first, l_t))
        if gid is None or (first, l_t) in clip["prop"] or clip["prop_left"].get(gid, 0) <= 0:
            return
        enc = clip["enc"]
        dev = enc.device
        h, w = enc.shape[1], enc.shape[2]
        l_t, group, step = clip["prop_groups"][gid]
        B = len(group)
        # frame index of (step j, window k) = group[0] + step * k + j -- built on the device (capturable: no host copy)
        base = torch.arange(B, device=dev) * step + group[0]
        idx = (torch.arange(l_t, device=dev)[:, None] + base[None, :]).reshape(-1)
        idxp = (torch.arange(l_t - 1, device=dev)[:, None] + base[None, :]).reshape(-1)
        g5 = lambda src, ix, n: src.index_select(0, ix).view(n, B, h, w, src.shape[-1])
        fused = self.feature_propagation(g5(enc, idx, l_t), None, None, None, clip["interpolation"],
                                         rows=(g5(clip["aux_b"], idxp, l_t - 1), g5(clip["aux_f"], idxp, l_t - 1), g5(clip["mk8"], idx, l_t)))
        for k, f in enumerate(group):
            clip["prop"][(f, l_t)] = (fused, k)

    def release_window(self, clip, first, l_t):
        """The window (first, l_t) has read its propagated frames: when it was the last one of its group, the group's tensor goes."""
        gid = clip.get("prop_plan", {}).get((first, l_t))
        if gid is None or (first, l_t) not in clip.get("prop", {}):
            return
        clip["prop_left"][gid] -= 1
        if clip["prop_left"][gid] <= 0:
            for f in clip["prop_groups"][gid][1]:
                clip["prop"].pop((f, l_t), None)

    # ------------------------------------------------------------------ transformer
    def _window_tables(self, Hp, Wp):
        key = (Hp, Wp)
        if key not in self._win_cache:
            own, rolled = hip.window_tables(Hp, Wp, *WIN)
            self._win_cache[key] = (torch.from_numpy(own).to(self.device), torch.from_numpy(rolled).to(self.device))
        return self._win_cache[key]

    def transformer(self, tok, size, token_mask, t_dilation=2):
        """TemporalSparseTransformerBlock (:328-344); tok [t,fh,fw,512]; token_mask [l_t,fh,fw] -> [t,fh,fw,512]."""
        t, fh, fw, c = tok.shape
        h, w = size
        dev, dt = tok.device, self.dtype
        Hp, Wp = math.ceil(fh / WIN[0]) * WIN[0], math.ceil(fw / WIN[1]) * WIN[1]
        padded = (Hp, Wp) != (fh, fw)
        own, rolled = self._window_tables(Hp, Wp)
        mpad = torch.zeros((1, token_mask.shape[0], Hp, Wp), dtype=dt, device=dev)
        mpad[0, :, :fh, :fw] = token_mask
        wmask = hip.window_mask(mpad, *WIN)
        tkey = (t, t_dilation, str(dev))
        if tkey not in self._tind_cache:
            self._tind_cache[tkey] = [torch.arange(i, t, t_dilation, dtype=torch.int32, device=dev) for i in range(t_dilation)]
        tinds = self._tind_cache[tkey]
        ypad = torch.zeros((t, Hp, Wp, c), dtype=dt, device=dev) if padded else None
        x = tok
        n_tok = t * fh * fw
        for i, B in enumerate(self.blocks):
            if padded:       # zero pad AFTER LayerNorm (:169-171; pad tokens' q/k/v = bias): LayerNorm writes into the padded grid, whose
                y = hip.layernorm_grid(x, *B["n1"], out=ypad)     # padding tokens were zero-filled once and are never written
            else:
                y = hip.layernorm(x, *B["n1"])
            qkv = B["qkv"]([y.view(1, 1, t * Hp * Wp, c)]).view(1, t, Hp, Wp, 3 * c)
            pooled = hip.depthwise_pool(y, B["pool_w"], B["pool_b"], 4)
            P = pooled.shape[1] * pooled.shape[2]
            pkv = B["kv"]([pooled.view(1, 1, t * P, c)]).view(1, t, P, 2 * c)
            att = hip.sparse_window_attention(
                qkv, qkv[..., c:], qkv[..., 2 * c:], pkv, pkv[..., c:], own, rolled, tinds[i % t_dilation], wmask,
                heads=HEADS, wh=WIN[0], ww=WIN[1], qkv_cstride=3 * c, pkv_cstride=2 * c, C_=c,
                out_hw=(fh, fw) if padded else None)      # the crop of :276-277 happens in the kernel's store
            att = att[0]
            x = B["proj"]([att.view(1, 1, n_tok, c)], residual=x.view(1, 1, n_tok, c)).view(t, fh, fw, c)
            y = hip.layernorm(x, *B["n2"])
            hid = B["fc1"]([y.view(1, 1, n_tok, c)])                      # [1,1,n_tok,1960]
            folded = hip.fold_tokens(hid.view(t, fh * fw, 1960), t, fh, fw, 40, h, w, normalize=True, act=hip.ACT_GELU)
            x = B["fc2"]([folded], residual=x)
        return x

    # ------------------------------------------------------------------ whole forward (:319-372)
    def encode_frames(self, masked_frames, masks_in, masks_updated, chunk=16):
        """Encoder features of every frame of [1,t,3,H,W] / [1,t,1,H,W] inputs -> NHWC [t,H/4,W/4,128].
        The encoder is per-frame (propainter.py:334-336: cat(frame, mask_in, mask_updated) -> Encoder), so the clip
        driver computes it once per frame and hands each window its slice instead of re-encoding every frame in every
        window (17.5 % of the reference's FLOPs); results are identical."""
        b, t, _, H, W = masked_frames.shape
        assert b == 1, "the inference path runs one clip per call (inference_propainter.py always has b == 1)"
        dt, dev = self.dtype, masked_frames.device
        outs = []
        for s in range(0, t, chunk):
            e = min(t, s + chunk)
            x = hip.pack_nhwc8([v[0, s:e].to(dt).contiguous() for v in (masked_frames, masks_in, masks_updated)])      # cat(frame, mask, updated mask) (:334-336)
            outs.append(self.encode(x))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    def forward(self, masked_frames, flows_bi, masks_in, masks_updated, l_t, interpolation, t_dilation, enc_feat=None):
        b, t, _, H, W = masked_frames.shape
        assert b == 1, "the inference path runs one clip per call (inference_propainter.py always has b == 1)"
        dt, dev = self.dtype, masked_frames.device
        if enc_feat is None:
            enc = self.encode_frames(masked_frames, masks_in, masks_updated)   # [t,h,w,128]
        else:
            enc = enc_feat
            assert enc.shape[0] == t and enc.shape[-1] == 128 and enc.dtype == dt and enc.is_contiguous(), enc.shape
        h, w = enc.shape[1], enc.shape[2]
        # 1/4-res flows (bilinear, align_corners=False, /4) and masks (nearest) — tiny host-side glue (:338-342)
        dsf = F.interpolate(flows_bi[0][0], scale_factor=1 / 4, mode="bilinear", align_corners=False) / 4.0
        dsb = F.interpolate(flows_bi[1][0], scale_factor=1 / 4, mode="bilinear", align_corners=False) / 4.0
        dsf = dsf.permute(0, 2, 3, 1).contiguous()
        dsb = dsb.permute(0, 2, 3, 1).contiguous()
        dm_in = masks_in[0, :l_t, :, ::4, ::4]
        dm_up = masks_updated[0, :l_t, :, ::4, ::4]
        mask2 = torch.cat([dm_in, dm_up], 1).permute(0, 2, 3, 1).contiguous()          # [l_t,h,w,2]
        token_mask = F.max_pool2d(dm_in, 7, 3, 3)[:, 0]                               # [l_t,fh,fw]
        if interpolation not in ("bilinear", "nearest"):
            raise ValueError(f"interpolation {interpolation!r}: 'bilinear' or 'nearest' (model/propainter.py:148)")
        encw = torch.empty((t, h, w, 128), dtype=dt, device=dev)          # [propagated local frames | reference frames]
        self.feature_propagation(enc[:l_t], dsf, dsb, mask2, interpolation, out=encw[:l_t])
        if t > l_t:
            encw[l_t:] = enc[l_t:]
        return self._window_tail(encw, l_t, token_mask, t_dilation, H, W)

    def _window_tail(self, encw, l_t, token_mask, t_dilation, H, W):
        """SoftSplit -> transformer -> SoftComp -> decoder of one window (:351-372); encw [t,h,w,128]."""
        t, h, w, _ = encw.shape
        tok = self.ss([encw])                                              # SoftSplit as one convolution
        tok = self.transformer(tok, (h, w), token_mask, t_dilation)
        fh, fw = tok.shape[1], tok.shape[2]
        emb = self.sc_embed([tok.view(1, 1, t * fh * fw, HIDDEN)])          # [1,1,n,6272]
        folded = hip.fold_tokens(emb.view(t, fh * fw, 128 * 49), t, fh, fw, 128, h, w, normalize=False)
        enc2 = self.sc_bias([folded], residual=encw)                       # bias_conv + (enc_feat + trans_feat)
        out = self.decode(enc2[:l_t])
        return out.view(1, l_t, 3, H, W)

    # ------------------------------------------------------------------ per-clip cache (engine extension used by pipeline.run_clip)
    def prepare_clip(self, frames, flows_bi, masks_in, masks_updated, interpolation="bilinear"):
        """Everything the generator windows of ONE clip need that depends on a frame or a flow pair only -- encoder features, 1/4
        resolution flows and masks, token masks, the per-pair rows of the propagation steps -- computed once per clip instead of once
        per window (a frame sits in 2-3 windows as a local frame and in several more as a reference).  frames [1,L,3,H,W], flows
        2 x [1,L-1,2,H,W], masks [1,L,1,H,W].  Same arithmetic on the same values as forward(): results are identical."""
        b, L, _, H, W = frames.shape
        assert b == 1
        dt = self.dtype
        enc = self.encode_frames(frames, masks_in, masks_updated)
        dsf = (F.interpolate(flows_bi[0][0], scale_factor=1 / 4, mode="bilinear", align_corners=False) / 4.0).permute(0, 2, 3, 1).contiguous()
        dsb = (F.interpolate(flows_bi[1][0], scale_factor=1 / 4, mode="bilinear", align_corners=False) / 4.0).permute(0, 2, 3, 1).contiguous()
        dm_in = masks_in[0, :, :, ::4, ::4]
        mask2 = torch.cat([dm_in, masks_updated[0, :, :, ::4, ::4]], 1).permute(0, 2, 3, 1).contiguous()        # [L,h,w,2]
        aux_b, aux_f, mk8 = self.propagation_rows(dsf, dsb, mask2)
        return dict(enc=enc, aux_b=aux_b, aux_f=aux_f, mk8=mk8, token_mask=F.max_pool2d(dm_in, 7, 3, 3)[:, 0].contiguous(),
                    H=H, W=W, L=L, interpolation=interpolation)

    def forward_window(self, clip, first, l_t, ref_index, t_dilation=2):
        """The generator call of the window whose local frames are clip frames [first, first + l_t) and whose reference frames are
        ref_index (int64 device tensor, may be empty) -> [1,l_t,3,H,W]; == forward() on the gathered inputs."""
        enc = clip["enc"]
        h, w = enc.shape[1], enc.shape[2]
        n_ref = int(ref_index.numel())
        encw = torch.empty((l_t + n_ref, h, w, 128), dtype=self.dtype, device=enc.device)
        a, b = first, first + l_t
        self.ensure_propagated(clip, first, l_t)      # (no-op when the pass did it on its main stream, or when the window is not batched)
        done = clip.get("prop", {}).get((first, l_t))
        if done is not None:          # propagated with the other windows of its group (propagate_windows)
            encw[:l_t].copy_(done[0][:, done[1]])
        else:
            self.feature_propagation(enc[a:b], None, None, None, clip["interpolation"],
                                     rows=(clip["aux_b"][a:b - 1], clip["aux_f"][a:b - 1], clip["mk8"][a:b]), out=encw[:l_t])
        if n_ref:
            torch.index_select(enc, 0, ref_index, out=encw[l_t:])
        return self._window_tail(encw, l_t, clip["token_mask"][a:b], t_dilation, clip["H"], clip["W"])

    # ------------------------------------------------------------------ image propagation (:104-190, non-learnable)
    def img_propagation(self, frames, flows_f, flows_b, masks, interpolation):
        b, t, c, H, W = frames.shape
        assert b == 1
        fr, mk = frames[0].contiguous(), masks[0].contiguous()
        ff, fb = flows_f[0].contiguous(), flows_b[0].contiguous()
        cur_x, cur_m = fr, mk
        for name in ("backward", "forward"):
            out_x, out_m = torch.empty_like(fr), torch.empty_like(mk)
            if name == "backward":
                order = list(range(t - 1, -1, -1))
                fidx = order
                f_prop, f_chk = ff, fb
            else:
                order = list(range(t))
                fidx = [None] + list(range(0, t - 1))
                f_prop, f_chk = fb, ff
            for i, idx in enumerate(order):
                if i == 0:
                    out_x[idx].copy_(cur_x[idx])
                    out_m[idx].copy_(cur_m[idx])
                else:
                    pidx = order[i - 1]
                    hip.img_prop_step(out_x[pidx:pidx + 1], out_m[pidx:pidx + 1], cur_x[idx:idx + 1], cur_m[idx:idx + 1],
                                      f_prop[fidx[i]:fidx[i] + 1], f_chk[fidx[i]:fidx[i] + 1], out_x[idx:idx + 1],
                                      out_m[idx:idx + 1], mode=interpolation)
            cur_x, cur_m = out_x, out_m
        return cur_x.view(1, t, c, H, W), cur_m.view(1, t, 1, H, W)


class InpaintGenerator(nn.Module):
    def __init__(self, init_weights=True, model_path=None):
        super().__init__()
        tree = ParamTree()
        # init_weights=True -> N(0, 0.02) weights, zero biases (reference BaseNetwork.init_weights); pool_layer 1/16
        populate(tree, generator_schema(), std=0.02 if init_weights else None)
        for name, child in tree._modules.items():
            self.add_module(name, child)
        for i in range(DEPTH):
            att = self.transformers.transformer._modules[str(i)].attention
            if not init_weights:
                att.pool_layer.weight.data.fill_(1.0 / 16)
            att.register_buffer("valid_ind_rolled", self._valid_ind_rolled())
        for m in ("backward_1", "forward_1"):      # DeformableAlignment.init_offset (:53-54)
            d = self.feat_prop_module.deform_align._modules[m].conv_offset._modules["6"]
            if not init_weights:
                d.weight.data.zero_()
        if model_path is not None:
            print('Pretrained ProPainter has loaded...')
            ckpt = torch.load(model_path, map_location='cpu')
            self.load_state_dict(ckpt, strict=True)
        self._engine = None

    @staticmethod
    def _valid_ind_rolled():
        """Indices kept from the four rolled windows (sparse_transformer.py:142-153) — kept as a buffer only for
        state-dict compatibility; the engine derives the same set in pp_window_tables."""
        wh, ww = WIN
        eh, ew = (wh + 1) // 2, (ww + 1) // 2
        keep = []
        for k in range(4):
            for i in range(wh):
                for j in range(ww):
                    zr = (i < wh - eh) if k < 2 else (i >= eh)
                    zc = (j < ww - ew) if k % 2 == 0 else (j >= ew)
                    if not (zr and zc):
                        keep.append(k * wh * ww + i * ww + j)
        return torch.tensor(keep, dtype=torch.long)

    def _get_engine(self, dtype, device):
        key = (dtype, str(device), sum(p._version for p in self.parameters()))
        if self._engine is None or self._engine[0] != key:
            sd = {k: v.detach().float().cpu() for k, v in self.state_dict().items() if v.is_floating_point()}
            self._engine = (key, _GenEngine(sd, dtype, device))
        return self._engine[1]

    @hip.on_input_device
    @torch.no_grad()
    def img_propagation(self, masked_frames, completed_flows, masks, interpolation='nearest'):
        hip.require_gpu(masked_frames, "InpaintGenerator")
        eng = self._get_engine(masked_frames.dtype, masked_frames.device)
        dt = masked_frames.dtype
        return eng.img_propagation(masked_frames, completed_flows[0].to(dt), completed_flows[1].to(dt), masks.to(dt),
                                   interpolation)

    @hip.on_input_device
    @torch.no_grad()
    def encode_frames(self, masked_frames, masks_in, masks_updated):
        """Engine extension (not in the reference API): per-frame encoder features, NHWC [t,H/4,W/4,128], to be passed
        to ``forward(..., enc_feat=feat[ids])`` so that a clip driver encodes every frame once."""
        hip.require_gpu(masked_frames, "InpaintGenerator")
        dt = masked_frames.dtype
        eng = self._get_engine(dt, masked_frames.device)
        return eng.encode_frames(masked_frames, masks_in.to(dt), masks_updated.to(dt))

    @hip.on_input_device
    @torch.no_grad()
    def prepare_clip(self, frames, completed_flows, masks_in, masks_updated, interpolation='bilinear'):
        """Engine extension (not in the reference API): the per-clip cache of ``forward_window`` -- encoder features, 1/4-resolution
        flows / masks, token masks and the propagation side inputs of EVERY frame / flow pair of a clip, computed once.
        frames [1,L,3,H,W] (the updated frames), completed_flows 2 x [1,L-1,2,H,W], masks [1,L,1,H,W]."""
        hip.require_gpu(frames, "InpaintGenerator")
        if interpolation not in ("bilinear", "nearest"):
            raise ValueError(f"interpolation {interpolation!r}: 'bilinear' or 'nearest' (model/propainter.py:148)")
        dt = frames.dtype
        if frames.shape[0] != 1 or frames.shape[3] % 8 or frames.shape[4] % 8:
            raise ValueError(f"prepare_clip takes one clip [1,L,3,H,W] with H, W multiples of 8 (got {tuple(frames.shape)})")
        eng = self._get_engine(dt, frames.device)
        return eng.prepare_clip(frames, (completed_flows[0].to(dt), completed_flows[1].to(dt)), masks_in.to(dt), masks_updated.to(dt),
                                interpolation)

    @torch.no_grad()
    def propagate_windows(self, clip, windows):
        """Engine extension: the feature propagation (model/propainter.py:345-349) of all windows ``[(first, num_local_frames), ...]`` of a
        prepared clip up front, windows of equal length as one batch; ``forward_window`` then reads its frames from ``clip['prop']``.
        Identical results (a window's recurrence only reads its own frames)."""
        enc = clip["enc"]
        import contextlib
        with (torch.cuda.device(enc.device) if enc.is_cuda else contextlib.nullcontext()):
            return self._get_engine(enc.dtype, enc.device).propagate_windows(clip, [(int(f), int(n)) for f, n in windows])

    @torch.no_grad()
    def ensure_propagated(self, clip, first, num_local_frames):
        """Engine extension: see _GenEngine.ensure_propagated (call on the stream the clip cache was prepared on)."""
        enc = clip["enc"]
        import contextlib
        with (torch.cuda.device(enc.device) if enc.is_cuda else contextlib.nullcontext()):
            self._get_engine(enc.dtype, enc.device).ensure_propagated(clip, int(first), int(num_local_frames))

    def release_window(self, clip, first, num_local_frames):
        enc = clip["enc"]
        self._get_engine(enc.dtype, enc.device).release_window(clip, int(first), int(num_local_frames))

    @hip.on_input_device
    @torch.no_grad()
    def forward_window(self, clip, first, num_local_frames, ref_index, t_dilation=2):
        """``forward`` of the window with local frames [first, first + num_local_frames) and reference frames ``ref_index`` (int64 device
        tensor) of a clip prepared by ``prepare_clip``; returns [1,l_t,3,H,W] -- identical to ``forward`` on the gathered tensors."""
        assert DEPTH % t_dilation == 0, 'wrong t_dilation input.'
        enc = clip["enc"]
        return self._get_engine(enc.dtype, enc.device).forward_window(clip, int(first), int(num_local_frames), ref_index, t_dilation)

    @hip.on_input_device
    @torch.no_grad()
    def forward(self, masked_frames, completed_flows, masks_in, masks_updated, num_local_frames,
                interpolation='bilinear', t_dilation=2, enc_feat=None):
        hip.require_gpu(masked_frames, "InpaintGenerator")
        if self.training:
            raise NotImplementedError("training is outside the inference hot path; call .eval()")
        assert DEPTH % t_dilation == 0, 'wrong t_dilation input.'
        dt = masked_frames.dtype
        b, t, _, H, W = masked_frames.shape
        if H % 8 or W % 8:
            # the 1/4-resolution masks are taken by striding (== F.interpolate(nearest, 1/4) only for multiples of 4) and the
            # driver always resizes to multiples of 8 (inference_propainter.py:34-47 via resize_frames)
            raise ValueError(f"InpaintGenerator.forward needs H, W multiples of 8 (got {H}x{W})")
        eng = self._get_engine(dt, masked_frames.device)
        cf = (completed_flows[0].to(dt), completed_flows[1].to(dt))
        mi, mu = masks_in.to(dt), masks_updated.to(dt)
        if b == 1:
            return eng.forward(masked_frames, cf, mi, mu, num_local_frames, interpolation, t_dilation, enc_feat=enc_feat)
        assert enc_feat is None, "enc_feat (a per-clip engine extension) goes with b == 1"
        outs = [eng.forward(masked_frames[i:i + 1], (cf[0][i:i + 1], cf[1][i:i + 1]), mi[i:i + 1], mu[i:i + 1], num_local_frames,
                            interpolation, t_dilation) for i in range(b)]     # samples are independent (reference :319-372)
        return torch.cat(outs, 0)
