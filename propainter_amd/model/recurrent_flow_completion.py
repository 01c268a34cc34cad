"""Recurrent flow completion on the MI355X engine, behind the reference's ``RecurrentFlowCompleteNet`` interface.

Drop-in for ``model/recurrent_flow_completion.py:203-347`` (reference): constructor ``(model_path=None)``,
``forward(masked_flows[b,t,2,h,w], masks[b,t,1,h,w]) -> (flow[b,t,2,h,w], None)``, ``forward_bidirect_flow`` and
``combine_flow``; identical state-dict keys (the training-only ``edgeDetector`` parameters are kept so that
``load_state_dict(strict=True)`` accepts the released checkpoint, but the edge branch — ``self.training`` only,
:301-305 — is not built).

Engine layout: frames are NHWC ``[b*t, h, w, C]``; Conv3d (1,k,k) layers are per-frame 2-D implicit GEMMs; the P3D
temporal (3,1,1) dilation-2 convs are implicit GEMMs over the ``[t, h*w]`` view with taps (t-2, t, t+2); the
second-order deformable propagation is strictly sequential in t (2*t dependent steps) with the concatenations
expressed as multi-source convolutions instead of copies.
"""
import torch
import torch.nn as nn

from .. import hip
from ..conv import ConvLayer
from ..param_tree import ParamTree, conv_entries, populate


def _conv3d_entries(prefix, cout, cin, kt, kh, kw):
    return [(f"{prefix}.weight", (cout, cin, kt, kh, kw), "w"), (f"{prefix}.bias", (cout,), "b")]


def fc_schema():
    e = _conv3d_entries("downsample.0", 32, 3, 1, 5, 5)
    for name, specs in (("encoder1", ((0, 32, 32), (2, 32, 64))), ("encoder2", ((0, 64, 64), (2, 64, 128)))):
        for idx, cin, cout in specs:
            e += _conv3d_entries(f"{name}.{idx}.conv1.0", cout, cin, 1, 3, 3)
            e += _conv3d_entries(f"{name}.{idx}.conv2.0", cout, cout, 3, 1, 1)
    for i in (0, 2, 4):
        e += _conv3d_entries(f"mid_dilation.{i}", 128, 128, 1, 3, 3)
    fp = "feat_prop_module"
    for i, m in enumerate(("backward_", "forward_")):
        d = f"{fp}.deform_align.{m}"
        e += [(f"{d}.weight", (128, 256, 3, 3), "w"), (f"{d}.bias", (128,), "b")]
        e += conv_entries(f"{d}.conv_offset.0", 128, 384, 3)
        e += conv_entries(f"{d}.conv_offset.2", 128, 128, 3) + conv_entries(f"{d}.conv_offset.4", 128, 128, 3)
        e += conv_entries(f"{d}.conv_offset.6", 432, 128, 3)
        e += conv_entries(f"{fp}.backbone.{m}.0", 128, (2 + i) * 128, 3) + conv_entries(f"{fp}.backbone.{m}.2", 128, 128, 3)
    e += conv_entries(f"{fp}.fusion", 128, 256, 1)
    e += conv_entries("decoder2.0", 128, 128, 3) + conv_entries("decoder2.2.conv", 64, 128, 3)
    e += conv_entries("decoder1.0", 64, 64, 3) + conv_entries("decoder1.2.conv", 32, 64, 3)
    e += conv_entries("upsample.0", 32, 32, 3) + conv_entries("upsample.2.conv", 2, 32, 3)
    e += conv_entries("edgeDetector.projection.0", 16, 2, 3) + conv_entries("edgeDetector.mid_layer_1.0", 16, 16, 3)
    e += conv_entries("edgeDetector.mid_layer_2.0", 16, 16, 3) + conv_entries("edgeDetector.out_layer", 1, 16, 1)
    return e


class _FCEngine:
    def __init__(self, sd, dtype, device):
        self.dtype = dtype
        mk = lambda w, b, **kw: ConvLayer(w, b, dtype=dtype, device=device, **kw)
        g = lambda n: (sd[n + ".weight"], sd[n + ".bias"])
        sp = lambda n, **kw: mk(sd[n + ".weight"][:, :, 0], sd[n + ".bias"], **kw)                 # (1,k,k) conv3d
        tp = lambda n: mk(sd[n + ".weight"][:, :, :, :, 0], sd[n + ".bias"], padding=(2, 0), dilation=(2, 1))  # (3,1,1)
        self.down = sp("downsample.0", stride=2, padding=2, pad_mode="replicate", src_channels=[3])
        self.p3d = []
        for name, stride in (("encoder1.0", 1), ("encoder1.2", 2), ("encoder2.0", 1), ("encoder2.2", 2)):
            self.p3d.append((sp(name + ".conv1.0", stride=stride, padding=1), tp(name + ".conv2.0")))
        self.mid = [sp(f"mid_dilation.{i}", padding=d, dilation=d) for i, d in ((0, 3), (2, 2), (4, 1))]
        fp = "feat_prop_module."
        self.prop = {}
        for i, m in enumerate(("backward_", "forward_")):
            d = fp + f"deform_align.{m}"
            self.prop[m] = dict(
                off0=mk(*g(d + ".conv_offset.0"), padding=1, src_channels=[128, 128, 128]),
                off2=mk(*g(d + ".conv_offset.2"), padding=1),
                off4=mk(*g(d + ".conv_offset.4"), padding=1),
                off6=mk(*g(d + ".conv_offset.6"), padding=1),
                dcn=mk(sd[d + ".weight"], sd[d + ".bias"], padding=1, src_channels=[128, 128], dcn_groups=16),
                bb0=mk(*g(fp + f"backbone.{m}.0"), padding=1, src_channels=[128] * (2 + i)),
                bb2=mk(*g(fp + f"backbone.{m}.2"), padding=1))
        self.fusion = mk(*g(fp + "fusion"), src_channels=[128, 128])
        self.dec2_0 = mk(*g("decoder2.0"), padding=1)
        self.dec2_2 = mk(*g("decoder2.2.conv"), padding=1)
        self.dec1_0 = mk(*g("decoder1.0"), padding=1)
        self.dec1_2 = mk(*g("decoder1.2.conv"), padding=1)
        self.up_0 = mk(*g("upsample.0"), padding=1)
        self.up_2 = mk(*g("upsample.2.conv"), padding=1)

    def temporal(self, layer, x, b, t, act, act_param):
        """P3D temporal conv: x [b*t,h,w,C] viewed as an image [b, t, h*w, C]."""
        bt, h, w, c = x.shape
        y = layer([x.view(b, t, h * w, c)], act=act, act_param=act_param)
        return y.view(bt, h, w, -1)

    def propagate(self, x, b, t):
        """BidirectionalPropagation.forward (:66-124); x [t, b, h, w, 128] (time-major) -> same."""
        _, _, h, w, c = x.shape
        dev, dt = x.device, self.dtype
        zeros = torch.zeros((b, h, w, c), dtype=dt, device=dev)
        feats = {}
        for name in ("backward_", "forward_"):
            L = self.prop[name]
            order = range(t - 1, -1, -1) if name == "backward_" else range(t)
            outs = torch.empty((t, b, h, w, c), dtype=dt, device=dev)
            prev, prev2 = None, None            # feat_prop of steps i-1 and i-2
            for i, idx in enumerate(order):
                cur = x[idx]
                if i == 0:
                    prop = zeros
                else:
                    n2 = prev2 if prev2 is not None else zeros
                    o = L["off0"]([prev, cur, n2], act="lrelu", act_param=0.1)
                    o = L["off2"]([o], act="lrelu", act_param=0.1)
                    o = L["off4"]([o], act="lrelu", act_param=0.1)
                    om = L["off6"]([o], fuse=dict(kind="dcn_om", mag=5.0))               # 5 * tanh(offsets) | sigmoid(masks)
                    prop = L["dcn"]([prev, n2], dcn_offmask=om)
                srcs = [cur] + ([feats["backward_"][idx]] if name == "forward_" else []) + [prop]
                y = L["bb0"](srcs, act="lrelu", act_param=0.1)
                L["bb2"]([y], out=outs[idx], residual=prop)
                prev2, prev = prev, outs[idx]
            feats[name] = outs
        tb = t * b
        fused = self.fusion([feats["backward_"].view(tb, h, w, c), feats["forward_"].view(tb, h, w, c)],
                            residual=x.view(tb, h, w, c))
        return fused.view(t, b, h, w, c)

    def _encode(self, masked_flows, masks):
        """downsample + P3D encoders + dilated mid convs of ONE sequence [t,2,H,W] / [t,1,H,W] (:272-286)."""
        t, _, H, W = masked_flows.shape
        if masked_flows.dtype == self.dtype and masks.dtype == self.dtype:     # one launch, whole 16-byte rows
            x = hip.pack_nhwc8([masked_flows.contiguous(), masks.contiguous()])
        else:
            x = torch.zeros((t, H, W, 8), dtype=self.dtype, device=masked_flows.device)
            hip.nchw_to_nhwc(masked_flows.contiguous(), out=x, out_choff=0)
            hip.nchw_to_nhwc(masks.contiguous(), out=x, out_choff=2)
        x = self.down([x], act="lrelu", act_param=0.2)
        feats = []
        for si, (spatial, temporal) in enumerate(self.p3d):
            x = spatial([x], act="lrelu", act_param=0.2)
            x = self.temporal(temporal, x, 1, t, "lrelu", 0.2)      # P3DBlock conv2, then the Sequential's LeakyReLU
            feats.append(x)
        for m in self.mid:
            x = m([x], act="lrelu", act_param=0.2)
        return x, feats[1]

    def _decode(self, p, e1):
        """decoder2 / decoder1 / upsample (:288-300) of one sequence; p [t,h8,w8,128] -> planar [t,2,H,W]."""
        y = self.dec2_0([p], act="lrelu", act_param=0.2)
        y = self.dec2_2([hip.upsample2x(y)], act="lrelu", act_param=0.2, residual=e1)
        y = self.dec1_0([y], act="lrelu", act_param=0.2)
        y = self.dec1_2([hip.upsample2x(y)], act="lrelu", act_param=0.2)
        y = self.up_0([y], act="lrelu", act_param=0.2)
        y = self.up_2([hip.upsample2x(y)])
        return hip.nhwc_to_nchw(y, 2)

    def forward(self, masked_flows, masks):
        """[b,t,2,H,W], [b,t,1,H,W] -> [b,t,2,H,W].  The feed-forward encoders / decoders run per sequence (their
        activations are large); the latency-bound recurrent propagation advances all b sequences per step."""
        b, t, _, H, W = masked_flows.shape
        enc = [self._encode(masked_flows[i], masks[i]) for i in range(b)]
        xt = torch.stack([e[0] for e in enc], 1)                                   # [t,b,h8,w8,128] time-major
        p = self.propagate(xt, b, t)
        return torch.stack([self._decode(p[:, i].contiguous(), enc[i][1]) for i in range(b)], 0)


class RecurrentFlowCompleteNet(nn.Module):
    def __init__(self, model_path=None):
        super().__init__()
        tree = ParamTree()
        populate(tree, fc_schema())
        for name, child in tree._modules.items():
            self.add_module(name, child)
        for m in ("backward_", "forward_"):      # SecondOrderDeformableAlignment.init_offset (:27-28)
            d = self.feat_prop_module.deform_align._modules[m].conv_offset._modules["6"]
            d.weight.data.zero_()
            d.bias.data.zero_()
        if model_path is not None:
            print('Pretrained flow completion model has loaded...')
            ckpt = torch.load(model_path, map_location='cpu')
            self.load_state_dict(ckpt, strict=True)
        self._engine = None

    def _get_engine(self, dtype, device):
        key = (dtype, str(device), sum(p._version for p in self.parameters()))
        if self._engine is None or self._engine[0] != key:
            sd = {k: v.detach().float().cpu() for k, v in self.state_dict().items()}
            self._engine = (key, _FCEngine(sd, dtype, device))
        return self._engine[1]

    @hip.on_input_device
    @torch.no_grad()
    def forward(self, masked_flows, masks):
        hip.require_gpu(masked_flows, "RecurrentFlowCompleteNet")
        if self.training:
            raise NotImplementedError("training (edge branch) is outside the inference hot path")
        b, t, _, h, w = masked_flows.size()
        if h % 8 or w % 8:
            raise ValueError(f"H, W must be multiples of 8 (got {h}x{w})")
        eng = self._get_engine(masked_flows.dtype, masked_flows.device)
        flow = eng.forward(masked_flows.contiguous(), masks.to(masked_flows.dtype).contiguous())
        return flow, None

    def forward_bidirect_flow(self, masked_flows_bi, masks):
        """reference :312-337 — masks: b t 1 h w; flows: b t-1 2 h w."""
        masks_forward = masks[:, :-1, ...].contiguous()
        masks_backward = masks[:, 1:, ...].contiguous()
        masked_flows_forward = masked_flows_bi[0] * (1 - masks_forward)
        masked_flows_backward = masked_flows_bi[1] * (1 - masks_backward)
        # The reference runs the net twice (forward flows, then the time-flipped backward flows).  The two sequences
        # are independent samples, so they are stacked along the batch axis: same results, and the 2*t sequential
        # propagation steps are paid once instead of twice.
        nb = masked_flows_forward.size(0)
        pred, pred_edges = self.forward(torch.cat([masked_flows_forward, torch.flip(masked_flows_backward, dims=[1])], 0),
                                        torch.cat([masks_forward, torch.flip(masks_backward, dims=[1])], 0))
        pred_flows_forward = pred[:nb]
        pred_flows_backward = torch.flip(pred[nb:], dims=[1])
        return [pred_flows_forward, pred_flows_backward], [pred_edges, pred_edges]

    def combine_flow(self, masked_flows_bi, pred_flows_bi, masks):
        """reference :340-347."""
        masks_forward = masks[:, :-1, ...].contiguous()
        masks_backward = masks[:, 1:, ...].contiguous()
        pred_flows_forward = pred_flows_bi[0] * masks_forward + masked_flows_bi[0] * (1 - masks_forward)
        pred_flows_backward = pred_flows_bi[1] * masks_backward + masked_flows_bi[1] * (1 - masks_backward)
        return pred_flows_forward, pred_flows_backward
