"""Host side of the implicit-GEMM convolution: weight packing, K-chunk tables and launch plumbing.

Activations are NHWC torch tensors ``[N, H, W, Cstride]``; a *source* is ``(tensor, choff)`` = the channel
window starting at ``choff``.  Channel counts of every source are padded to a multiple of 8 (the K-chunk
of the kernel); padded channels must hold finite values (zero-initialised buffers) because their packed
weights are zero.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import hip

_CHECK_INPLACE = os.environ.get("PP_CHECK_INPLACE") == "1"     # diagnostic: see check_inplace
ACTS = {None: hip.ACT_NONE, "none": hip.ACT_NONE, "relu": hip.ACT_RELU, "lrelu": hip.ACT_LRELU,
        "sigmoid": hip.ACT_SIGMOID, "tanh": hip.ACT_TANH, "gelu": hip.ACT_GELU}


def pad8(c):
    return (int(c) + 7) // 8 * 8


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def fold_batchnorm(weight, bias, bn_weight, bn_bias, mean, var, eps=1e-5):
    """conv followed by eval-mode BatchNorm == conv with scaled weights (RAFT cnet, RAFT/extractor.py:126)."""
    s = bn_weight.double() / torch.sqrt(var.double() + eps)
    w = weight.double() * s.view(-1, 1, 1, 1)
    b = (bias.double() - mean.double()) * s + bn_bias.double()
    return w.float(), b.float()


KT_PLANE_LO = 1 << 24      # K-table code bits of a split-plane expansion: the chunk reads the lo plane of its source ...
KT_WEIGHT_LO = 1 << 25     # ... / multiplies with W_lo = fp16(W - fp16(W)) instead of W_hi = fp16(W)


def split_ktable(ktable, src_lo):
    """Split-plane ("f16x3") expansion of a K table built by hip.build_ktable (channel-block-major order).

    A split-plane source stores the fp32 value of channel c as hi = fp16(v) at c and lo = fp16(v - hi) at c + src_lo[s].  Every
    block of the table (all taps of up to 64 channels of one source) is walked three times: hi plane x W_hi, lo plane x W_hi,
    hi plane x W_lo (smallest terms last does not matter for the fp32 accumulators) -- so the fp16 kernels compute
    hi*W_hi + lo*W_hi + hi*W_lo without knowing it.  The plane / weight choice of a chunk is recorded in bits 24 / 25 of its
    code word (the kernels read bits 0..23 only); pack_weight() follows them.  Returns int32 [3 * chunks (padded to 8) + 1, 4]."""
    kt = np.asarray(ktable)
    if kt.shape[0] % 8 == 1:
        kt = kt[:-1]
    live = kt[(kt[:, 2] & 0xff) != 255]
    key = (live[:, 2] & 0xff).astype(np.int64) * (1 << 20) + live[:, 3] // 64          # (source, 64-channel block)
    starts = [0] + [i for i in range(1, len(live)) if key[i] != key[i - 1]] + [len(live)]
    rows = []
    for a, b in zip(starts[:-1], starts[1:]):
        blk = live[a:b]
        lo = blk.copy()
        lo[:, 3] += np.asarray(src_lo, dtype=np.int64)[blk[:, 2] & 0xff].astype(np.int32)
        lo[:, 2] |= KT_PLANE_LO
        wl = blk.copy()
        wl[:, 2] |= KT_WEIGHT_LO
        rows += [blk, lo, wl]
    out = np.concatenate(rows, 0)
    pad = (-out.shape[0]) % 8
    tail = np.zeros((pad + 1, 4), dtype=np.int32)
    tail[:pad, 2] = 255
    return np.ascontiguousarray(np.concatenate([out, tail], 0).astype(np.int32))


def tri_ktable(taps, src_cpad, src_lo):
    """K table of the TRI-PRODUCT split-plane format (halo-tile kernel, csrc/conv_halo.h SPLIT): one block = 32 channels of BOTH
    planes of one source; per tap 8 chunks -- 4 of the hi plane, then the same 4 channel chunks of the lo plane (flags PLANE_LO |
    WEIGHT_LO: the packed weight row holds [32 ch W_hi | 32 ch W_lo] per tap).  The kernel multiplies the four fragment sets of a
    tap step as W_hi x A_hi + W_hi x A_lo + W_lo x A_hi, so -- unlike split_ktable's -- this table is NOT a plain K walk: only the
    tri-product kernels (halo-tile kernel, v2 kernel with TRI; the CPU emulation's matching branch) may consume it.  Works for any tap
    list; a source that is no multiple of 32 channels gets a ragged last block padded with zero chunks.  Returns int32 [chunks + 1, 4]."""
    rows = []
    for s, (c, lo) in enumerate(zip(src_cpad, src_lo)):
        assert c % 8 == 0 and lo % 8 == 0 and lo >= c
        for cb in range(0, c, 32):
            n = min(4, (c - cb) // 8)          # chunks of this block (the last block of a source that is no multiple of 32 is ragged:
            for t, (dy, dx) in enumerate(taps):       # zero chunks -- source id 255 -- fill both halves of the step)
                for plane in (0, 1):
                    fl = (KT_PLANE_LO | KT_WEIGHT_LO) if plane else 0
                    for j in range(4):
                        rows.append([dy, dx, s | (t << 16) | fl, cb + 8 * j + (lo if plane else 0)] if j < n else [0, 0, 255 | fl, 0])
    rows.append([0, 0, 0, 0])
    return np.ascontiguousarray(np.asarray(rows, dtype=np.int32))


def pack_weight(weight, src_channels, groups=1, ktable=None, src_lo=None):
    """weight [Cout, Cin_g, kh, kw] (any float dtype, CPU or GPU) -> (packed fp32 [groups, cout_pad, K], K, cout_g).

    The K order is whatever ``ktable`` (int32 [kchunks(+1), 4] from pp_conv_build_ktable: {dy, dx, src | grp<<8 | tap<<16,
    choff} per 8-channel chunk, src 255 = zero chunk) says -- the table is the single source of truth shared with the
    kernels; the tap id indexes the kernel window in row-major (ky, kx) order.  Every source is padded to a multiple of
    8 channels (zero weights)."""
    w = weight.detach().float().cpu()
    cout, cin_g, kh, kw = w.shape
    assert sum(src_channels) == cin_g, (src_channels, cin_g)
    parts, off, bases, base = [], 0, [], 0
    for c in src_channels:
        part = w[:, off:off + c]
        if pad8(c) != c:
            part = torch.cat([part, part.new_zeros(cout, pad8(c) - c, kh, kw)], 1)
        parts.append(part)
        bases.append(base)
        base += pad8(c)
        off += c
    ctot = base
    wt = torch.cat(parts, 1).permute(0, 2, 3, 1).reshape(cout, kh * kw * ctot)          # [Cout, tap, Cpad_total]
    if ktable is None:
        ktable = hip.build_ktable([(ky, kx) for ky in range(kh) for kx in range(kw)], [pad8(c) for c in src_channels])
    kt = np.asarray(ktable)
    if kt.shape[0] % 8 == 1:
        kt = kt[:-1]                                                                     # drop the zero-page row
    K = kt.shape[0] * 8
    code = kt[:, 2].astype(np.int64)
    src, tap, choff = code & 0xff, (code >> 16) & 0xff, kt[:, 3].astype(np.int64)
    live = src != 255
    if src_lo is not None:       # split-plane table: a lo-plane chunk multiplies with the weights of the channel it shadows
        choff = choff - np.where(live & ((code & KT_PLANE_LO) != 0), np.asarray(list(src_lo) + [0] * 256, dtype=np.int64)[np.minimum(src, len(src_lo))], 0)
    col0 = np.where(live, tap * ctot + np.asarray(bases + [0] * 256, dtype=np.int64)[np.minimum(src, len(bases))] + choff, 0)
    cols = torch.from_numpy((col0[:, None] + np.arange(8)[None, :]).reshape(-1))
    wk = wt[:, cols] * torch.from_numpy(np.repeat(live, 8)).float()[None, :]              # [Cout, K]
    if (code & (KT_PLANE_LO | KT_WEIGHT_LO)).any():      # split-plane table: W_hi = fp16(W) / W_lo = fp16(W - W_hi) per chunk
        w_hi = wk.half().float()
        w_lo = (wk - w_hi).half().float()
        wk = torch.where(torch.from_numpy(np.repeat((code & KT_WEIGHT_LO) != 0, 8))[None, :], w_lo, w_hi)
    cout_g = cout // groups
    cout_pad = (cout_g + 15) // 16 * 16
    packed = wk.new_zeros(groups, cout_pad, K)
    packed[:, :cout_g] = wk.view(groups, cout_g, K)
    return packed, K, cout_g


def pconv_act_none(act):
    return act is None or ACTS[act] == hip.ACT_NONE


inplace_findings = []     # filled by check_inplace when PP_CHECK_INPLACE=1 or a hazard recorder is active (diagnostic; see tools/check_hazards.py)


def _windows_overlap(a, b):
    """a, b: (first byte, row stride in bytes, window bytes, rows) of two channel windows of NHWC buffers.  True when some byte belongs to both."""
    a0, sa, wa, ra = a
    b0, sb, wb, rb = b
    if a0 + (ra - 1) * sa + wa <= b0 or b0 + (rb - 1) * sb + wb <= a0:
        return False
    if sa != sb:
        return True                       # differently strided views of overlapping extents: not analysed, reported
    d = (b0 - a0) % sa
    return d < wa or d + wb > sa


def check_inplace(layer, srcs, out, out_choff, fuse=None, dcn_offmask=None):
    """INTRA-launch hazard of one convolution launch: does the output window share bytes with a window the SAME launch reads at OTHER
    pixels (its sources -- every tap reads neighbours, other cout tiles re-read the pixel -- or the deformable offsets)?  Blocks of one
    launch are unordered, so such a launch is a race no stream / event edge can repair; the hazard recorder (inter-launch) cannot see it.
    Epilogue operands (residual, pre-activation addend, h / z of the fused gating) are read at the very element that is written: in place
    is well defined for them and they are not checked."""
    esz = out.element_size()
    rows_o = out.shape[0] * out.shape[1] * out.shape[2]
    so = out.shape[-1] * esz
    split_out = layer.split and out.dtype == torch.float16
    zr = fuse is not None and fuse.get("out2") is not None          # fused z | r gating: `out` gets the first `split` couts, out2 the rest
    plane = out.shape[-1] // 2 if split_out else out.shape[-1]
    wo = min(int(fuse["split"]) if zr else layer.cout_pad * layer.groups, plane - out_choff)
    outs = [(out.data_ptr() + out_choff * esz, so, wo * esz, rows_o)]
    if split_out:
        outs.append((out.data_ptr() + (plane + out_choff) * esz, so, wo * esz, rows_o))
    if zr:
        ot, oc = fuse["out2"] if not torch.is_tensor(fuse["out2"]) else (fuse["out2"], 0)
        plane2 = ot.shape[-1] // 2 if split_out else ot.shape[-1]
        w2 = min(layer.cout_pad - int(fuse["split"]), plane2 - oc)
        outs.append((ot.data_ptr() + oc * esz, ot.shape[-1] * esz, w2 * esz, rows_o))
        if split_out:
            outs.append((ot.data_ptr() + (plane2 + oc) * esz, ot.shape[-1] * esz, w2 * esz, rows_o))
    reads = []
    for i, s in enumerate(srcs):
        t, co = (s, 0) if torch.is_tensor(s) else s
        rows = t.shape[0] * t.shape[1] * t.shape[2]
        st = t.shape[-1] * t.element_size()
        reads.append((f"source {i}", (t.data_ptr() + co * t.element_size(), st, layer.src_cpad[i] * t.element_size(), rows)))
        if layer.split:
            reads.append((f"source {i} (lo plane)", (t.data_ptr() + (layer.src_lo[i] + co) * t.element_size(), st, layer.src_cpad[i] * t.element_size(), rows)))
    if dcn_offmask is not None:
        reads.append(("deformable offsets / masks", (dcn_offmask.data_ptr(), dcn_offmask.shape[-1] * dcn_offmask.element_size(),
                                                     dcn_offmask.shape[-1] * dcn_offmask.element_size(), rows_o)))
    for name, r in reads:
        for o in outs:
            if _windows_overlap(r, o):
                import traceback
                where = next((f"{f.filename.split('/')[-1]}:{f.lineno}" for f in reversed(traceback.extract_stack()[:-1])
                              if "conv.py" not in f.filename and "cpu_emulation" not in f.filename), "?")
                inplace_findings.append(f"conv {layer.kh}x{layer.kw} cout {layer.cout} [{where}]: the output window overlaps {name}")
    for i in range(len(outs)):
        for j in range(i + 1, len(outs)):
            if _windows_overlap(outs[i], outs[j]):
                inplace_findings.append(f"conv {layer.kh}x{layer.kw} cout {layer.cout}: two output windows of one launch overlap")


class ConvLayer:
    """One convolution / linear layer prepared for pp_conv2d (weights packed once, on the device)."""

    def __init__(self, weight, bias, *, stride=1, padding=0, dilation=1, groups=1, src_channels=None,
                 pad_mode="zeros", dtype=torch.float16, device="cuda", taps=None, dcn_groups=0, split3=False, split=False, src_lo=None, tri=None):
        if weight.dim() == 2:                       # nn.Linear
            weight = weight[:, :, None, None]
        cout, cin_g, kh, kw = weight.shape
        self.stride, self.padding, self.dilation = _pair(stride), _pair(padding), _pair(dilation)
        self.groups = groups
        self.kh, self.kw = kh, kw
        self.src_channels = list(src_channels) if src_channels is not None else [cin_g]
        self.src_cpad = [pad8(c) for c in self.src_channels]
        self.pad_mode = 1 if pad_mode == "replicate" else 0
        self.dtype = dtype
        # dense dilation-1 window in row-major order: the halo-tile kernel may serve it (pp_conv_args_t.tap_h / tap_w)
        self.tap_hw = (kh, kw) if (taps is None and self.dilation == (1, 1) and dcn_groups == 0) else (0, 0)
        if taps is None:
            taps = [(ky * self.dilation[0], kx * self.dilation[1]) for ky in range(kh) for kx in range(kw)]
        kt = hip.build_ktable(taps, self.src_cpad, dcn_groups)
        # split=True: split-plane ("f16x3") layer -- fp16 kernels, every source / epilogue operand a pair of fp16 planes
        # (hi, lo) of one buffer, lo plane src_lo[i] channels after the hi plane (default: a dedicated buffer [.., 2 * cpad])
        self.split = bool(split)
        self.tri = False
        if self.split:
            if dtype != torch.float16 or dcn_groups or groups != 1:
                raise ValueError("split-plane layers are plain fp16 convolutions (groups == 1)")
            self.src_lo = [int(v) for v in (src_lo if src_lo is not None else self.src_cpad)]
            assert len(self.src_lo) == len(self.src_cpad) and all(l >= c and l % 8 == 0 for l, c in zip(self.src_lo, self.src_cpad))
            # tri-product format (halo-tile kernel: 48 MFMAs per 16 fragment reads) for the stride-1 "same" 3x3 / 1x5 / 5x1 layers
            # over 32-channel-multiple sources; every other layer walks its blocks three times through the LDS-DMA (v2) kernel
            pad_same = self.padding == ((kh - 1) // 2, (kw - 1) // 2) and self.stride == (1, 1)
            halo = (self.tap_hw in ((3, 3), (1, 5), (5, 1)) and pad_same and all(c % 32 == 0 for c in self.src_cpad) and
                    not (16 < cout < 48) and self.pad_mode == 0)
            # ... and, through the v2 kernel's tri step (64-wide K steps: couts > 32), every other layer whose 32-channel blocks are
            # (nearly) full: 1x1 over 324 / 128 / 256 channels, strided 3x3 (a 7x7 over 8 or a 7x1 over 16 channels would be 2-4x zeros)
            full = sum((c + 31) // 32 * 32 for c in self.src_cpad) <= 1.1 * sum(self.src_cpad)
            self.tri = halo or (full and cout > 32)
            if tri is not None:      # (tests: tri=False walks a layer through the v2 kernel's plain format)
                assert self.tri or not tri or cout > 32, "tri-product format: halo family, or more than 32 couts (v2 kernel, 64-wide K steps)"
                self.tri = bool(tri)
            kt = tri_ktable(taps, self.src_cpad, self.src_lo) if self.tri else split_ktable(kt, self.src_lo)
        packed, K, cout_g = pack_weight(weight, self.src_channels, groups, ktable=kt,     # K order = the table's order
                                        src_lo=self.src_lo if self.split else None)
        self.cout_g, self.cout = cout_g, cout
        self.cout_pad = packed.shape[1]
        self.K = K
        self.weight = packed.to(device=device, dtype=dtype).contiguous()
        self.bias = None if bias is None else bias.detach().float().to(device).contiguous()
        assert (kt.shape[0] - 1) * 8 == K, (kt.shape, K)     # + the trailing zero-page entry
        self.kchunks = kt.shape[0] - 1
        self.ktable = torch.from_numpy(kt).to(device)
        self.dcn = dcn_groups > 0
        # K steps of 4 / 8 chunks are (tap, source)-uniform when every source is a multiple of 32 / 64 channels
        self.ktable_uniform = (4 if all(c % 32 == 0 for c in self.src_cpad) else 0) | (8 if all(c % 64 == 0 for c in self.src_cpad) else 0)
        if self.split and self.tri:      # every run of 8 chunks is one (tap, source) [4 hi + 4 lo chunks] when no block is ragged
            self.ktable_uniform = 8 if all(c % 32 == 0 for c in self.src_cpad) else 0
        self.impl = 0            # pp_conv_args_t.impl: 0 auto, 1 register-staged kernel, >= 10 a specific LDS-DMA tile
        # fp32 tensors, products on the fp16 matrix cores as hi*hi + hi*lo + lo*hi (fp32 accumulate): ~2^-21 per
        # product instead of fp32's 2^-24, 5x the rate of the exact fp32 MFMA (pp_conv_args_t.impl 3)
        self.split3 = bool(split3)
        if self.split3 and (dtype != torch.float32 or dcn_groups):
            raise ValueError("split3 (3 x fp16 MFMA on fp32 data) applies to plain fp32 convolutions")

    def out_hw(self, H, W):
        OH = (H + 2 * self.padding[0] - self.dilation[0] * (self.kh - 1) - 1) // self.stride[0] + 1
        OW = (W + 2 * self.padding[1] - self.dilation[1] * (self.kw - 1) - 1) // self.stride[1] + 1
        return OH, OW

    def __call__(self, srcs, out=None, out_choff=0, act=None, act_param=0.0, out_scale=1.0, residual=None,
                 res_choff=0, act2=None, out_dtype=None, dcn_offmask=None, out_hw=None, preadd=None, fuse=None):
        """srcs: list of tensors or (tensor, choff) (NHWC, dtype == layer dtype).  Returns `out` NHWC.
        preadd: tensor or (tensor, choff) added to (acc + bias) * out_scale BEFORE the activation (a partial sum another
        convolution computed ahead of time).  fuse: fused SepConvGRU gating (pp_conv_args_t.fuse), one of
          dict(kind="gru_zr", h=(t, choff), out2=(t, choff), split=C)   out <- z = act(v)[:C], out2 <- act(v)[C:] * h
          dict(kind="gru_h", h=(t, choff), z=(t, choff))                 out <- (1 - z) * h + z * act(v)
          dict(kind="dcn_om", mag=m, flow=(t, choff) | None, split=288)  out <- m * tanh(v[:split]) + flow(y, x) | sigmoid(v[split:])"""
        srcs = [(s, 0) if torch.is_tensor(s) else s for s in srcs]
        assert len(srcs) == len(self.src_channels), (len(srcs), self.src_channels)
        x0 = srcs[0][0]
        N, H, W = x0.shape[0], x0.shape[1], x0.shape[2]
        OH, OW = out_hw if out_hw is not None else self.out_hw(H, W)
        odt = out_dtype or self.dtype
        if out is None:
            cp = pad8(self.cout)
            out = (torch.zeros if cp != self.cout else torch.empty)((N, OH, OW, 2 * cp if (self.split and odt == torch.float16) else cp),
                                                                    dtype=odt, device=x0.device)
        split_out = self.split and out.dtype == torch.float16
        if _CHECK_INPLACE or hip._hazard.active() is not None:
            check_inplace(self, srcs, out, out_choff, fuse, dcn_offmask)
        a = hip.ConvArgs()
        a.dtype = hip.dtype_code(self.dtype)
        a.N, a.H, a.W, a.OH, a.OW = N, H, W, OH, OW
        a.stride_h, a.stride_w = self.stride
        a.pad_h, a.pad_w = self.padding
        a.pad_mode = self.pad_mode
        a.groups, a.cout_g, a.cout_pad, a.kchunks, a.nsrc = self.groups, self.cout_g, self.cout_pad, self.kchunks, len(srcs)
        for i, (t, choff) in enumerate(srcs):
            if t.dtype != self.dtype or not t.is_contiguous() or t.shape[:3] != x0.shape[:3]:
                raise ValueError(f"conv source {i}: dtype {t.dtype} / shape {tuple(t.shape)} incompatible")
            if self.split and (t.shape[-1] != 2 * self.src_lo[i] or choff + self.src_cpad[i] > self.src_lo[i]):
                raise ValueError(f"conv source {i}: split-plane buffer of {t.shape[-1]} channels, window {choff}+{self.src_cpad[i]}, "
                                 f"layer built for lo offset {self.src_lo[i]}")
            a.src[i].ptr = t.data_ptr()
            a.src[i].cstride = t.shape[-1]
            a.src[i].choff = choff
            a.src[i].cgroup = self.src_channels[i] if self.groups > 1 else 0
            a.src[i].lo_off = self.src_lo[i] if self.split else 0
        a.ktable = self.ktable.data_ptr()
        a.weight = self.weight.data_ptr()
        a.weight_gstride = self.cout_pad * self.K
        a.bias = self.bias.data_ptr() if self.bias is not None else None
        a.act, a.act_param, a.out_scale = ACTS[act], float(act_param), float(out_scale)
        if residual is not None:
            assert residual.dtype == self.dtype and residual.is_contiguous()
            a.residual, a.res_cstride, a.res_choff = residual.data_ptr(), residual.shape[-1], res_choff
        a.act2 = ACTS[act2]
        a.out_dtype = hip.dtype_code(out.dtype)
        assert out.is_contiguous() and out.shape[:3] == (N, OH, OW)
        a.out, a.out_cstride, a.out_choff, a.out_cgroup = out.data_ptr(), out.shape[-1], out_choff, self.cout_g
        if dcn_offmask is not None:
            assert self.dcn and dcn_offmask.dtype == self.dtype and dcn_offmask.is_contiguous()
            a.dcn_offmask, a.dcn_cstride, a.dcn_mask_off = dcn_offmask.data_ptr(), dcn_offmask.shape[-1], 288
        win = lambda x: (x, 0) if torch.is_tensor(x) else x
        if self.split:      # every fp16 operand of the epilogue is split-plane: lo plane half a pixel row after the hi plane
            a.split = 2 if self.tri else 1
            a.out_lo = out.shape[-1] // 2 if split_out else 0
            a.res_lo = residual.shape[-1] // 2 if residual is not None else 0
            assert fuse is None or fuse["kind"] != "dcn_om"
        if preadd is not None:
            t, co = win(preadd)
            assert t.dtype == self.dtype and t.is_contiguous() and t.shape[:3] == (N, OH, OW)
            a.preadd, a.preadd_cstride, a.preadd_choff = t.data_ptr(), t.shape[-1], co
            a.preadd_lo = t.shape[-1] // 2 if self.split else 0
        if fuse is not None and fuse["kind"] == "dcn_om":
            # offset / mask head of a deformable alignment: mag * tanh(offsets) + flow | sigmoid(masks) in the epilogue
            assert pconv_act_none(act) and out.dtype == self.dtype and preadd is None and residual is None
            a.fuse, a.fuse_split, a.act_param = hip.FUSE_DCN_OFFMASK, int(fuse.get("split", 288)), float(fuse["mag"])
            if fuse.get("flow") is not None:
                (ft, fc) = win(fuse["flow"])
                assert ft.dtype == self.dtype and ft.is_contiguous() and ft.shape[:3] == (N, OH, OW) and fc % 2 == 0
                a.fuse_a, a.fuse_a_cstride, a.fuse_a_choff = ft.data_ptr(), ft.shape[-1], fc
        elif fuse is not None:
            (ht, hc) = win(fuse["h"])
            assert ht.dtype == self.dtype and ht.is_contiguous() and out.dtype == self.dtype
            a.fuse_a, a.fuse_a_cstride, a.fuse_a_choff = ht.data_ptr(), ht.shape[-1], hc
            a.fuse_a_lo = ht.shape[-1] // 2 if self.split else 0
            if fuse["kind"] == "gru_zr":
                (ot, oc) = win(fuse["out2"])
                assert ot.dtype == self.dtype and ot.is_contiguous()
                a.fuse, a.fuse_split = hip.FUSE_GRU_ZR, int(fuse["split"])
                a.out2, a.out2_cstride, a.out2_choff = ot.data_ptr(), ot.shape[-1], oc
                a.out2_lo = ot.shape[-1] // 2 if self.split else 0
            elif fuse["kind"] == "gru_h":
                (zt, zc) = win(fuse["z"])
                assert zt.dtype == self.dtype and zt.is_contiguous()
                a.fuse = hip.FUSE_GRU_H
                a.fuse_b, a.fuse_b_cstride, a.fuse_b_choff = zt.data_ptr(), zt.shape[-1], zc
                a.fuse_b_lo = zt.shape[-1] // 2 if self.split else 0
            else:
                raise ValueError(fuse["kind"])
        a.impl = 3 if self.split3 else self.impl
        a.ktable_uniform = self.ktable_uniform
        a.tap_h, a.tap_w = self.tap_hw
        # (no references are kept past the launch: the caching allocator is stream-ordered, so the operands may be released as soon as the
        #  kernel is enqueued -- round 2 parked (srcs, out, ...) of the LAST call on every layer, which kept ~70 GB of dead activations of
        #  the flow-completion / RAFT layers alive through the rest of a 720p pass)
        hip.conv2d_raw(a, cin_read=sum(self.src_cpad) * self.groups, on=x0, split_k=(2 if self.tri else 3) if self.split else 0)
        return out


_gemm_tables = {}


_split_gemm_tables = {}


def batched_gemm_nt_split(a, b, out_scale=1.0, impl=0):
    """Split-plane ("f16x3") batched GEMM: a, b fp16 [B, M | N, 2K] = hi | lo planes of fp32 operands;
    out[b, m, n] = out_scale * sum_k a[b, m, k] * b[b, n, k] as hi*hi + lo*hi + hi*lo on the fp16 matrix cores, fp32 output.
    Tri-product K format: the A side needs no copy (its K table lists 4 hi + 4 lo chunks per 32-channel block); the B side is the
    kernel's dense "weight" operand, so its rows are gathered once into that K order ([32 ch hi | 32 ch lo] per block; N is padded
    to a multiple of 16 rows with zeros -- the pooled levels of a pyramid have any size).  impl: tile configuration (0 = auto).
    RAFT all-pairs correlation volume at reference precision (RAFT/corr.py:52-60)."""
    B, M, K2 = a.shape
    Bb, Nn, K2b = b.shape
    K = K2 // 2
    assert B == Bb and K2 == K2b and K % 64 == 0 and a.dtype == b.dtype == torch.float16 and a.is_contiguous() and b.is_contiguous()
    key = (K, str(a.device))
    if key not in _split_gemm_tables:
        kt = tri_ktable([(0, 0)], [K], [K])            # per 32-channel block: 4 hi chunks, then the same 4 chunks of the lo plane
        code = kt[:-1, 2].astype(np.int64)
        plane = np.where(code & KT_WEIGHT_LO, 1, 0)                                  # B-side plane of every chunk
        cols = (plane * K + kt[:-1, 3] % K)[:, None] + np.arange(8)[None, :]         # [chunks, 8] columns of b's rows (lo-plane A chunks: c + K -> c)
        cols[(code & 0xff) == 255] = 0
        _split_gemm_tables[key] = (torch.from_numpy(kt).to(a.device), torch.from_numpy(cols.reshape(-1)).to(a.device), kt.shape[0] - 1)
    kt, cols, kchunks = _split_gemm_tables[key]
    npad = (Nn + 15) // 16 * 16
    if npad == Nn:
        bt = b.index_select(2, cols)                                                    # [B, N, 2K] in K-table order
    else:
        bt = torch.zeros((B, npad, cols.numel()), dtype=b.dtype, device=b.device)
        bt[:, :Nn] = b.index_select(2, cols)
    out = torch.empty((B, M, Nn), dtype=torch.float32, device=a.device)
    g = hip.ConvArgs()
    g.dtype = hip.PP_F16
    g.N, g.H, g.W, g.OH, g.OW = 1, 1, M, 1, M
    g.stride_h = g.stride_w = 1
    g.groups, g.cout_g, g.cout_pad, g.kchunks, g.nsrc = B, Nn, npad, kchunks, 1
    g.src[0].ptr, g.src[0].cstride, g.src[0].choff, g.src[0].cgroup = a.data_ptr(), K2, 0, 0
    g.src[0].lo_off = K
    g.ktable, g.weight, g.weight_gstride = kt.data_ptr(), bt.data_ptr(), npad * kchunks * 8
    g.act, g.out_scale, g.act2 = hip.ACT_NONE, float(out_scale), hip.ACT_NONE
    g.out_dtype = hip.PP_F32
    g.split = 2                    # tri-product K step (plain fp32 output: no split-plane epilogue operands)
    # tile configuration (tri-product steps exist for the 64-wide-K tiles 12 / 13 / 22 only): 128 x 128 tiles at two blocks per CU --
    # measured 7 % faster than the 256 x 128 tile the cout >= 512 rule of the convolutions would pick (16.6 vs 17.8 ms per 35-pair level-0
    # volume at 720p, profiles/r3u_split_sweep.txt) --, 256 x 64 for the tiny pooled levels of small inputs
    g.impl = int(impl) if impl else (12 if Nn > 64 else 22)
    g.ktable_uniform = 8
    g.out, g.out_cstride, g.out_choff, g.out_cgroup = out.data_ptr(), Nn, 0, 0
    g.src_gstride, g.out_gstride = M * K2, M * Nn
    hip.conv2d_raw(g, cin_read=K * B, on=a, split_k=2)
    del bt
    return out


def batched_gemm_nt(a, bt, out_scale=1.0, split3=False):
    """out[b, m, n] = out_scale * sum_k a[b, m, k] * bt[b, n, k]   (fp32 output; a, bt same dtype, K % 32 == 0).
    Used for the RAFT all-pairs correlation volume (RAFT/corr.py:52-60)."""
    B, M, K = a.shape
    Bb, Nn, Kb = bt.shape
    assert B == Bb and K == Kb and K % 64 == 0 and a.dtype == bt.dtype and a.is_contiguous() and bt.is_contiguous()
    key = (K, str(a.device))
    if key not in _gemm_tables:
        _gemm_tables[key] = torch.from_numpy(hip.build_ktable([(0, 0)], [K])).to(a.device)
    kt = _gemm_tables[key]
    out = torch.empty((B, M, Nn), dtype=torch.float32, device=a.device)
    g = hip.ConvArgs()
    g.dtype = hip.dtype_code(a.dtype)
    g.N, g.H, g.W, g.OH, g.OW = 1, 1, M, 1, M
    g.stride_h = g.stride_w = 1
    g.groups, g.cout_g, g.cout_pad, g.kchunks, g.nsrc = B, Nn, Nn, K // 8, 1
    g.src[0].ptr, g.src[0].cstride, g.src[0].choff, g.src[0].cgroup = a.data_ptr(), K, 0, 0
    g.ktable, g.weight, g.weight_gstride = kt.data_ptr(), bt.data_ptr(), Nn * K
    g.act, g.out_scale, g.act2 = hip.ACT_NONE, float(out_scale), hip.ACT_NONE
    g.out_dtype = hip.PP_F32
    g.ktable_uniform = 12
    g.impl = 3 if split3 else 0
    g.out, g.out_cstride, g.out_choff, g.out_cgroup = out.data_ptr(), Nn, 0, 0
    g.src_gstride, g.out_gstride = M * K, M * Nn
    hip.conv2d_raw(g, cin_read=K * B, on=a)
    return out
