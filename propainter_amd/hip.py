"""ctypes binding of libpropainter_hip.so (the C-ABI declared in include/propainter_hip.h).

There is no CPU fallback: every op below requires the HIP library, and raises if it cannot be
loaded or if a tensor is not on a GPU.  Tensors are passed as raw ``data_ptr()`` + explicit sizes and
all work is enqueued on torch's current stream of the device that owns the tensors.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import build as _build
from . import hazard as _hazard

PP_F32, PP_F16, PP_F16S = 0, 1, 2      # PP_F16S: split-plane fp16 pair (value = hi + lo, lo plane at cstride / 2 in the aux ops)
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID, ACT_TANH, ACT_GELU = 0, 1, 2, 3, 4, 5
MAX_SRC = 4
FUSE_NONE, FUSE_GRU_ZR, FUSE_GRU_H, FUSE_DCN_OFFMASK = 0, 1, 2, 3


class ConvSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("cstride", C.c_int32), ("choff", C.c_int32), ("cgroup", C.c_int32),
                ("lo_off", C.c_int32)]


class ConvArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("OH", C.c_int32),
        ("OW", C.c_int32), ("stride_h", C.c_int32), ("stride_w", C.c_int32), ("pad_h", C.c_int32),
        ("pad_w", C.c_int32), ("pad_mode", C.c_int32), ("groups", C.c_int32), ("cout_g", C.c_int32),
        ("cout_pad", C.c_int32), ("kchunks", C.c_int32), ("nsrc", C.c_int32), ("src", ConvSrc * MAX_SRC),
        ("ktable", C.c_void_p), ("weight", C.c_void_p), ("weight_gstride", C.c_int64), ("bias", C.c_void_p),
        ("act", C.c_int32), ("act_param", C.c_float), ("out_scale", C.c_float), ("residual", C.c_void_p),
        ("res_cstride", C.c_int32), ("res_choff", C.c_int32), ("act2", C.c_int32), ("out_dtype", C.c_int32),
        ("out", C.c_void_p), ("out_cstride", C.c_int32), ("out_choff", C.c_int32), ("out_cgroup", C.c_int32),
        ("src_gstride", C.c_int64), ("out_gstride", C.c_int64), ("dcn_offmask", C.c_void_p),
        ("dcn_cstride", C.c_int32), ("dcn_mask_off", C.c_int32), ("impl", C.c_int32), ("ktable_uniform", C.c_int32),
        ("tap_h", C.c_int32), ("tap_w", C.c_int32),
        ("preadd", C.c_void_p), ("preadd_cstride", C.c_int32), ("preadd_choff", C.c_int32),
        ("fuse", C.c_int32), ("fuse_split", C.c_int32),
        ("fuse_a", C.c_void_p), ("fuse_a_cstride", C.c_int32), ("fuse_a_choff", C.c_int32),
        ("fuse_b", C.c_void_p), ("fuse_b_cstride", C.c_int32), ("fuse_b_choff", C.c_int32),
        ("out2", C.c_void_p), ("out2_cstride", C.c_int32), ("out2_choff", C.c_int32),
        ("split", C.c_int32), ("out_lo", C.c_int32), ("out2_lo", C.c_int32), ("preadd_lo", C.c_int32), ("res_lo", C.c_int32),
        ("fuse_a_lo", C.c_int32), ("fuse_b_lo", C.c_int32), ("pad2_", C.c_int32),
        ("dcn_stats", C.c_void_p),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("T", C.c_int32), ("Hp", C.c_int32), ("Wp", C.c_int32),
        ("C", C.c_int32), ("heads", C.c_int32), ("wh", C.c_int32), ("ww", C.c_int32), ("n_rolled", C.c_int32),
        ("P", C.c_int32), ("n_tind", C.c_int32), ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p),
        ("qkv_cstride", C.c_int32), ("pk", C.c_void_p), ("pv", C.c_void_p), ("pkv_cstride", C.c_int32),
        ("own", C.c_void_p), ("rolled", C.c_void_p), ("tind", C.c_void_p), ("wmask", C.c_void_p),
        ("out", C.c_void_p), ("impl", C.c_int32), ("work_ints", C.c_int32), ("work", C.c_void_p),
        ("out_h", C.c_int32), ("out_w", C.c_int32),
    ]


_lib = None


def lib():
    """Loads (building in-tree with hipcc if needed) libpropainter_hip.so.  Raises on failure."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PP_LIB_PATH") or _build.build(force=False, verbose=False)      # (build: no-op unless the sources no longer match the built library; PP_LIB_PATH: a tools/build_variant.sh library for A/B tuning runs)
    L = C.CDLL(path)
    L.pp_last_error_string.restype = C.c_char_p
    for name in ("pp_conv2d", "pp_sparse_window_attention"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
    if L.pp_sizeof_conv_args() != C.sizeof(ConvArgs) or L.pp_sizeof_attn_args() != C.sizeof(AttnArgs):
        raise RuntimeError("libpropainter_hip.so ABI mismatch with propainter_amd/hip.py (struct sizes differ)")
    _lib = L
    return L


def loaded_library_path():
    return _build.LIB_PATH if _lib is not None else None


_FENCE_EVERY_LAUNCH = int(os.environ.get("PP_FENCE_EVERY_LAUNCH", "0"))      # diagnosis (profiles/r6_replay_bytes.txt): L2 write-back / invalidate kernel behind every engine launch


def _check(rc, what):
    if _hazard.active() is not None:          # capture-time hazard checker (propainter_amd/hazard.py): this launch's pointers are complete
        _hazard.active().flush(what)
    if _FENCE_EVERY_LAUNCH and rc == 0 and what != "pp_debug_cache_fence" and torch.cuda.is_available():
        lib().pp_debug_cache_fence(C.c_int(_FENCE_EVERY_LAUNCH), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        msg = lib().pp_last_error_string().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


class KernelProfiler:
    """Per-kernel-class HIP-event timing of the C-ABI launches (used by bench.py for the roofline figures).

    While installed (``with KernelProfiler() as kp``) every launch that goes through ``timed()`` is bracketed by a
    pair of HIP events recorded on the stream the kernel is launched on (torch's current stream) and tagged with its
    algorithmic FLOPs / bytes.  ``summary()`` synchronises and aggregates per class."""

    def __init__(self, detail=False):
        self.records = []          # (name, flops, bytes, ev0, ev1)
        self.detail = detail       # split the convolution classes by layer shape

    def __enter__(self):
        global _profiler
        self._prev, _profiler = _profiler, self
        return self

    def __exit__(self, *exc):
        global _profiler
        _profiler = self._prev
        return False

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, fl, by, e0, e1 in self.records:
            a = agg.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            a["launches"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += fl
            a["bytes"] += by
        return agg


_profiler = None


class FallbackStats:
    """Counts how often the data-dependent slow paths of two kernels are taken (round 5; bench.py's `fallback` / `stress` objects):

      * the volume-free correlation lookups (``pp_corr_lookup_otf[_split]``): a tile whose correlation windows' bounding box outgrows
        the LDS tile is processed as 16-pixel sub-tiles, and those as single pixels;
      * the patch-staged deformable convolution: a bilinear sample with a corner outside the tile's mean-shifted LDS patch is served by
        per-corner global reads.

    While installed (``with FallbackStats(device) as fs``) the wrappers pass a device counter block to the ``*_stats`` entry points / the
    counting instantiation of the deformable kernel (same arithmetic, same results); ``fs.read()`` synchronises and returns the counts
    and fractions.  Host-side state of this wrapper only -- the C-ABI stays stateless (the counter block is an argument)."""

    def __init__(self, device):
        self.buf = torch.zeros(8, dtype=torch.int64, device=device)

    def __enter__(self):
        global _stats
        self._prev, _stats = _stats, self
        return self

    def __exit__(self, *exc):
        global _stats
        _stats = self._prev
        return False

    def read(self):
        torch.cuda.synchronize(self.buf.device)
        v = [int(x) for x in self.buf.cpu().tolist()]
        units, sub, single, samples, outside = v[0], v[1], v[2], v[4], v[5]
        return {"corr_tile_levels": units, "corr_subtile_fallbacks": sub, "corr_single_pixel_fallbacks": single,
                "corr_subtile_fallback_frac": sub / units if units else None,
                # a (tile, level) unit that falls back splits into 2 sub-tiles of 16 pixels in the split-plane kernel and into 4 in the
                # fp16 kernel: both add to the same counters, so the fraction is given per fallback UNIT (exact for either kernel), and per
                # sub-tile under the split-plane kernel's count (the timed default; half that under the fp16 kernel)
                "corr_single_pixel_fallbacks_per_fallback_unit": single / sub if sub else (0.0 if units else None),
                "corr_single_pixel_fallback_frac_of_subtiles": single / (2 * sub) if sub else (0.0 if units else None),
                "corr_single_pixel_fallback_frac_of_subtiles_note": "2 sub-tiles per unit (corr_otf_split_kernel); corr_otf_kernel (f16) has 4: halve it there",
                "dcn_samples": samples, "dcn_out_of_patch_samples": outside,
                "dcn_out_of_patch_frac": outside / samples if samples else None}


_stats = None


def _stats_ptr(offset):
    """device pointer into the installed FallbackStats block (element `offset`), or NULL"""
    return C.c_void_p(_stats.buf.data_ptr() + 8 * offset) if _stats is not None else C.c_void_p(0)


def _nbytes(t):
    return t.numel() * t.element_size()


def timed(name, flops, nbytes, fn):
    """Runs ``fn()`` (one kernel launch); records HIP events around it when a KernelProfiler is installed."""
    if _profiler is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    _profiler.records.append((name, float(flops), float(nbytes), e0, e1))
    return r


def dtype_code(dt):
    if dt == torch.float32:
        return PP_F32
    if dt == torch.float16:
        return PP_F16
    raise TypeError(f"unsupported dtype {dt} (libpropainter_hip supports float32 and float16)")


def _stream(t):
    """The stream the launch goes to: torch's current stream of the device that owns the tensors.  A launch must be
    issued with that device current (HIP binds a launch to the current device): the module entry points wrap their work
    in ``torch.cuda.device(x.device)``; anything else fails loudly here instead of launching on another GPU's stream."""
    dev = t.device
    if dev.type != "cuda":
        raise RuntimeError("libpropainter_hip kernels need GPU tensors: there is no CPU fallback in the product path")
    if dev.index != torch.cuda.current_device():
        raise RuntimeError(f"tensor on {dev} but the current device is cuda:{torch.cuda.current_device()}: wrap the call in "
                           "torch.cuda.device(tensor.device) (the drop-in modules do)")
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def debug_cache_fence(device, mode=3):
    """[diagnosis] one small kernel on the CURRENT stream of `device` that writes back (mode & 1) and / or invalidates (mode & 2) the L2 of
    every XCD (csrc/api.hip: pp_debug_cache_fence).  Used by sharding.StreamingClipGraph under PP_SG_FENCE=1 only."""
    _check(lib().pp_debug_cache_fence(C.c_int(int(mode)), C.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "pp_debug_cache_fence")


_side_streams = {}


def side_streams(device, n, key=None):
    """n side streams of `device` (created once, outside any graph capture).  ``key`` (optional, hashable): a private set of lanes
    for one owner -- the logical ranks of a streaming pass captured into ONE hipGraph each get their own, so that two ranks' windows
    are not chained through a shared lane (a false dependency in the captured graph)."""
    if n < 2:
        return []
    key = (str(device), n, key)
    if key not in _side_streams:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("side streams must be created by an eager pass before graph capture")
        _side_streams[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _side_streams[key]


def fork_join(device, thunks, n_streams):
    """Runs independent pieces of GPU work (`thunks`: callables returning a tensor or a tuple / list of tensors) on up to
    n_streams side streams forked from and joined to the current stream -- under hipGraph capture this records parallel
    branches.  Same kernels on the same data: results are identical to running the thunks one after another.  The inputs a
    thunk reads must stay referenced by the caller until this function returns (they belong to the current stream's pool)."""
    device = torch.device(device)
    lanes = side_streams(device, min(n_streams, len(thunks))) if device.type == "cuda" else []
    if len(lanes) < 2:
        return [th() for th in thunks]
    cur = torch.cuda.current_stream(device)
    results = [None] * len(thunks)
    for i0 in range(0, len(thunks), len(lanes)):
        group = []
        for s_, k in zip(lanes, range(i0, min(len(thunks), i0 + len(lanes)))):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                results[k] = thunks[k]()
            group.append((k, s_))
        for k, s_ in group:
            cur.wait_stream(s_)
            r = results[k]
            for t in (r if isinstance(r, (tuple, list)) else (r,)):
                if torch.is_tensor(t):
                    t.record_stream(cur)
    return results


def on_device_of(t):
    """Context manager: makes the device of `t` current for the launches inside (no-op when it already is)."""
    return torch.cuda.device(t.device)


def on_input_device(fn):
    """Decorator for module entry points: runs the method with the device of its first tensor argument current, so that
    ``RAFT_bi(path, 'cuda:1')(frames_on_cuda1)`` launches on GPU 1 whatever the caller's current device is."""
    import functools

    def first_tensor(objs):
        for o in objs:
            if torch.is_tensor(o):
                return o
            if isinstance(o, (tuple, list)):
                t = first_tensor(o)
                if t is not None:
                    return t
        return None

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        t = first_tensor(args) if args else first_tensor(list(kwargs.values()))
        if t is None or not t.is_cuda:
            return fn(self, *args, **kwargs)
        with torch.cuda.device(t.device):
            return fn(self, *args, **kwargs)
    return wrapper


def _p(t, write=False):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libpropainter_hip kernels need GPU tensors: there is no CPU fallback in the product path")
    if _hazard.active() is not None:
        _hazard.active().note(t, write)
    return C.c_void_p(t.data_ptr())


def _pw(t):
    """pointer of a tensor the launch WRITES (same as _p; the hazard checker tells reads from writes by it)"""
    return _p(t, True)


def _i(v):
    return C.c_int(int(v))


def require_gpu(t, who):
    """The product path has no CPU implementation: modules call this before touching the engine."""
    if not t.is_cuda:
        raise RuntimeError(f"{who} runs on the HIP engine only (no CPU path in the product); got a {t.device} tensor")


# ----------------------------------------------------------------------------------------------
# host helpers (no GPU needed)
# ----------------------------------------------------------------------------------------------
def build_ktable(taps, src_channels, dcn_groups=0):
    """taps: list of (dy, dx); src_channels: padded (multiple-of-8) channel count per source.
    Returns int32 numpy array [kchunks + 1, 4] (last row = 16 zero bytes)."""
    L = lib()
    n = len(taps)
    dy = (C.c_int32 * n)(*[int(t[0]) for t in taps])
    dx = (C.c_int32 * n)(*[int(t[1]) for t in taps])
    sc = (C.c_int32 * len(src_channels))(*[int(c) for c in src_channels])
    size = L.pp_conv_build_ktable(n, dy, dx, len(src_channels), sc, int(dcn_groups), None, 0)
    if size < 0:
        _check(size, "pp_conv_build_ktable")
    out = np.zeros((size + 1, 4), dtype=np.int32)      # + the trailing all-zero entry (the kernels' zero page)
    rc = L.pp_conv_build_ktable(n, dy, dx, len(src_channels), sc, int(dcn_groups),
                                out.ctypes.data_as(C.POINTER(C.c_int32)), size + 1)
    if rc < 0:
        _check(rc, "pp_conv_build_ktable")
    return out


def window_tables(Hp, Wp, wh=5, ww=9):
    """Returns (own [nW, wh*ww], rolled [nW, n_rolled]) int32 numpy arrays."""
    L = lib()
    n_rolled = L.pp_window_tables(Hp, Wp, wh, ww, None, None, 0)
    if n_rolled < 0:
        _check(n_rolled, "pp_window_tables")
    nW = (Hp // wh) * (Wp // ww)
    own = np.zeros((nW, wh * ww), dtype=np.int32)
    rolled = np.zeros((nW, n_rolled), dtype=np.int32)
    rc = L.pp_window_tables(Hp, Wp, wh, ww, own.ctypes.data_as(C.POINTER(C.c_int32)),
                            rolled.ctypes.data_as(C.POINTER(C.c_int32)), rolled.size)
    if rc < 0:
        _check(rc, "pp_window_tables")
    return own, rolled


# ----------------------------------------------------------------------------------------------
# device ops
def _note_conv(rec, a):
    """hazard checker: the byte extents a pp_conv2d launch reads / writes, from the raw pointers and strides of its ConvArgs"""
    esz = 2 if a.dtype == PP_F16 else 4
    osz = 2 if a.out_dtype == PP_F16 else 4
    npix_in, npix_out = a.N * a.H * a.W, a.N * a.OH * a.OW
    for i in range(a.nsrc):
        if a.src[i].ptr:
            rec.note_range(a.src[i].ptr, npix_in * a.src[i].cstride * esz, False, True)
    for ptr, cs in ((a.residual, a.res_cstride), (a.preadd, a.preadd_cstride), (a.fuse_a, a.fuse_a_cstride), (a.fuse_b, a.fuse_b_cstride),
                    (a.dcn_offmask, a.dcn_cstride)):
        if ptr:
            rec.note_range(ptr, npix_out * cs * esz, False, True)
    whole = a.out_choff == 0 and a.out_cstride in (a.cout_g * a.groups, a.cout_pad * a.groups, 2 * a.cout_pad * a.groups)
    rec.note_range(a.out, npix_out * a.out_cstride * osz, True, not whole)
    if a.out2:
        rec.note_range(a.out2, npix_out * a.out2_cstride * osz, True, True)


# ----------------------------------------------------------------------------------------------
def conv2d_raw(args: ConvArgs, cin_read=None, on=None, split_k=0):
    """cin_read: channels read per input pixel (for the profiler's byte model; defaults to K per group x groups);
    on: a tensor of the launch (names the device / stream); split_k: the K table is a split-plane expansion (profiler accounting)."""
    if _stats is not None and args.dcn_offmask:
        args.dcn_stats = _stats.buf.data_ptr() + 8 * 4
    if _hazard.active() is not None:
        _note_conv(_hazard.active(), args)
    if _profiler is None:
        _check(lib().pp_conv2d(C.byref(args), _stream(on)), "pp_conv2d")
        return
    # algorithmic work of this launch: 2*M*Cout*K FLOPs; bytes = sources read once + weights + output written once
    M = args.N * args.OH * args.OW
    K = args.kchunks * 8
    esz = 2 if args.dtype == PP_F16 else 4
    osz = 2 if args.out_dtype == PP_F16 else 4
    # executed_k_mult = 3 for a split-plane ("f16x3") layer: its K table walks every block three times (hi*W_hi, lo*W_hi, hi*W_lo);
    # the ALGORITHMIC work is that of the fp32 layer it stands for: K / 3 products of 4-byte values
    if split_k:            # 3: blocks walked three times; 2: tri-product format (both planes of a block in one K step)
        K //= split_k
        esz = 4
        osz = 4
    flops = 2.0 * M * args.cout_g * K * args.groups
    cin = cin_read if cin_read is not None else K * args.groups
    nbytes = args.N * args.H * args.W * cin * esz + args.groups * args.cout_pad * K * esz + M * args.cout_g * args.groups * osz
    if args.dcn_offmask:
        nbytes += M * 432 * esz
    # operands of the fused epilogue are algorithmic traffic too (each read / written once per output element): residual or pre-activation
    # addend, the GRU state h (z | r gate: its r half; h gate: all couts) and gate z, the second output r * h -- without them the byte model of
    # the fused SepConvGRU layers was ~40 % short and their measured HBM traffic looked like re-reads
    if args.residual:
        nbytes += M * args.cout_g * args.groups * esz
    if args.preadd:
        nbytes += M * args.cout_g * esz
    if args.fuse == FUSE_GRU_ZR:
        nbytes += 2 * M * (args.cout_g - args.fuse_split) * esz              # h read, r * h written
    elif args.fuse == FUSE_GRU_H:
        nbytes += 2 * M * args.cout_g * esz                                  # h and z read
    name = "conv_gemm_dcn" if args.dcn_offmask else ("conv_gemm_f16x3" if split_k else "conv_gemm_f16" if args.dtype == PP_F16 else "conv_gemm_f32")
    if _profiler is not None and getattr(_profiler, "detail", False):      # per-layer-shape classes (bench.py --detail)
        name += f" | taps{args.tap_h}x{args.tap_w} s{args.stride_h} K{K} cout{args.cout_g}x{args.groups} M{M} {args.H}x{args.W}"
    timed(name, flops, nbytes, lambda: _check(lib().pp_conv2d(C.byref(args), _stream(on)), "pp_conv2d"))


def flow_warp(x, flow, out=None, mode="bilinear", x_choff=0, C_=None, fl_choff=0, out_choff=0):
    """x [N,H,W,Cx] NHWC, flow [N,H,W,>=2] NHWC; warps channels [x_choff, x_choff+C_)."""
    N, H, W, Cx = x.shape
    C_ = Cx if C_ is None else C_
    if out is None:
        out = torch.empty((N, H, W, C_), dtype=x.dtype, device=x.device)
    assert x.is_contiguous() and flow.is_contiguous() and out.is_contiguous() and flow.dtype == x.dtype
    timed("flow_warp", 0, _nbytes(out) * 2 + _nbytes(flow), lambda: _check(lib().pp_flow_warp(_p(x), _i(Cx), _i(x_choff), _p(flow), _i(flow.shape[-1]), _i(fl_choff), _pw(out),
                              _i(out.shape[-1]), _i(out_choff), _i(N), _i(H), _i(W), _i(C_),
                              _i(1 if mode == "nearest" else 0), _i(dtype_code(x.dtype)), _stream(x)),
           "pp_flow_warp"))
    return out


def fb_check(flow_fw, flow_bw, out=None, out_choff=0):
    """flows NHWC [N,H,W,2(+)]; returns/updates `out` NHWC with the validity map in channel out_choff."""
    N, H, W, _ = flow_fw.shape
    if out is None:
        out = torch.empty((N, H, W, 1), dtype=flow_fw.dtype, device=flow_fw.device)
    assert flow_fw.is_contiguous() and flow_bw.is_contiguous() and out.is_contiguous()
    timed("fb_check", 0, _nbytes(flow_fw) * 3, lambda: _check(lib().pp_fb_check(_p(flow_fw), _i(flow_fw.shape[-1]), _p(flow_bw), _i(flow_bw.shape[-1]), _pw(out),
                             _i(out.shape[-1]), _i(out_choff), _i(N), _i(H), _i(W), _i(dtype_code(flow_fw.dtype)),
                             _stream(flow_fw)),
           "pp_fb_check"))
    return out


def img_prop_step(x_prop, m_prop, x_cur, m_cur, flow_prop, flow_check, x_out, m_out, mode="nearest"):
    """planar NCHW tensors: x [N,C,H,W], m [N,1,H,W], flows [N,2,H,W]."""
    N, Cc, H, W = x_cur.shape
    for t in (x_prop, m_prop, x_cur, m_cur, flow_prop, flow_check, x_out, m_out):
        assert t.is_contiguous() and t.dtype == x_cur.dtype
    timed("img_prop_step", 0, _nbytes(x_prop) + _nbytes(m_prop) + _nbytes(x_cur) + _nbytes(m_cur) + _nbytes(flow_prop) + _nbytes(flow_check) + _nbytes(x_out) + _nbytes(m_out), lambda: _check(lib().pp_img_prop_step(_p(x_prop), _p(m_prop), _p(x_cur), _p(m_cur), _p(flow_prop), _p(flow_check),
                                  _pw(x_out), _pw(m_out), _i(N), _i(Cc), _i(H), _i(W),
                                  _i(1 if mode == "nearest" else 0), _i(dtype_code(x_cur.dtype)), _stream(x_cur)),
           "pp_img_prop_step"))


def binary_dilate(mask_u8, iterations):
    """uint8 [N,H,W] (non-zero = hole) on the GPU -> uint8 {0,255}, == scipy.ndimage.binary_dilation(mask, iterations) * 255."""
    N, H, W = mask_u8.shape
    assert mask_u8.dtype == torch.uint8 and mask_u8.is_contiguous()
    out = torch.empty_like(mask_u8)
    timed("binary_dilate", 0, 2 * mask_u8.numel(), lambda: _check(lib().pp_binary_dilate(_p(mask_u8), _pw(out), _i(N), _i(H), _i(W),
                                                                                       _i(iterations), _stream(mask_u8)), "pp_binary_dilate"))
    return out


def resize_bilinear_u8(x_u8, size):
    """uint8 NHWC [N,H,W,C] on the GPU -> [N,OH,OW,C] for size = (OW, OH): cv2.resize(f, size) (INTER_LINEAR) of every frame, with OpenCV's
    fixed-point 8-bit arithmetic (pp_resize_bilinear_u8; byte-identical to video_io.resize_u8_linear)."""
    N, H, W, Cc = x_u8.shape
    OW, OH = int(size[0]), int(size[1])
    assert x_u8.dtype == torch.uint8 and x_u8.is_contiguous()
    if (OW, OH) == (W, H):
        return x_u8
    out = torch.empty((N, OH, OW, Cc), dtype=torch.uint8, device=x_u8.device)
    timed("resize_bilinear_u8", 0, x_u8.numel() + out.numel(),
          lambda: _check(lib().pp_resize_bilinear_u8(_p(x_u8), _pw(out), _i(N), _i(H), _i(W), _i(Cc), _i(OH), _i(OW), _stream(x_u8)),
                         "pp_resize_bilinear_u8"))
    return out


def composite_window(pred, mask_u8, ori_u8, comp_u8, frame_ids, blend_flags):
    """pred [n,3,H,W] in [-1,1] (fp32 / fp16); mask_u8 [L,H,W(,1)] uint8 (non-zero = hole); ori_u8 / comp_u8 uint8 [L,H,W,3]; frame_ids: the clip
    frames of the n local frames; blend_flags[i]: frame i was composited before (0.5 / 0.5 blend).  Updates comp_u8 in place."""
    n, c3, H, W = pred.shape
    assert c3 == 3 and n <= 32 and pred.is_contiguous() and mask_u8.is_contiguous() and ori_u8.is_contiguous() and comp_u8.is_contiguous()
    assert mask_u8.dtype == ori_u8.dtype == comp_u8.dtype == torch.uint8 and len(frame_ids) == n == len(blend_flags)
    ids = (C.c_int32 * n)(*[int(i) for i in frame_ids])
    bits = sum(1 << i for i, b in enumerate(blend_flags) if b)
    timed("composite_window", 0, _nbytes(pred) + n * H * W * 10,
          lambda: _check(lib().pp_composite_window(_p(pred), _i(dtype_code(pred.dtype)), _p(mask_u8), _i(1), _p(ori_u8), _pw(comp_u8), ids, C.c_uint32(bits),
                                                   _i(n), _i(H), _i(W), _stream(pred)), "pp_composite_window"))
    return comp_u8


def corr_avgpool(x, M, H, W):
    out = torch.empty((M, H // 2, W // 2), dtype=torch.float32, device=x.device)
    timed("corr_avgpool", 0, _nbytes(out) * 5, lambda: _check(lib().pp_corr_avgpool(_p(x), _pw(out), C.c_int64(M), _i(H), _i(W), _stream(x)),
           "pp_corr_avgpool"))
    return out


def corr_lookup(levels, coords, out, split=False):
    """levels: 4 fp32 tensors [B*h*w, Hl, Wl]; coords fp32 [B,h,w,2]; out NHWC [B,h,w,Cpad>=324] (split: fp16 [B,h,w,2*Cpad], hi | lo planes)."""
    B, h, w, _ = coords.shape
    assert coords.dtype == torch.float32 and coords.is_contiguous() and out.is_contiguous()
    cs = out.shape[-1]
    timed("corr_lookup", 0, B * h * w * 4 * 100 * 4 + _nbytes(out), lambda: _check(lib().pp_corr_lookup(_p(levels[0]), _p(levels[1]), _p(levels[2]), _p(levels[3]), _p(coords), _pw(out),
                                _i(cs), _i(cs // 2 if split else cs), _i(B), _i(h), _i(w), _i(PP_F16S if split else dtype_code(out.dtype)),
                                _stream(coords)),
           "pp_corr_lookup"))
    return out


def corr_feature_pyramid(f2):
    """f2 fp16 NHWC [P,h,w,256] -> [f2, level1, level2, level3]: the avg-pooled FEATURE pyramid whose correlation with
    f1 equals the avg-pooled correlation pyramid of RAFT/corr.py:21-27 (pooling is linear)."""
    P, h, w, c = f2.shape
    assert f2.dtype == torch.float16 and c == 256 and f2.is_contiguous()
    lv = [torch.empty((P, h >> l, w >> l, 256), dtype=f2.dtype, device=f2.device) for l in (1, 2, 3)]
    timed("corr_feature_pyramid", 0, _nbytes(f2) * 3 + sum(_nbytes(t) for t in lv),
          lambda: _check(lib().pp_corr_feature_pyramid(_p(f2), _pw(lv[0]), _pw(lv[1]), _pw(lv[2]), _i(P), _i(h), _i(w), _stream(f2)),
                         "pp_corr_feature_pyramid"))
    return [f2] + lv


def corr_feature_pyramid_split(f2):
    """f2 split-plane fp16 NHWC [P,h,w,512] (256 hi | 256 lo) -> [level1, level2, level3], split-plane [P, h>>l, w>>l, 512]: the fp32 means
    over the 2^l x 2^l blocks (pp_corr_feature_pyramid_split).  f1 . level_l = level l of the avg-pooled correlation pyramid."""
    P, h, w, c = f2.shape
    assert f2.dtype == torch.float16 and c == 512 and f2.is_contiguous()
    lv = [torch.empty((P, h >> l, w >> l, 512), dtype=f2.dtype, device=f2.device) for l in (1, 2, 3)]
    timed("corr_feature_pyramid", 0, _nbytes(f2) * 3 + sum(_nbytes(t) for t in lv),
          lambda: _check(lib().pp_corr_feature_pyramid_split(_p(f2), _pw(lv[0]), _pw(lv[1]), _pw(lv[2]), _i(P), _i(h), _i(w), _stream(f2)),
                         "pp_corr_feature_pyramid_split"))
    return lv


def corr_lookup_otf(f1, f2_levels, coords, out):
    """Correlation lookup without the all-pairs volume (fp16): f1 [P,h,w,256], f2_levels from corr_feature_pyramid,
    coords fp32 [P,h,w,2]; writes out NHWC [P,h,w,328] (channels l*81 + a*9 + b as pp_corr_lookup, 324.. zero)."""
    P, h, w, _ = coords.shape
    assert coords.dtype == torch.float32 and coords.is_contiguous() and out.is_contiguous() and out.dtype == torch.float16
    assert f1.dtype == torch.float16 and f1.is_contiguous() and f1.shape == (P, h, w, 256) and all(t.is_contiguous() for t in f2_levels)
    npix = P * h * w
    timed("corr_lookup_otf", 2.0 * 256 * 400 * npix, _nbytes(f1) + sum(_nbytes(t) for t in f2_levels) + _nbytes(coords) + _nbytes(out),
          lambda: _check(lib().pp_corr_lookup_otf_stats(_p(f1), _p(f2_levels[0]), _p(f2_levels[1]), _p(f2_levels[2]), _p(f2_levels[3]),
                                                        _p(coords), _pw(out), _i(out.shape[-1]), _i(out.shape[-1]), _i(P), _i(h), _i(w),
                                                        _stats_ptr(0), _stream(f1)), "pp_corr_lookup_otf"))
    return out


OTF_SPLIT_LEVEL_CHANNELS = 88       # channels per pyramid level in pp_corr_lookup_otf_split's output planes (81 taps + 7 zeros)


def corr_lookup_otf_split(f1, f2_levels, coords, out):
    """Volume-free correlation lookup at fp32-class precision: f1 split-plane [P,h,w,512], f2_levels = [f2] + corr_feature_pyramid_split(f2),
    coords fp32 [P,h,w,2]; writes out split-plane fp16 [P,h,w,704]: level l, tap (a, b) at channel l*88 + a*9 + b of each 352-channel plane."""
    P, h, w, _ = coords.shape
    assert coords.dtype == torch.float32 and coords.is_contiguous() and out.is_contiguous() and out.dtype == torch.float16
    assert out.shape == (P, h, w, 8 * OTF_SPLIT_LEVEL_CHANNELS), out.shape
    assert f1.dtype == torch.float16 and f1.is_contiguous() and f1.shape == (P, h, w, 512)
    assert len(f2_levels) == 4 and all(t.is_contiguous() and t.dtype == torch.float16 and t.shape == (P, h >> l, w >> l, 512) for l, t in enumerate(f2_levels))
    npix = P * h * w
    # FLOPs: ~400 positions per pixel x 256 channels (the algorithmic dot products of the 4 x (10 x 10) neighbourhoods), three fp16 products each
    timed("corr_lookup_otf_split", 2.0 * 256 * 400 * npix, _nbytes(f1) + sum(_nbytes(t) for t in f2_levels) + _nbytes(coords) + _nbytes(out),
          lambda: _check(lib().pp_corr_lookup_otf_split_stats(_p(f1), _p(f2_levels[0]), _p(f2_levels[1]), _p(f2_levels[2]), _p(f2_levels[3]),
                                                              _p(coords), _pw(out), _i(out.shape[-1]), _i(P), _i(h), _i(w), _stats_ptr(0),
                                                              _stream(f1)),
                         "pp_corr_lookup_otf_split"))
    return out


def raft_flow_taps(coords1, coords0, rows, flow_out=None, flow_choff=0, split=False):
    """rows[P,h,w,16] <- the 7 horizontal taps of flow = coords1 - coords0 (fp32 [P,h,w,2]) per pixel, channel 2*kx + c; optionally
    flow_out[..., flow_choff:flow_choff+2] <- flow (pp_raft_flow_taps).  split: rows fp16 [P,h,w,32] = 16 hi | 16 lo, flow_out split-plane."""
    P, h, w, _ = coords1.shape
    assert coords1.dtype == coords0.dtype == torch.float32 and coords1.is_contiguous() and coords0.is_contiguous()
    assert rows.shape == (P, h, w, 32 if split else 16) and rows.is_contiguous() and (flow_out is None or (flow_out.is_contiguous() and flow_out.dtype == rows.dtype))
    assert not split or rows.dtype == torch.float16
    timed("raft_flow_taps", 0, 2 * _nbytes(coords1) + _nbytes(rows),
          lambda: _check(lib().pp_raft_flow_taps(_p(coords1), _p(coords0), _pw(rows), _pw(flow_out), _i(flow_out.shape[-1] if flow_out is not None else 0),
                                                 _i(flow_choff), _i(P), _i(h), _i(w), _i(PP_F16S if split else dtype_code(rows.dtype)), _stream(coords1)),
                         "pp_raft_flow_taps"))
    return rows


def convex_upsample(flow, mask):
    """flow fp32 [B,h,w,2]; mask NHWC [B,h,w,576(+)] -> fp32 [B,2,8h,8w]."""
    B, h, w, _ = flow.shape
    out = torch.empty((B, 2, 8 * h, 8 * w), dtype=torch.float32, device=flow.device)
    assert flow.dtype == torch.float32 and flow.is_contiguous() and mask.is_contiguous()
    timed("convex_upsample", 0, _nbytes(flow) + _nbytes(mask) + _nbytes(out), lambda: _check(lib().pp_convex_upsample(_p(flow), _p(mask), _i(mask.shape[-1]), _i(dtype_code(mask.dtype)), _pw(out),
                                    _i(B), _i(h), _i(w), _stream(flow)),
           "pp_convex_upsample"))
    return out


def window_mask(mask, wh=5, ww=9):
    """mask [B,Lt,Hp,Wp] -> fp32 [B, nW]."""
    B, Lt, Hp, Wp = mask.shape
    out = torch.empty((B, (Hp // wh) * (Wp // ww)), dtype=torch.float32, device=mask.device)
    assert mask.is_contiguous()
    timed("window_mask", 0, _nbytes(mask) + _nbytes(out), lambda: _check(lib().pp_window_mask(_p(mask), _pw(out), _i(B), _i(Lt), _i(Hp), _i(Wp), _i(wh), _i(ww),
                                _i(dtype_code(mask.dtype)), _stream(mask)),
           "pp_window_mask"))
    return out


def sparse_window_attention(q, k, v, pk, pv, own, rolled, tind, wmask, heads=4, wh=5, ww=9, qkv_cstride=None,
                            pkv_cstride=None, C_=None, impl=0, out_hw=None):
    """q/k/v: [B,T,Hp,Wp,*] token grids (possibly channel windows of one fused buffer: pass cstride);
    pk/pv: [B,T,P,*]; returns [B,T,Hp,Wp,C], or the cropped [B,T,oh,ow,C] when out_hw=(oh, ow) (padding tokens are not stored)."""
    B, T, Hp, Wp = q.shape[:4]
    C_ = q.shape[-1] if C_ is None else C_
    a = AttnArgs()
    a.dtype = dtype_code(q.dtype)
    a.B, a.T, a.Hp, a.Wp, a.C, a.heads, a.wh, a.ww = B, T, Hp, Wp, C_, heads, wh, ww
    a.n_rolled = rolled.shape[1]
    a.P = pk.shape[2] if pk is not None else 0
    a.n_tind = tind.numel()
    a.q, a.k, a.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    a.qkv_cstride = qkv_cstride if qkv_cstride is not None else q.shape[-1]
    a.pk = pk.data_ptr() if pk is not None else None
    a.pv = pv.data_ptr() if pv is not None else None
    a.pkv_cstride = pkv_cstride if pkv_cstride is not None else (pk.shape[-1] if pk is not None else 0)
    a.own, a.rolled, a.tind, a.wmask = own.data_ptr(), rolled.data_ptr(), tind.data_ptr(), wmask.data_ptr()
    if out_hw is not None:
        a.out_h, a.out_w = int(out_hw[0]), int(out_hw[1])
        out = torch.empty((B, T, a.out_h, a.out_w, C_), dtype=q.dtype, device=q.device)
    else:
        out = torch.empty((B, T, Hp, Wp, C_), dtype=q.dtype, device=q.device)
    a.out = out.data_ptr()
    a.impl = impl
    work = torch.empty((1 + B * wmask.shape[-1],), dtype=torch.int32, device=q.device)   # compacted masked-window list
    a.work, a.work_ints = work.data_ptr(), work.numel()
    if _hazard.active() is not None:
        for t in (q, k, v, pk, pv, own, rolled, tind, wmask):
            _hazard.active().note(t)
        _hazard.active().note(out, True)
        _hazard.active().note(work, True)
    for t in (q, k, v, own, rolled, tind, wmask):
        if not t.is_cuda:
            raise RuntimeError("sparse_window_attention needs GPU tensors")
    assert own.dtype == torch.int32 and rolled.dtype == torch.int32 and tind.dtype == torch.int32
    if _profiler is None:
        _check(lib().pp_sparse_window_attention(C.byref(a), _stream(q)), "pp_sparse_window_attention")
        return out
    # algorithmic work (profiling only; reads the window flags back): QK^T + PV = 4 FLOP per (query, key, channel)
    nmask = int((wmask > 0).sum().item())
    nW = wmask.numel()
    wsz = wh * ww
    keys_m = a.n_tind * (wsz + a.n_rolled + a.P)
    flops = 4.0 * C_ * (nmask * (T * wsz) * keys_m + (nW - nmask) * T * wsz * wsz)
    esz = q.element_size()
    nbytes = (3 * B * T * Hp * Wp * C_ + 2 * B * T * a.P * C_ + B * T * Hp * Wp * C_) * esz
    timed("sparse_window_attention", flops, nbytes,
          lambda: _check(lib().pp_sparse_window_attention(C.byref(a), _stream(q)), "pp_sparse_window_attention"))
    return out


def fold_tokens(tokens, BT, fh, fw, Cc, H, W, normalize=False, act=ACT_NONE):
    """tokens [BT, fh*fw, Cc*49] -> NHWC [BT,H,W,Cc]."""
    out = torch.empty((BT, H, W, Cc), dtype=tokens.dtype, device=tokens.device)
    assert tokens.is_contiguous()
    timed("fold_tokens", 0, _nbytes(tokens) + _nbytes(out), lambda: _check(lib().pp_fold_tokens(_p(tokens), _pw(out), _i(BT), _i(fh), _i(fw), _i(Cc), _i(H), _i(W),
                                _i(1 if normalize else 0), _i(act), _i(dtype_code(tokens.dtype)), _stream(tokens)),
           "pp_fold_tokens"))
    return out


def layernorm(x, gamma, beta, eps=1e-5):
    """x [..., C] contiguous; gamma/beta fp32."""
    Cc = x.shape[-1]
    out = torch.empty_like(x)
    assert x.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    timed("layernorm", 0, _nbytes(x) + _nbytes(out), lambda: _check(lib().pp_layernorm(_p(x), _p(gamma), _p(beta), _pw(out), C.c_int64(x.numel() // Cc), _i(Cc),
                              C.c_float(eps), _i(dtype_code(x.dtype)), _stream(x)),
           "pp_layernorm"))
    return out


def layernorm_grid(x, gamma, beta, out, eps=1e-5):
    """LayerNorm of a token grid x [N,gh,gw,C] written into the top-left corner of the padded grid out [N,Hp,Wp,C] (padding tokens of
    `out` are left as they are: zero-filled once by the caller)."""
    N, gh, gw, Cc = x.shape
    assert out.shape[0] == N and out.shape[3] == Cc and out.dtype == x.dtype and x.is_contiguous() and out.is_contiguous()
    timed("layernorm", 0, 2 * _nbytes(x), lambda: _check(lib().pp_layernorm_grid(_p(x), _p(gamma), _p(beta), _pw(out), _i(N), _i(gh), _i(gw), _i(out.shape[1]),
                                                                                 _i(out.shape[2]), _i(Cc), C.c_float(eps), _i(dtype_code(x.dtype)), _stream(x)),
                                                 "pp_layernorm_grid"))
    return out


def depthwise_pool(x, weight, bias, k=4):
    """x NHWC [N,H,W,C]; weight fp32 [C,k,k]; bias fp32 [C]."""
    N, H, W, Cc = x.shape
    out = torch.empty((N, H // k, W // k, Cc), dtype=x.dtype, device=x.device)
    assert x.is_contiguous() and weight.dtype == torch.float32
    timed("depthwise_pool", 0, _nbytes(x) + _nbytes(out), lambda: _check(lib().pp_depthwise_pool(_p(x), _p(weight), _p(bias), _pw(out), _i(N), _i(H), _i(W), _i(Cc), _i(k),
                                   _i(dtype_code(x.dtype)), _stream(x)),
           "pp_depthwise_pool"))
    return out


def instance_norm(x, relu=False, eps=1e-5, out=None):
    N, H, W, Cc = x.shape
    out = torch.empty_like(x) if out is None else out
    L = lib()
    L.pp_instance_norm_workspace_floats.restype = C.c_int64
    ws = torch.empty((int(L.pp_instance_norm_workspace_floats(N, H, W, Cc)),), dtype=torch.float32, device=x.device)
    assert x.is_contiguous()
    timed("instance_norm", 0, _nbytes(x) * 2 + _nbytes(out), lambda: _check(lib().pp_instance_norm(_p(x), _pw(out), _pw(ws), _i(N), _i(H), _i(W), _i(Cc), C.c_float(eps),
                                  _i(1 if relu else 0), _i(dtype_code(x.dtype)), _stream(x)),
           "pp_instance_norm"))
    return out


def instance_norm_split(x, relu=False, eps=1e-5, residual=None, res_choff=0, relu2=False):
    """x fp32 NHWC [N,H,W,C] -> split-plane fp16 [N,H,W,2C]: relu2(relu(IN(x)) + residual) (pp_instance_norm_split; the tail of RAFT's
    ResidualBlock, RAFT/extractor.py:44-57).  residual: split-plane fp16 [N,H,W,2*Cr] window starting at res_choff, or None."""
    N, H, W, Cc = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty((N, H, W, 2 * Cc), dtype=torch.float16, device=x.device)
    L = lib()
    L.pp_instance_norm_workspace_floats.restype = C.c_int64
    ws = torch.empty((int(L.pp_instance_norm_workspace_floats(N, H, W, Cc)),), dtype=torch.float32, device=x.device)
    if residual is not None:
        assert residual.dtype == torch.float16 and residual.is_contiguous() and residual.shape[:3] == (N, H, W)
    timed("instance_norm", 0, _nbytes(x) * 2 + _nbytes(out) * (2 if residual is not None else 1),
          lambda: _check(lib().pp_instance_norm_split(_p(x), _pw(out), _pw(ws), _i(N), _i(H), _i(W), _i(Cc), C.c_float(eps), _i(1 if relu else 0),
                                                      _p(residual), _i(residual.shape[-1] if residual is not None else 0), _i(res_choff),
                                                      _i(1 if relu2 else 0), _stream(x)), "pp_instance_norm_split"))
    return out


def upsample2x(x):
    N, H, W, Cc = x.shape
    out = torch.empty((N, 2 * H, 2 * W, Cc), dtype=x.dtype, device=x.device)
    assert x.is_contiguous()
    timed("upsample2x", 0, _nbytes(x) + _nbytes(out), lambda: _check(lib().pp_upsample2x(_p(x), _pw(out), _i(N), _i(H), _i(W), _i(Cc), _i(dtype_code(x.dtype)), _stream(x)),
           "pp_upsample2x"))
    return out


def dcn_offset_mask_act(offmask, mag, flow=None, fl_choff=0):
    """in place on NHWC [N,H,W,432(+pad)]."""
    npix = offmask.numel() // offmask.shape[-1]
    assert offmask.is_contiguous() and (flow is None or (flow.is_contiguous() and flow.dtype == offmask.dtype))
    timed("dcn_offset_mask_act", 0, _nbytes(offmask) * 2, lambda: _check(lib().pp_dcn_offset_mask_act(_pw(offmask), _i(offmask.shape[-1]), _p(flow),
                                        _i(flow.shape[-1] if flow is not None else 0), _i(fl_choff), C.c_float(mag),
                                        C.c_int64(npix), _i(dtype_code(offmask.dtype)), _stream(offmask)),
           "pp_dcn_offset_mask_act"))
    return offmask


def gru_gate(zr, h, h_choff, Cc, out, out_choff, q=None):
    """mode 0 (q is None): out = r*h with r = zr[..., C:2C]; mode 1: out = (1-z)*h + z*q, z = zr[..., :C]."""
    npix = zr.numel() // zr.shape[-1]
    timed("gru_gate", 0, npix * Cc * 4 * zr.element_size(), lambda: _check(lib().pp_gru_gate(_p(zr), _i(zr.shape[-1]), _p(h), _i(h.shape[-1]), _i(h_choff), _p(q),
                             _i(q.shape[-1] if q is not None else 0), _pw(out), _i(out.shape[-1]), _i(out_choff),
                             C.c_int64(npix), _i(Cc), _i(0 if q is None else 1), _i(dtype_code(zr.dtype)), _stream(zr)),
           "pp_gru_gate"))
    return out


def nchw_to_nhwc(x, out=None, out_choff=0, out_dtype=None, cpad=None, scale=1.0, split=False):
    """x planar [N,C,H,W] -> NHWC window of `out` (allocated zero-filled [N,H,W,cpad] if None).  split: fp32 input -> split-plane fp16
    [N,H,W,2*cpad] (hi | lo planes)."""
    N, Cc, H, W = x.shape
    if out is None:
        cpad = cpad or ((Cc + 7) // 8 * 8)
        dt = torch.float16 if split else (out_dtype or x.dtype)
        out = (torch.zeros if cpad != Cc else torch.empty)((N, H, W, 2 * cpad if split else cpad), dtype=dt, device=x.device)
    assert x.is_contiguous() and out.is_contiguous() and (not split or (x.dtype == torch.float32 and out.dtype == torch.float16))
    timed("nchw_to_nhwc", 0, _nbytes(x) * 2, lambda: _check(lib().pp_nchw_to_nhwc(_p(x), _i(dtype_code(x.dtype)), _pw(out), _i(PP_F16S if split else dtype_code(out.dtype)),
                                 _i(out.shape[-1]), _i(out_choff), _i(N), _i(Cc), _i(H), _i(W), C.c_float(scale),
                                 _stream(x)),
           "pp_nchw_to_nhwc"))
    return out


def pack_nhwc8(srcs, out=None, split=False):
    """srcs: one to three planar [N,c_i,H,W] tensors of one dtype with sum(c_i) <= 8 -> NHWC [N,H,W,8], missing channels zero; one launch
    that writes whole 16-byte rows (the encoder input cat(frame, mask, updated mask), model/propainter.py:334-336).  split: fp32 sources ->
    split-plane fp16 [N,H,W,8 hi | 8 lo] (the RAFT encoders' input of the f16x3 engine), whole 32-byte rows."""
    x0 = srcs[0]
    N, _, H, W = x0.shape
    assert 1 <= len(srcs) <= 3 and all(t.is_contiguous() and t.dtype == x0.dtype and t.shape[0] == N and t.shape[2:] == x0.shape[2:] for t in srcs)
    cs = [t.shape[1] for t in srcs] + [0] * (3 - len(srcs))
    assert sum(cs) <= 8 and (not split or x0.dtype == torch.float32)
    oshape, odt = ((N, H, W, 16), torch.float16) if split else ((N, H, W, 8), x0.dtype)
    if out is None:
        out = torch.empty(oshape, dtype=odt, device=x0.device)
    assert out.is_contiguous() and out.dtype == odt and tuple(out.shape) == oshape
    ptrs = [_p(t) for t in srcs] + [None] * (3 - len(srcs))
    timed("nchw_to_nhwc", 0, sum(_nbytes(t) for t in srcs) + _nbytes(out),
          lambda: _check(lib().pp_pack_nhwc8(ptrs[0], _i(cs[0]), ptrs[1], _i(cs[1]), ptrs[2], _i(cs[2]), _pw(out), _i(N), _i(H), _i(W),
                                             _i(PP_F16S if split else dtype_code(x0.dtype)), _stream(x0)), "pp_pack_nhwc8"))
    return out


def nhwc_to_nchw(x, Cc=None, choff=0, out_dtype=None, act=ACT_NONE):
    N, H, W, Cs = x.shape
    Cc = Cs if Cc is None else Cc
    out = torch.empty((N, Cc, H, W), dtype=out_dtype or x.dtype, device=x.device)
    assert x.is_contiguous()
    timed("nhwc_to_nchw", 0, _nbytes(out) * 2, lambda: _check(lib().pp_nhwc_to_nchw(_p(x), _i(dtype_code(x.dtype)), _i(Cs), _i(choff), _pw(out),
                                 _i(dtype_code(out.dtype)), _i(N), _i(Cc), _i(H), _i(W), _i(act), _stream(x)),
           "pp_nhwc_to_nchw"))
    return out
