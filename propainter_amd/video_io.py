"""Host-side pre/post-processing of the CLI (what inference_propainter.py:34-156,219-264,453-472 of the reference does
around the device path): frame / mask reading, resize to multiples of 8, mask dilation, outpainting canvases, result
writing.  PIL + numpy + scipy only -- cv2, imageio and torchvision.io are optional and probed at run time."""
import os

import numpy as np
import scipy.ndimage
from PIL import Image

VIDEO_EXT = ('mp4', 'mov', 'avi', 'MP4', 'MOV', 'AVI')
IMAGE_EXT = ('jpg', 'jpeg', 'png', 'JPG', 'JPEG', 'PNG')


def resize_frames(frames, size=None):
    """frames: list of PIL images; size (w, h) or None.  Processing size = size rounded down to multiples of 8
    (inference_propainter.py:37-48).  Returns (frames, process_size, out_size)."""
    out_size = tuple(size) if size is not None else frames[0].size
    process_size = (out_size[0] - out_size[0] % 8, out_size[1] - out_size[1] % 8)
    if size is not None or out_size != process_size:
        frames = [f.resize(process_size) for f in frames]
    return frames, process_size, out_size


def read_frames(path):
    """Image folder or video file -> (list of RGB PIL images, fps or None, (w, h), name) (:52-71)."""
    if path.endswith(VIDEO_EXT):
        name = os.path.basename(path)[:-4]
        frames, first_error = None, None
        try:
            import imageio.v2 as imageio
            rd = imageio.get_reader(path)
            fps = rd.get_meta_data().get('fps')
            frames = [Image.fromarray(np.asarray(f)[..., :3]) for f in rd]
        except Exception as e:      # noqa: BLE001 -- imageio missing, or importable without its ffmpeg plugin / binary (RuntimeError, IOError, ...)
            first_error, frames = e, None
            try:
                import torchvision
                v, _, info = torchvision.io.read_video(filename=path, pts_unit='sec')
                frames, fps = [Image.fromarray(f) for f in v.numpy()], info['video_fps']
            except Exception:       # noqa: BLE001 -- no torchvision, or one without a working video reader
                frames = None
        if not frames:
            # neither imageio(+ffmpeg) nor torchvision.io: the pure-Python path reads JPEG-coded .mp4 / .mov tracks (what save_results
            # writes in such an image, and what `ffmpeg -c:v mjpeg` writes); other codecs raise, naming the codec and the remedy
            from . import mp4_mjpeg
            try:
                frames, fps = mp4_mjpeg.read_mp4(path)
            except Exception:       # noqa: BLE001
                if first_error is not None and not isinstance(first_error, ImportError):
                    raise first_error         # a decoder WAS there and failed: its message is the useful one
                raise
    else:
        name = os.path.basename(os.path.normpath(path))
        files = sorted(f for f in os.listdir(path) if f.endswith(IMAGE_EXT))
        frames = [Image.open(os.path.join(path, f)).convert('RGB') for f in files]
        fps = None
    if not frames:
        raise RuntimeError(f"no frames found in {path}")
    return frames, fps, frames[0].size, name


def read_masks(path, length, size, flow_mask_dilates=8, mask_dilates=5, device=None):
    """Single mask image or folder of per-frame masks -> (flow_masks, masks_dilated): lists of uint8 {0,255} arrays
    [h,w] (:81-115: nearest resize, grey conversion, scipy binary dilation, or the 0.1 threshold when dilation is 0).
    With ``device`` (a GPU) the dilation runs in pp_binary_dilate (bit-identical to scipy, tests/test_ops_gpu.py)."""
    if path.endswith(IMAGE_EXT):
        imgs = [Image.open(path)]
    else:
        imgs = [Image.open(os.path.join(path, f)) for f in sorted(os.listdir(path)) if f.endswith(IMAGE_EXT)]
    if device is not None:
        import torch
        from . import hip
        raw = np.stack([np.array((im.resize(size, Image.NEAREST) if size is not None else im).convert('L')) for im in imgs])
        t = torch.from_numpy(np.ascontiguousarray(raw)).to(device)
        fm = hip.binary_dilate(t, flow_mask_dilates).cpu().numpy()
        md = fm if mask_dilates == flow_mask_dilates else hip.binary_dilate(t, mask_dilates).cpu().numpy()
        flow_masks, masks_dilated = list(fm), list(md)
        if len(imgs) == 1:
            flow_masks, masks_dilated = flow_masks * length, masks_dilated * length
        if len(flow_masks) < length:
            raise RuntimeError(f"{len(flow_masks)} masks for {length} frames")
        return flow_masks[:length], masks_dilated[:length]

    def dil(a, it):
        if it > 0:
            return scipy.ndimage.binary_dilation(a, iterations=it).astype(np.uint8) * 255
        return (a > 0.1).astype(np.uint8) * 255          # binary_mask(th=0.1) on a uint8 image == non-zero

    flow_masks, masks_dilated = [], []
    for im in imgs:
        if size is not None:
            im = im.resize(size, Image.NEAREST)
        a = np.array(im.convert('L'))
        flow_masks.append(dil(a, flow_mask_dilates))
        masks_dilated.append(dil(a, mask_dilates))
    if len(imgs) == 1:
        flow_masks, masks_dilated = flow_masks * length, masks_dilated * length
    if len(flow_masks) < length:
        raise RuntimeError(f"{len(flow_masks)} masks for {length} frames")
    return flow_masks[:length], masks_dilated[:length]


def extrapolation(frames, scale):
    """Outpainting canvas (:118-156): frames centred in a (scale_h, scale_w) larger field of view (multiples of 8);
    flow mask keeps a 4-px rim of known pixels out when the border is wider than 10 px."""
    n = len(frames)
    w, h = frames[0].size
    H = int(scale[0] * h); W = int(scale[1] * w)
    H -= H % 8; W -= W % 8
    y0, x0 = int((H - h) / 2), int((W - w) / 2)
    out = []
    for f in frames:
        c = np.zeros((H, W, 3), dtype=np.uint8)
        c[y0:y0 + h, x0:x0 + w] = np.asarray(f)
        out.append(Image.fromarray(c))
    dh, dw = (4 if y0 > 10 else 0), (4 if x0 > 10 else 0)
    fm = np.ones((H, W), dtype=np.uint8)
    fm[y0 + dh:y0 + h - dh, x0 + dw:x0 + w - dw] = 0
    md = np.ones((H, W), dtype=np.uint8)
    md[y0:y0 + h, x0:x0 + w] = 0
    return out, [fm * 255] * n, [md * 255] * n, (W, H)


def masked_preview(frames_u8, masks_dilated, alpha=0.6):
    """Green overlay of the masked region (:247-258)."""
    out = []
    for f, m in zip(frames_u8, masks_dilated):
        mk = (m[..., None] / 255.0)
        green = np.zeros_like(f, dtype=np.float64); green[..., 1] = 255
        fuse = (1 - alpha) * f + alpha * green
        out.append((mk * fuse + (1 - mk) * f).astype(np.uint8))
    return out


def _resize_u8(a, size, resample):
    return np.asarray(Image.fromarray(a).resize(size, resample)) if (a.shape[1], a.shape[0]) != tuple(size) else a


def _cv_linear_coeffs(n_src, n_dst):
    """source index pair and 11-bit integer weights of OpenCV's 8-bit INTER_LINEAR along one axis (see resize_u8_linear)"""
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * (n_src / n_dst) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo, hi = s < 0, s >= n_src - 1
    f[lo | hi] = 0
    s[lo] = 0
    s[hi] = n_src - 1
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, np.minimum(s + 1, n_src - 1), a0, a1


def resize_u8_linear(a, size):
    """cv2.resize(a, size, interpolation=cv2.INTER_LINEAR) for a uint8 image [H,W,C] / [H,W]: what the reference uses for its inputs
    (core/dataset.py:186-188: no antialiasing when shrinking, unlike PIL's BILINEAR) and for the saved videos
    (inference_propainter.py:469-470).  OpenCV (un-vendored, absent offline) resizes 8-bit images in fixed point -- 11-bit weights
    cvRound(w * 2048), integer horizontal pass, (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2 vertically; an exact 2 : 1
    reduction in both axes averages 2 x 2 blocks (its INTER_AREA shortcut).  Restated from the published algorithm; the device kernel
    pp_resize_bilinear_u8 computes the same bytes (tests/test_ops_gpu.py)."""
    a = np.asarray(a)
    H, W = a.shape[:2]
    OW, OH = int(size[0]), int(size[1])
    if (OW, OH) == (W, H):
        return a
    x = a.reshape(H, W, -1).astype(np.int64)
    if W == 2 * OW and H == 2 * OH:
        out = (x[0::2, 0::2] + x[0::2, 1::2] + x[1::2, 0::2] + x[1::2, 1::2] + 2) >> 2
    else:
        x0, x1, a0, a1 = _cv_linear_coeffs(W, OW)
        y0, y1, b0, b1 = _cv_linear_coeffs(H, OH)
        rows = x[:, x0] * a0[None, :, None] + x[:, x1] * a1[None, :, None]                 # [H, OW, C]
        out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8).reshape((OH, OW) + a.shape[2:])


def save_results(save_root, comp_frames, masked_frames, out_size, fps, save_frames, comp_video_frames=None):
    """results/<name>/{masked_in.mp4, inpaint_out.mp4, frames/%04d.png} (:453-472).  With imageio + ffmpeg the videos are H.264 as the
    reference's; without them (this image) the same files are written as Motion-JPEG .mp4 by the pure-Python muxer (mp4_mjpeg.py).
    The video frames are resized like cv2.resize(f, out_size) (INTER_LINEAR, :469-470): ``comp_video_frames`` = the composites already
    resized on the device (hip.resize_bilinear_u8), else ``resize_u8_linear`` here; the PNG frames use bicubic (:458)."""
    os.makedirs(save_root, exist_ok=True)
    wrote = []
    if save_frames:
        d = os.path.join(save_root, 'frames')
        os.makedirs(d, exist_ok=True)
        for i, f in enumerate(comp_frames):
            Image.fromarray(_resize_u8(f, out_size, Image.BICUBIC)).save(os.path.join(d, f"{i:04d}.png"))
        wrote.append(d)
    masked = [resize_u8_linear(f, out_size) for f in masked_frames]
    comp = list(comp_video_frames) if comp_video_frames is not None else [resize_u8_linear(f, out_size) for f in comp_frames]
    try:
        import imageio.v2 as imageio
        imageio.mimwrite(os.path.join(save_root, 'masked_in.mp4'), masked, fps=fps, quality=7)
        imageio.mimwrite(os.path.join(save_root, 'inpaint_out.mp4'), comp, fps=fps, quality=7)
    except Exception as e:   # no imageio / no ffmpeg in this image: the same file names as Motion-JPEG .mp4 (mp4_mjpeg.py; quality 7 -> JPEG 85)
        from . import mp4_mjpeg
        mp4_mjpeg.write_mp4(os.path.join(save_root, 'masked_in.mp4'), masked, fps=fps, quality=7)
        mp4_mjpeg.write_mp4(os.path.join(save_root, 'inpaint_out.mp4'), comp, fps=fps, quality=7)
        print(f"[propainter_amd] imageio / ffmpeg unavailable ({type(e).__name__}): wrote Motion-JPEG .mp4 files (propainter_amd/mp4_mjpeg.py)")
    wrote += [os.path.join(save_root, 'masked_in.mp4'), os.path.join(save_root, 'inpaint_out.mp4')]
    return wrote
