"""Runs the REFERENCE'S OWN driver script -- /root/reference/inference_propainter.py, unmodified, as ``__main__`` -- on CPU and
captures what it produces.  Test infrastructure (authoring container only: /root/reference does not travel to the GPU box).

Why: ``oracle.propainter_oracle.inpaint_video`` RESTATES the driver (inference_propainter.py:298-452: chunked RAFT, flow completion,
image propagation, the window schedule of ``get_ref_index``, the order-dependent uint8 blend) and so did the e2e golden; this module
pins that restatement -- and the mask pre-processing of ``read_mask`` (:77-115) -- to the script itself.

How the script runs here without its missing wheels (no source edits, nothing copied):
  * ``cv2``: a stub module with the five calls the script makes on this route -- ``imread`` / ``cvtColor(BGR2RGB)`` (PIL decode, channel
    flip), ``resize`` (identity when the size is unchanged, else OpenCV's fixed-point INTER_LINEAR as restated in
    ``propainter_amd.video_io.resize_u8_linear``), ``imwrite`` (PIL), plus the constants / no-op setters the imports touch;
  * ``imageio``: a stub whose ``mimwrite`` keeps the frame lists (``masked_in.mp4`` / ``inpaint_out.mp4``) instead of encoding them;
  * ``torchvision``: a stub with ``ops.deform_conv2d`` = ``oracle/deform_conv_ref.py`` (the un-vendored dependency, restated: parity
    unpinned at THAT boundary) and ``transforms.Compose`` (three lines: ``core/utils.py:94-95`` composes ``Stack`` / ``ToTorchFormatTensor``);
  * ``torch.__version__`` is a plain ``X.Y.Z`` while the script runs (``model/misc.py:56-57`` cannot parse ``2.10.0+rocm7.0``);
  * the working directory is a scratch folder whose ``weights/{raft-things,recurrent_flow_completion,ProPainter}.pth`` already exist, so
    ``utils/download_util.py:105-108`` returns the cached files; ``raft-things.pth`` carries the ``module.`` prefix (``flow_comp_raft.py:18-19``).

    python -m oracle.run_reference_driver --work DIR --out captured.npz -- -i DIR/clip -m DIR/clip_mask --subvideo_length 6 ...

``run_reference_main`` is the in-process form the module's ``__main__`` uses; ``reference_main_on_clip`` prepares the scratch folder
(frame / mask PNGs, checkpoint files) from arrays and state dicts, spawns this module and returns the captured arrays."""
import argparse
import os
import runpy
import subprocess
import sys
import tempfile
import types

import numpy as np

from .ref_shims import REFERENCE_ROOT, reference_available


def _stub_modules(captured):
    """cv2 / imageio / torchvision stand-ins for exactly the calls inference_propainter.py makes on the frame-folder route."""
    from PIL import Image
    from . import deform_conv_ref

    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2RGB, cv2.COLOR_RGB2BGR, cv2.INTER_LINEAR, cv2.INTER_CUBIC = 4, 4, 1, 2
    cv2.setNumThreads = lambda *a, **k: None
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda *a, **k: None)

    def imread(path):
        return np.ascontiguousarray(np.array(Image.open(path).convert("RGB"))[:, :, ::-1])       # BGR, as cv2.imread

    def cvtColor(img, code):
        assert code == cv2.COLOR_BGR2RGB, code
        return np.ascontiguousarray(img[:, :, ::-1])

    def resize(img, size, interpolation=None):
        if (img.shape[1], img.shape[0]) == tuple(size):
            return img
        assert interpolation in (None, cv2.INTER_LINEAR), "only INTER_LINEAR is restated (propainter_amd/video_io.py)"
        from propainter_amd.video_io import resize_u8_linear
        return resize_u8_linear(img, size)

    def imwrite(path, img, params=None):
        Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(path)
        return True

    cv2.imread, cv2.cvtColor, cv2.resize, cv2.imwrite = imread, cvtColor, resize, imwrite

    imageio = types.ModuleType("imageio")

    def mimwrite(path, frames, fps=None, quality=None, **kw):
        captured[os.path.basename(path)] = [np.asarray(f).copy() for f in frames]
        captured[os.path.basename(path) + ".fps"] = fps

    imageio.mimwrite = mimwrite

    tv = types.ModuleType("torchvision")
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.deform_conv2d = deform_conv_ref.deform_conv2d
    tv.transforms = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    tv.transforms.Compose = Compose
    tv.io = types.ModuleType("torchvision.io")
    return {"cv2": cv2, "imageio": imageio, "torchvision": tv, "torchvision.ops": tv.ops, "torchvision.transforms": tv.transforms,
            "torchvision.io": tv.io}


def run_reference_main(work, cli_args):
    """Executes the reference script as ``__main__`` with cwd = ``work`` and returns (its module globals, the captured videos)."""
    import torch
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    captured = {}
    stubs = _stub_modules(captured)
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    real_version, cwd, argv = torch.__version__, os.getcwd(), sys.argv
    torch.__version__ = real_version.split("+")[0]
    script = os.path.join(REFERENCE_ROOT, "inference_propainter.py")
    sys.path.insert(0, REFERENCE_ROOT)
    os.chdir(work)
    sys.argv = [script] + list(cli_args)
    try:
        g = runpy.run_path(script, run_name="__main__")
    finally:
        sys.argv = argv
        os.chdir(cwd)
        sys.path.remove(REFERENCE_ROOT)
        torch.__version__ = real_version
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return g, captured


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    rest = a.rest[1:] if a.rest[:1] == ["--"] else a.rest
    import torch
    torch.set_num_threads(int(os.environ.get("PP_REF_THREADS", os.cpu_count())))
    g, cap = run_reference_main(a.work, rest)
    comp = np.stack(cap["inpaint_out.mp4"])                      # what the script hands to the video writer (:470-472)
    out = {"comp": comp, "masked_in": np.stack(cap["masked_in.mp4"]), "fps": np.int64(cap["inpaint_out.mp4.fps"]),
           # stage results the script keeps in module globals: dilated masks (:246-252), completed flows (:355-368), image propagation (:373-404)
           "masks_dilated": (g["masks_dilated"][0, :, 0].cpu().numpy() * 255).astype(np.uint8),
           "flow_masks": (g["flow_masks"][0, :, 0].cpu().numpy() * 255).astype(np.uint8),
           "pred_f": g["pred_flows_bi"][0][0].float().cpu().numpy(), "pred_b": g["pred_flows_bi"][1][0].float().cpu().numpy(),
           "upd_masks": g["updated_masks"][0].float().cpu().numpy(), "upd_frames": g["updated_frames"][0].float().cpu().numpy(),
           "size": np.array(g["size"]), "out_size": np.array(g["out_size"])}
    np.savez_compressed(a.out, **out)


def reference_main_on_clip(frames_u8, mask_u8, sds, extra_args=(), threads=None, keep=None):
    """frames_u8 [L,H,W,3] RGB, mask_u8 [H,W] or [L,H,W] (UNDILATED, non-zero = hole), sds = {"raft","fc","gen"} plain-key state dicts.
    Writes the frame / mask PNG folders and the three checkpoint files, runs the reference script in a child process, returns the npz dict."""
    import torch
    from PIL import Image
    work = keep or tempfile.mkdtemp(prefix="pp_refmain_")
    os.makedirs(os.path.join(work, "weights"), exist_ok=True)
    os.makedirs(os.path.join(work, "clip"), exist_ok=True)
    os.makedirs(os.path.join(work, "clip_mask"), exist_ok=True)
    for i, f in enumerate(frames_u8):
        Image.fromarray(f).save(os.path.join(work, "clip", f"{i:05d}.png"))
    ms = mask_u8[None] if mask_u8.ndim == 2 else mask_u8
    for i, m in enumerate(ms):
        Image.fromarray(((m > 0) * 255).astype(np.uint8)).save(os.path.join(work, "clip_mask", f"{i:05d}.png"))
    torch.save({"module." + k: v for k, v in sds["raft"].items()}, os.path.join(work, "weights", "raft-things.pth"))
    torch.save(dict(sds["fc"]), os.path.join(work, "weights", "recurrent_flow_completion.pth"))
    torch.save(dict(sds["gen"]), os.path.join(work, "weights", "ProPainter.pth"))
    out = os.path.join(work, "captured.npz")
    env = dict(os.environ)
    if threads:
        env["PP_REF_THREADS"] = str(threads)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "oracle.run_reference_driver", "--work", work, "--out", out, "--",
           "-i", os.path.join(work, "clip"), "-m", os.path.join(work, "clip_mask"), "-o", os.path.join(work, "results")] + [str(x) for x in extra_args]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"reference driver failed ({r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")
    with np.load(out) as z:
        res = {k: z[k] for k in z.files}
    if keep is None:
        import shutil
        shutil.rmtree(work, ignore_errors=True)
    return res


def reference_modules(sds):
    """The REAL reference's modules (oracle/ref_shims.py) behind propainter_oracle.inpaint_video's ``modules=`` signatures: the restated
    DRIVER over the reference's own stages, for a byte-for-byte comparison with the script (the stages' restatements are pinned elsewhere)."""
    import torch
    from .ref_shims import build_reference_raft, load_reference
    ns = load_reference()
    raft = build_reference_raft()
    raft.load_state_dict(sds["raft"])
    fc = ns.RecurrentFlowCompleteNet().eval()
    fc.load_state_dict(sds["fc"])
    gen = ns.InpaintGenerator(init_weights=False).eval()
    gen.load_state_dict(sds["gen"])

    def raft_bi(frames, iters):                       # model/modules/flow_comp_raft.py:39-55
        b, l_t, c, h, w = frames.size()
        a, bq = frames[:, :-1].reshape(-1, c, h, w), frames[:, 1:].reshape(-1, c, h, w)
        _, ff = raft(a, bq, iters=iters, test_mode=True)
        _, fb = raft(bq, a, iters=iters, test_mode=True)
        return ff.view(b, l_t - 1, 2, h, w), fb.view(b, l_t - 1, 2, h, w)

    def image_propagation(frames, ff, fb, masks, interp):
        b, t, _, h, w = masks.size()
        pi, pm = gen.img_propagation(frames, (ff, fb), masks, interp)
        return pi.view(b, t, 3, h, w), pm.view(b, t, 1, h, w)

    return {"raft_bi": raft_bi,
            "fc_forward_bidirect": lambda flows, masks: fc.forward_bidirect_flow(flows, masks)[0],
            "fc_combine": fc.combine_flow,
            "image_propagation": image_propagation,
            "generator_forward": lambda fr, flows, m_in, m_upd, l_t: gen(fr, flows, m_in, m_upd, l_t)}


if __name__ == "__main__":
    main()
