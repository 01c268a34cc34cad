"""``ProInpainter`` -- the library entry point the reference's web demos use
(``web-demos/hugging_face/inpainter/base_inpainter.py:163-374``): same constructor and ``inpaint`` signature, same
pre-processing (resize to even / multiple-of-8 sizes, per-frame masks dilated with scipy), same chunking of the
device path (it is ``inference_propainter.py:298-452`` again), numpy frames out.  The device work runs on the HIP
engine through ``pipeline.run_clip``."""
import numpy as np
import torch
from PIL import Image

from . import hip, video_io
from .model.modules.flow_comp_raft import assert_finite_flows
from .pipeline import InferenceConfig, run_clip


class ProInpainter:
    def __init__(self, propainter_checkpoint, raft_checkpoint, flow_completion_checkpoint, device="cuda:0", use_half=True,
                 raft_precision=None):
        """Checkpoint paths as in the reference; passing ``None`` for all three builds the deterministic seeded weights
        (no checkpoints ship with either repository).  ``raft_precision`` in {"f32", "f16x3", "f16"} (see ``RAFT_bi``);
        default: "f16x3" with ``use_half`` (the reference keeps RAFT fp32 in half mode, base_inpainter.py:240-262),
        "f32" otherwise."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ProInpainter runs on the HIP engine only (no CPU path in the product)")
        self.use_half = bool(use_half)
        hip.lib()
        prec = raft_precision or ("f16x3" if self.use_half else "f32")
        if propainter_checkpoint is None and raft_checkpoint is None and flow_completion_checkpoint is None:
            from .synthetic import seeded_models
            self.fix_raft, self.fix_flow_complete, self.model = seeded_models(self.device, raft_precision=prec)
        else:
            from .model.modules.flow_comp_raft import RAFT_bi
            from .model.propainter import InpaintGenerator
            from .model.recurrent_flow_completion import RecurrentFlowCompleteNet
            self.fix_raft = RAFT_bi(raft_checkpoint, self.device, precision=prec)
            self.fix_flow_complete = RecurrentFlowCompleteNet(flow_completion_checkpoint)
            for p in self.fix_flow_complete.parameters():
                p.requires_grad = False
            self.fix_flow_complete.to(self.device).eval()
            self.model = InpaintGenerator(model_path=propainter_checkpoint).to(self.device).eval()
            if self.use_half:
                self.fix_flow_complete, self.model = self.fix_flow_complete.half(), self.model.half()

    def _masks(self, masks, length, size, dilate):
        """read_mask_demo (base_inpainter.py:128-160): per-frame uint8 masks -> (flow_masks, masks_dilated) uint8 {0,255} ON THE DEVICE.
        The resize (nearest) stays on the host with the decode; the binary dilation -- scipy.ndimage.binary_dilation(iterations) in the
        reference -- runs as ``pp_binary_dilate`` (bit-identical to scipy: tests/test_ops_gpu.py)."""
        raw = []
        for m in masks:
            im = Image.fromarray(np.asarray(m).astype('uint8'))
            if size is not None:
                im = im.resize(size, Image.NEAREST)
            raw.append(np.array(im.convert('L')))
        if len(raw) == 1:
            raw = raw * length
        t = torch.from_numpy(np.stack(raw[:length])).to(self.device)
        if dilate > 0:
            d = hip.binary_dilate(t.contiguous(), int(dilate))
        else:
            d = (t.float() > 0.1).to(torch.uint8) * 255
        return d, d

    @torch.no_grad()
    def inpaint(self, npframes, masks, ratio=1.0, dilate_radius=4, raft_iter=20, subvideo_length=80, neighbor_length=10,
                ref_stride=10):
        """npframes: T x H x W x 3 uint8 (RGB); masks: T (or 1) x H x W, non-zero = hole.  Returns a list of T uint8 frames
        at the (even-sized) output resolution."""
        frames = [Image.fromarray(np.asarray(f).astype('uint8'), mode="RGB") for f in npframes]
        size = frames[0].size
        size = (int(ratio * size[0]) // 2 * 2, int(ratio * size[1]) // 2 * 2)      # even sizes for libx264 (:197-198)
        frames, size, out_size = video_io.resize_frames(frames, size)
        flow_masks, masks_dilated = self._masks(masks, len(frames), size, dilate_radius)
        frames_u8 = np.stack([np.asarray(f, dtype=np.uint8) for f in frames])
        cfg = InferenceConfig(raft_iter=raft_iter, subvideo_length=subvideo_length, neighbor_length=neighbor_length,
                              ref_stride=ref_stride, fp16=self.use_half)
        comp = run_clip((self.fix_raft, self.fix_flow_complete, self.model), frames_u8, flow_masks, masks_dilated, cfg,
                        self.device)
        # final resize (cv2.resize(f, out_size), base_inpainter.py:368-372) on the device, then ONE device->host copy
        comp = hip.resize_bilinear_u8(comp.contiguous(), out_size).cpu().numpy()
        assert_finite_flows(self.fix_raft)            # (after the D2H: the pass is synchronised)
        return list(comp)
