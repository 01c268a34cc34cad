"""BASELINE config 1 on REAL frames: the first 8 frames of the reference's own sample clip
(/root/reference/inputs/object_removal/bmx-trees{,_mask}: 432x240 JPEG frames + per-frame object masks) through the
driver pre-processing (resize to the same size, mask dilation 4: inference_propainter.py:70-115) and the fp32 CPU oracle
with the reference's default settings (raft_iter 20, neighbor_length 10, ref_stride 10).

Run in the authoring container only:  python -m oracle.make_golden_bmx [8 | 40]
Writes tests/golden/bmx_trees_432x240x<n>.npz = decoded frames, dilated masks, composited oracle frames.  (Seeded weights: the
pretrained checkpoints are not available offline; what the fixture adds over the synthetic clips is real image statistics --
JPEG texture, camera motion, a moving object mask that changes every frame.)"""
import os
import warnings

import numpy as np
import torch
from PIL import Image

from . import propainter_oracle as O
from propainter_amd import video_io
from propainter_amd.synthetic import seeded_models

SRC = "/root/reference/inputs/object_removal"
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main(n=8):
    OUT = os.path.join(GOLDEN, f"bmx_trees_432x240x{n}.npz")
    warnings.filterwarnings("ignore")
    torch.set_num_threads(os.cpu_count())
    names = sorted(os.listdir(os.path.join(SRC, "bmx-trees")))[:n]
    frames = [Image.open(os.path.join(SRC, "bmx-trees", f)).convert("RGB") for f in names]
    frames, size, _ = video_io.resize_frames(frames, frames[0].size)
    flow_masks, masks_dilated = video_io.read_masks(os.path.join(SRC, "bmx-trees_mask"), n, size, flow_mask_dilates=4, mask_dilates=4)
    fr = np.stack([np.asarray(f, dtype=np.uint8) for f in frames])
    fm, md = np.stack(flow_masks[:n]), np.stack(masks_dilated[:n])
    raft, fc, gen = seeded_models("cpu")
    sds = {"raft": {k: v.float() for k, v in raft.fix_raft.state_dict().items()},
           "fc": {k: v.float() for k, v in fc.state_dict().items()},
           "gen": {k: v.float() for k, v in gen.state_dict().items()}}
    kw = dict(raft_iter=20, subvideo_length=80, neighbor_length=10, ref_stride=10)
    with torch.no_grad():
        comp = O.inpaint_video(sds, fr, fm, md, **kw)
    comp = np.stack(comp)
    hole = md > 0
    assert np.array_equal(comp[~hole], fr[~hole])          # outside the dilated mask the composite IS the input frame
    np.savez_compressed(OUT, frames_u8=fr, flow_masks_u8=fm, masks_u8=md, comp_hole=comp[hole], **kw)    # comp = frames with comp_hole pasted
    print(OUT, os.path.getsize(OUT), fr.shape, float((md > 0).mean()))


if __name__ == "__main__":
    import sys
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)      # 8: the small fixture; 40: BASELINE config 1 as stated (the first 40 frames)
