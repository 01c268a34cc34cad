import numpy as np, torch, scipy.ndimage, dataclasses, sys
sys.path.insert(0, "/root/repo")
from propainter_amd.pipeline import InferenceConfig, run_clip
from propainter_amd.sharding import StreamingClipGraph
from propainter_amd.synthetic import synthetic_clip, synthetic_mask, seeded_models
models = seeded_models("cuda")
L, H, W = 34, 128, 192
m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.uint8) * 255
masks = np.repeat(m[None], L, 0)
clip_a, clip_b = synthetic_clip(L, H, W, seed=12), synthetic_clip(L, H, W, seed=13)
dev = torch.device("cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "batch"
cfg = InferenceConfig(raft_iter=3, subvideo_length=10, neighbor_length=4, ref_stride=3, fp16=True, batch_propagation=(mode == "batch"))
models[0].precision = "f16x3"
ref_a, ref_b = run_clip(models, clip_a, masks, masks, cfg, dev), run_clip(models, clip_b, masks, masks, cfg, dev)
sc = StreamingClipGraph(models, L, H, W, cfg, dev)
sc.load(clip_a, masks, masks); sc.capture()
out_a = sc.replay()
out_a2 = sc.replay()
sc.load(clip_b, masks, masks)
import os
if os.environ.get('SYNC_AFTER_LOAD') == '1':
    torch.cuda.synchronize()
out_b = sc.replay()
out_b_lock = sc.replay(lockstep=True)
torch.cuda.synchronize()
ref_b2 = run_clip(models, clip_b, masks, masks, cfg, dev)
torch.cuda.synchronize()
def d(n, x, y):
    ne = (x != y)
    print(f"[{mode}] {n}: {ne.float().mean().item():.3e} frames {[i for i in range(L) if ne[i].any()]}")
d("a", out_a, ref_a); d("a again", out_a2, ref_a); d("b", out_b, ref_b); d("b lock", out_b_lock, ref_b); d("ref_b vs ref_b2", ref_b, ref_b2); d("b vs ref_b2", out_b, ref_b2)
