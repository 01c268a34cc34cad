"""Statistics of the streaming replay after loading a new clip (see profiles/r4_streaming_race.txt)."""
import numpy as np, torch, scipy.ndimage, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd.pipeline import InferenceConfig, run_clip
from propainter_amd.sharding import StreamingClipGraph
from propainter_amd.synthetic import synthetic_clip, synthetic_mask, seeded_models
models = seeded_models("cuda")
L, H, W = 34, 128, 192
m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.uint8) * 255
masks = np.repeat(m[None], L, 0)
clips = [synthetic_clip(L, H, W, seed=12 + i) for i in range(3)]
dev = torch.device("cuda")
cfg = InferenceConfig(raft_iter=3, subvideo_length=10, neighbor_length=4, ref_stride=3, fp16=True, batch_propagation=False)
models[0].precision = "f16x3"
refs = [run_clip(models, c, masks, masks, cfg, dev).clone() for c in clips]
sc = StreamingClipGraph(models, L, H, W, cfg, dev, share_pool=False, single_graph=False)      # the lockstep / concurrent orders need private pools
sc.load(clips[0], masks, masks); sc.capture()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for mode in ("wavefront", "wavefront_concurrent", "lockstep"):
    bad = []
    for it in range(N):
        k = (it + 1) % 3
        sc.load(clips[k], masks, masks)
        out = sc.replay(lockstep=(mode == "lockstep"), concurrent=(mode == "wavefront_concurrent"))
        torch.cuda.synchronize()
        ne = out != refs[k]
        if ne.any():
            bad.append((it, [i for i in range(L) if ne[i].any()][:3]))
    print(f"STREAM_DIAG {mode}: {len(bad)} of {N} passes wrong {bad[:4]}")

