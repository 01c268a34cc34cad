"""Is a stage's output independent of what else runs on the GPU?  Runs a stage alone, then again while a second stream keeps the chip busy
(HBM-bound elementwise passes + matrix-core convolutions of this library), and compares bit for bit.  Round 5: the streaming schedule fails
exactly when RAFT overlaps other work (profiles/r5_streaming_single_graph.txt)."""
import os
import sys

import numpy as np
import scipy.ndimage
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd import hip                                                        # noqa: E402
from propainter_amd.conv import ConvLayer                                             # noqa: E402
from propainter_amd.synthetic import seeded_models, synthetic_clip, synthetic_mask   # noqa: E402

dev = torch.device("cuda")
H, W, L = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 192, 11)))
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 6
raft, fc, gen = seeded_models(dev)
clip = synthetic_clip(L, H, W, seed=12)
fr = (torch.from_numpy(clip).to(dev).permute(0, 3, 1, 2).float().div(255)[None] * 2 - 1).contiguous()
m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.float32)
masks = torch.from_numpy(np.repeat(m[None, None], L, 0))[None].to(dev)

noise_stream = torch.cuda.Stream(dev)
big = torch.randn(64 << 20, device=dev)
wt = torch.randn(256, 256, 3, 3) / 48
layer = ConvLayer(wt, None, padding=1, dtype=torch.float16, device=dev)
xin = torch.randn(8, 180, 320, 256, device=dev).half()
layer([xin])
torch.cuda.synchronize()


def noise(n):
    with torch.cuda.stream(noise_stream):
        for _ in range(n):
            big.mul_(1.0001)
            layer([xin])


def check(name, fn, noise_launches=400):
    ref = fn()
    torch.cuda.synchronize()
    ref = [t.clone() for t in ref]
    same_solo = all(all(torch.equal(a, b) for a, b in zip(fn(), ref)) for _ in range(2))
    torch.cuda.synchronize()
    bad = 0
    for _ in range(REPS):
        noise(noise_launches)
        out = fn()
        torch.cuda.synchronize()
        bad += int(not all(torch.equal(a, b) for a, b in zip(out, ref)))
    print(f"NOISE_DIAG {name} {H}x{W}x{L}: solo repeatable {same_solo}; wrong under concurrent load: {bad} of {REPS}", flush=True)


for prec in ("f16x3", "f16", "f32"):
    raft.precision = prec
    check(f"RAFT {prec}", lambda: raft(fr, iters=3))

raft.precision = "f16x3"
eng = raft._get_engine("f16x3", dev)
x = hip.nchw_to_nhwc(fr[0].contiguous().float(), cpad=8, split=True)
check("RAFT f16x3 fnet encoder", lambda: [eng.encode(eng.fnet, x, True)])
check("RAFT f16x3 cnet encoder", lambda: list(eng.encode(eng.cnet, x, False)))
fm = eng.encode(eng.fnet, x, True)
cx = eng.encode(eng.cnet, x, False)
torch.cuda.synchronize()
check("RAFT f16x3 refine", lambda: [eng.refine(fm[:-1], fm[1:], tuple(c[:-1] for c in cx), 3)])
flows = raft(fr, iters=3)
fl16 = (flows[0].half(), flows[1].half())
fcm = masks.half()
check("flow completion f16", lambda: list(fc.half().forward_bidirect_flow(fl16, fcm)[0]))
