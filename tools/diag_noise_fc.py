"""Flow completion at 720x1280 alone vs next to a second stream of matrix-core + HBM-bound work, bit for bit, stage by stage
(round 5: inside the stage-pipelined streaming graph the last sub-video's completed flows differ from replay to replay)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd.conv import ConvLayer                                             # noqa: E402
from propainter_amd.synthetic import seeded_models                                   # noqa: E402

dev = torch.device("cuda")
H, W = 720, 1280
T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
raft, fc, gen = seeded_models(dev)
fc = fc.half()
g = torch.Generator().manual_seed(3)
fl = (torch.randn(1, T, 2, H, W, generator=g) * 3).half().to(dev), (torch.randn(1, T, 2, H, W, generator=g) * 3).half().to(dev)
m = torch.zeros(1, T + 1, 1, H, W)
m[..., H // 3: 2 * H // 3, W // 3: 2 * W // 3] = 1
m = m.half().to(dev)
eng = fc._get_engine(torch.float16, dev)

noise_stream = torch.cuda.Stream(dev)
big = torch.randn(64 << 20, device=dev)
layer = ConvLayer(torch.randn(256, 256, 3, 3) / 48, None, padding=1, dtype=torch.float16, device=dev)
xin = torch.randn(8, 180, 320, 256, device=dev).half()
layer([xin])
torch.cuda.synchronize()


def noise(n):
    with torch.cuda.stream(noise_stream):
        for _ in range(n):
            big.mul_(1.0001)
            layer([xin])


def stages():
    mf = (fl[0] * (1 - m[:, :-1]))[0]
    x, e1 = eng._encode(mf, m[0, :-1])
    p = eng.propagate(x[:, None].contiguous(), 1, T)
    out = eng._decode(p[:, 0].contiguous(), e1)
    return {"encode": x, "encode_skip": e1, "propagate": p, "decode": out}


def whole():
    (pf, pb), _ = fc.forward_bidirect_flow(fl, m)
    return {"bidirect_f": pf, "bidirect_b": pb}


for name, fn in (("stages", stages), ("forward_bidirect_flow", whole)):
    ref = {k: v.clone() for k, v in fn().items()}
    torch.cuda.synchronize()
    solo = all(torch.equal(v, ref[k]) for k, v in fn().items())
    torch.cuda.synchronize()
    bad = {}
    for _ in range(REPS):
        noise(600)
        out = fn()
        torch.cuda.synchronize()
        for k, v in out.items():
            if not torch.equal(v, ref[k]):
                bad[k] = bad.get(k, 0) + 1
    print(f"NOISE_FC {name} 720x1280 t={T}: solo repeatable {solo}; stages differing under concurrent load (of {REPS}): {bad or 'none'}", flush=True)
