"""Which torch (aten) device kernels does one steady-state pass still launch, from where, and what do they cost?  (round 6: the
"other (torch elementwise / copies)" row of the rocprof family table: 267 launches, 23 ms per 720x1280x80 pass.)
    python tools/list_torch_glue.py [frames=80] [height=720] [width=1280]
Runs the eager pass under torch.profiler (device activities + Python stacks) and prints, per (aten op, first propainter_amd call site),
the launch count and the device time of everything that is NOT a libpropainter_hip kernel."""
import collections
import os
import sys

import numpy as np
import scipy.ndimage
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from propainter_amd.pipeline import InferenceConfig, run_clip                          # noqa: E402
from propainter_amd.synthetic import seeded_models, synthetic_clip, synthetic_mask    # noqa: E402

L, H, W = (int(v) for v in (sys.argv[1:4] + ["80", "720", "1280"][len(sys.argv) - 1:]))
dev = torch.device("cuda")
models = seeded_models(dev, raft_precision="f16x3")
cfg = InferenceConfig(fp16=True, window_streams=1, raft_streams=1)
clip = torch.from_numpy(synthetic_clip(L, H, W)).to(dev)
m = scipy.ndimage.binary_dilation(synthetic_mask(H, W), iterations=4).astype(np.uint8) * 255
masks = torch.from_numpy(np.repeat(m[None], L, 0)).to(dev)
for _ in range(2):
    run_clip(models, clip, masks, masks, cfg, dev)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile                                    # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    run_clip(models, clip, masks, masks, cfg, dev)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0) or 0
    if dt <= 0 or not ev.name.startswith("aten::"):
        continue
    site = next((f"{os.path.basename(s.split('(')[0].strip())}:{s.split('(')[1].split(')')[0]}:{s.split(': ')[-1]}" if "(" in s else s
                 for s in (ev.stack or []) if "propainter_amd" in s and "hazard" not in s), "?")
    a = agg[(ev.name, site)]
    a[0] += 1
    a[1] += dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"TORCH_GLUE {L}x{H}x{W}: {sum(v[0] for _, v in rows)} aten ops with device time, {tot / 1e3:.2f} ms in all")
for (name, site), (n, us) in rows[:45]:
    print(f"TORCH_GLUE {us / 1e3:8.3f} ms {n:5d}x  {name:28s} {site}")
