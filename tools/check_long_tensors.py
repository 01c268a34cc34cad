"""Do the kernels address tensors beyond 2^31 elements / 2^32 bytes correctly?  (round 5: at BASELINE config 4 -- 320 frames at 720x1280 -- the LAST
sub-video's frames differ by a byte here and there between graph forms and runs; its tensors are the ones that end beyond 4 GiB.)
 (1) flow completion at 720x1280 over t flows (default 86: a sub-video chunk with its halos) against the CPU oracle, error PER FRAME;
 (2) the generator's per-clip cache over 320 frames against the same cache built from two halves (bitwise)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import propainter_oracle as O                                             # noqa: E402  (a checker, like the tests)
from propainter_amd.synthetic import seeded_models                                   # noqa: E402

dev = torch.device("cuda")
H, W = 720, 1280
T = int(sys.argv[1]) if len(sys.argv) > 1 else 86
raft, fc, gen = seeded_models(dev)
sd = {k: v.float().cpu() for k, v in fc.state_dict().items()}
g = torch.Generator().manual_seed(9)
fl = torch.randn(1, T, 2, H, W, generator=g) * 3
m = torch.zeros(1, T, 1, H, W)
m[..., H // 3: 2 * H // 3, W // 3: 2 * W // 3] = 1
torch.set_num_threads(min(32, os.cpu_count() or 1))
with torch.no_grad():
    ref = O.fc_forward(sd, fl * (1 - m), m)
for dt in (torch.float32, torch.float16):
    fcm = fc.float() if dt == torch.float32 else fc.half()
    out, _ = fcm((fl * (1 - m)).to(dev, dt), m.to(dev, dt))
    torch.cuda.synchronize()
    err = (out.float().cpu() - ref).abs().amax(dim=(0, 2, 3, 4))
    rng = ref.abs().max().item()
    worst = int(err.argmax())
    print(f"LONG_TENSORS flow completion {dt} t={T}: per-frame max |d| / range: first {err[0] / rng:.2e}, median {err.median() / rng:.2e}, "
          f"last {err[-1] / rng:.2e}, worst frame {worst}: {err[worst] / rng:.2e}; frames above 3x the median: "
          f"{[int(i) for i in (err > 3 * err.median()).nonzero().flatten()[:12]]}", flush=True)
del out, ref
torch.cuda.empty_cache()
L = 320
gen = gen.half()
fr = (torch.rand(1, L, 3, H, W, generator=g) * 2 - 1).half().to(dev)
mk = torch.zeros(1, L, 1, H, W, dtype=torch.float16, device=dev)
mk[..., H // 3: 2 * H // 3, W // 3: 2 * W // 3] = 1
enc_all = gen.encode_frames(fr, mk, mk)
a = gen.encode_frames(fr[:, :160].contiguous(), mk[:, :160].contiguous(), mk[:, :160].contiguous())
b = gen.encode_frames(fr[:, 160:].contiguous(), mk[:, 160:].contiguous(), mk[:, 160:].contiguous())
torch.cuda.synchronize()
ne = [i for i in range(L) if not torch.equal(enc_all[i], (a if i < 160 else b)[i % 160])]
print(f"LONG_TENSORS generator encoder cache over {L} frames ({enc_all.numel() * 2 / 2 ** 30:.2f} GiB) vs two halves: frames that differ: {ne[:12]} ({len(ne)} of {L})", flush=True)
