"""Generates tests/golden/*.npz from the REAL reference (/root/reference) run on CPU with seeded weights.

Run in the authoring container only:  python -m oracle.make_golden
The fixtures pin (a) the state-dict schemas, (b) per-module forward outputs of the reference at tiny sizes and
(c) an end-to-end clip produced by the reference's own driver script run as __main__ (oracle/run_reference_driver.py;
``python -m oracle.make_golden e2e`` regenerates only that fixture).
"""
import json
import os
import warnings

import numpy as np
import torch

from . import propainter_oracle as O
from .ref_shims import build_reference_raft, load_reference
from propainter_amd.synthetic import seeded_weights, synthetic_clip, synthetic_mask

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def e2e_from_reference_main(sds):
    """End-to-end fixture: 10 frames 128x192 through the REFERENCE'S OWN SCRIPT run as __main__ (oracle/run_reference_driver.py: frame /
    mask PNG folders, seeded checkpoint files in ./weights, --subvideo_length 6 so that every chunked branch of
    inference_propainter.py:341-404 and the ref_num branch of get_ref_index:159-173 run).  Rounds 1-5 built this file from the restated
    driver; tests/test_oracle_cpu.py now holds the restatement to this run (byte for byte over the reference's modules)."""
    from .run_reference_driver import reference_main_on_clip
    H, W, L = 128, 192, 10
    clip = synthetic_clip(L, H, W, seed=7)
    kw = dict(raft_iter=4, subvideo_length=6, neighbor_length=4, ref_stride=3)
    r = reference_main_on_clip(clip, synthetic_mask(H, W).astype(np.uint8), sds,
                               [a for k, v in kw.items() for a in ("--" + k, v)] + ["--mask_dilation", 4])
    assert np.array_equal(r["masks_dilated"], r["flow_masks"])
    np.savez_compressed(os.path.join(OUT, "e2e_128x192.npz"), frames_u8=clip, masks_u8=r["masks_dilated"], comp=r["comp"],
                        pred_f=r["pred_f"].astype(np.float16)[None], upd_masks=r["upd_masks"].astype(np.uint8)[None],
                        source="reference __main__ (oracle/run_reference_driver.py)", **kw)


def main(only=None):
    warnings.filterwarnings("ignore")
    torch.set_num_threads(os.cpu_count())
    os.makedirs(OUT, exist_ok=True)
    ns = load_reference()
    raft = build_reference_raft()
    fc = ns.RecurrentFlowCompleteNet().eval()
    gen = ns.InpaintGenerator(init_weights=True).eval()
    schema = {k: {n: list(v.shape) for n, v in m.state_dict().items()} for k, m in (("raft", raft), ("fc", fc), ("gen", gen))}
    with open(os.path.join(OUT, "state_dict_schema.json"), "w") as f:
        json.dump(schema, f, indent=0, sort_keys=True)
    sds = {"raft": seeded_weights("raft", raft.state_dict()), "fc": seeded_weights("fc", fc.state_dict()),
           "gen": seeded_weights("gen", gen.state_dict())}
    if only == "e2e":
        e2e_from_reference_main(sds)
        return
    raft.load_state_dict(sds["raft"]); fc.load_state_dict(sds["fc"]); gen.load_state_dict(sds["gen"])

    with torch.no_grad():
        # ---- RAFT: 3 frames 128x192, 6 iterations, through the reference's RAFT_bi call pattern
        H, W = 128, 192
        clip = synthetic_clip(3, H, W, seed=5)
        frames = torch.from_numpy(clip).permute(0, 3, 1, 2).float().div(255)[None] * 2 - 1
        a, b = frames[0, :-1], frames[0, 1:]
        _, ff = raft(a, b, iters=6, test_mode=True)
        _, fb = raft(b, a, iters=6, test_mode=True)
        np.savez_compressed(os.path.join(OUT, "raft_128x192.npz"), frames_u8=clip, iters=6, flows_f=_np(ff), flows_b=_np(fb))

        # ---- flow completion: 5 flows 64x96
        H, W, t = 64, 96, 5
        g = torch.Generator().manual_seed(21)
        ffl = torch.randn(1, t, 2, H, W, generator=g) * 2
        fbl = torch.randn(1, t, 2, H, W, generator=g) * 2
        m = torch.zeros(1, t + 1, 1, H, W); m[:, :, :, 20:44, 30:66] = 1
        (pf, pb), _ = fc.forward_bidirect_flow((ffl, fbl), m)
        cf, cb = fc.combine_flow((ffl, fbl), (pf, pb), m)
        np.savez_compressed(os.path.join(OUT, "fc_64x96.npz"), flows_f=_np(ffl), flows_b=_np(fbl), masks=_np(m),
                            pred_f=_np(pf), pred_b=_np(pb), comb_f=_np(cf), comb_b=_np(cb))

        # ---- generator: t=5 (3 local + 2 reference) 64x96, and image propagation
        t, lt = 5, 3
        fr = torch.rand(1, t, 3, H, W, generator=g) * 2 - 1
        mk = torch.zeros(1, t, 1, H, W); mk[:, :, :, 20:44, 30:66] = 1
        mu = torch.zeros(1, t, 1, H, W); mu[:, :, :, 26:40, 40:60] = 1
        fl = (torch.randn(1, lt - 1, 2, H, W, generator=g) * 1.5, torch.randn(1, lt - 1, 2, H, W, generator=g) * 1.5)
        out = gen(fr * (1 - mk), fl, mk, mu, lt)
        f1, f2 = torch.randn(1, t - 1, 2, H, W, generator=g) * 2, torch.randn(1, t - 1, 2, H, W, generator=g) * 2
        pi, pm = gen.img_propagation(fr * (1 - mk), (f1, f2), mk, 'nearest')
        np.savez_compressed(os.path.join(OUT, "gen_64x96.npz"), frames=_np(fr), masks_in=_np(mk), masks_upd=_np(mu),
                            flows_f=_np(fl[0]), flows_b=_np(fl[1]), lt=lt, out=_np(out), ip_flows_f=_np(f1),
                            ip_flows_b=_np(f2), ip_frames=_np(pi.view(1, t, 3, H, W)), ip_masks=_np(pm.view(1, t, 1, H, W)))

    e2e_from_reference_main(sds)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    import sys
    main(sys.argv[1] if len(sys.argv) > 1 else None)
