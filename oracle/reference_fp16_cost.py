"""What does the REFERENCE'S OWN fp16 path cost in accuracy?  (test infrastructure; authoring container only)

The reference runs stages B-D in half precision under ``--fp16`` (inference_propainter.py:333-337: ``fix_flow_complete.half()``,
``model.half()``, half frames / masks / flows).  This script runs the REAL reference modules (oracle/ref_shims.py) in half ON CPU --
PyTorch's CPU half kernels: half storage, half arithmetic with the accumulation of the respective kernel -- on exactly the inputs on
which tests/test_modules_gpu.py measures this engine's fp16 path, and reports both against the same fp32 reference output:

    python -m oracle.reference_fp16_cost            # prints one table row per stage / size

  * fc_64x96.npz / gen_64x96.npz  (tests/golden: inputs + the reference's fp32 outputs)
  * the seeded 432x240 stage inputs of test_stages_at_432x240_vs_oracle (flow completion t = 6; generator window t = 7, l_t = 5)

The engine's figures (fp16 storage + fp16 MFMA products, fp32 accumulation, fp32 coordinates) come from the GPU runs of those tests
(MODULE_PARITY lines, profiles/).  A CPU half kernel is not the CUDA half kernel bit for bit; the order of magnitude is what the
table is for."""
import os
import sys
import warnings

import numpy as np
import torch

from .ref_shims import load_reference

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(got, ref):
    got, ref = got.float(), ref.float()
    return float((got - ref).abs().max() / ref.abs().max()), float((got - ref).abs().mean() / ref.abs().max())


def main():
    warnings.filterwarnings("ignore")
    torch.set_num_threads(int(os.environ.get("PP_REF_THREADS", "4")))
    from tests.helpers import load_golden, seeded_sds
    ns = load_reference()
    sds = seeded_sds()
    fc32 = ns.RecurrentFlowCompleteNet().eval()
    fc32.load_state_dict(sds["fc"])
    gen32 = ns.InpaintGenerator(init_weights=False).eval()
    gen32.load_state_dict(sds["gen"])
    import copy
    fc16, gen16 = copy.deepcopy(fc32).half(), copy.deepcopy(gen32).half()
    rows = []
    with torch.no_grad():
        g = load_golden("fc_64x96.npz")
        fl = (torch.from_numpy(g["flows_f"]), torch.from_numpy(g["flows_b"]))
        m = torch.from_numpy(g["masks"])
        (pf, pb), _ = fc16.forward_bidirect_flow((fl[0].half(), fl[1].half()), m.half())
        rows.append(("flow completion 64x96 (t = 5)", *rel(torch.cat([pf, pb]), torch.cat([torch.from_numpy(g["pred_f"]), torch.from_numpy(g["pred_b"])]))))
        g = load_golden("gen_64x96.npz")
        fr, mk, mu = (torch.from_numpy(g[k]) for k in ("frames", "masks_in", "masks_upd"))
        fl = (torch.from_numpy(g["flows_f"]), torch.from_numpy(g["flows_b"]))
        out = gen16((fr * (1 - mk)).half(), (fl[0].half(), fl[1].half()), mk.half(), mu.half(), int(g["lt"]))
        rows.append(("generator window 64x96 (t = 5, l_t = 3)", *rel(out, torch.from_numpy(g["out"]))))
        # the 432x240 inputs of tests/test_modules_gpu.py::test_stages_at_432x240_vs_oracle (same generator, same draws)
        gq = torch.Generator().manual_seed(77)
        H, W, t = 240, 432, 6
        fl = (torch.randn(1, t, 2, H, W, generator=gq) * 3, torch.randn(1, t, 2, H, W, generator=gq) * 3)
        m = torch.zeros(1, t + 1, 1, H, W)
        m[:, :, :, 80:160, 144:288] = 1
        (rf, rb), _ = fc32.forward_bidirect_flow(fl, m)
        (pf, pb), _ = fc16.forward_bidirect_flow((fl[0].half(), fl[1].half()), m.half())
        rows.append(("flow completion 240x432 (t = 6)", *rel(torch.cat([pf, pb]), torch.cat([rf, rb]))))
        tt, lt = 7, 5
        fr = torch.rand(1, tt, 3, H, W, generator=gq) * 2 - 1
        mk = torch.zeros(1, tt, 1, H, W)
        mk[:, :, :, 80:160, 144:288] = 1
        mu = torch.zeros(1, tt, 1, H, W)
        mu[:, :, :, 100:140, 180:250] = 1
        gfl = (torch.randn(1, lt - 1, 2, H, W, generator=gq) * 2, torch.randn(1, lt - 1, 2, H, W, generator=gq) * 2)
        ref = gen32(fr * (1 - mk), gfl, mk, mu, lt)
        out = gen16((fr * (1 - mk)).half(), (gfl[0].half(), gfl[1].half()), mk.half(), mu.half(), lt)
        rows.append(("generator window 240x432 (t = 7, l_t = 5)", *rel(out, ref)))
    print("REFERENCE_FP16_COST  stage | max|d| / range | mean|d| / range   (the reference's half path vs its own fp32 path, CPU)")
    for name, mx, mean in rows:
        print(f"REFERENCE_FP16_COST  {name} | {mx:.2e} | {mean:.2e}")


if __name__ == "__main__":
    main()
