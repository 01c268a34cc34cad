"""Host-side engine graphs under the CPU emulation of the device ops (tests/cpu_emulation.py) against the golden
vectors of the REAL reference.  Proves channel windows, multi-source convolutions, packed weights / K tables,
fused projections, fold rewrites, recurrences and the clip scheduler without a GPU; the HIP kernels themselves are
proven by the -m gpu tests."""
import numpy as np
import pytest
import torch

from oracle import propainter_oracle as O
from tests.cpu_emulation import emulated_device_ops
from tests.helpers import load_golden, seeded_models


@pytest.fixture(scope="module")
def models():
    return seeded_models("cpu")


def _engine(mod, dtype=torch.float32):
    if hasattr(mod, "PRECISIONS"):       # RAFT_bi: engines are keyed by precision mode
        return mod._get_engine("f32" if dtype == torch.float32 else "f16", torch.device("cpu"))
    return mod._get_engine(dtype, torch.device("cpu"))


def test_raft_graph(models):
    raft = models[0]
    g = load_golden("raft_128x192.npz")
    fr = torch.from_numpy(g["frames_u8"]).permute(0, 3, 1, 2).float().div(255)[None] * 2 - 1
    with emulated_device_ops():
        eng = _engine(raft)
        b, l_t, c, h, w = fr.shape
        from propainter_amd import hip
        x = hip.nchw_to_nhwc(fr.reshape(b * l_t, c, h, w), out_dtype=torch.float32, cpad=8)
        fmap = eng.encode(eng.fnet, x, True)
        ctx = eng.encode(eng.cnet, x, False)
        up = eng.refine(torch.cat([fmap[:-1], fmap[1:]]), torch.cat([fmap[1:], fmap[:-1]]), torch.cat([ctx[:-1], ctx[1:]]),
                        int(g["iters"]))
    assert (up[:2] - torch.from_numpy(g["flows_f"])).abs().max() < 2e-3
    assert (up[2:] - torch.from_numpy(g["flows_b"])).abs().max() < 2e-3


def test_flow_completion_graph(models):
    fc = models[1]
    g = load_golden("fc_64x96.npz")
    fl, m = torch.from_numpy(g["flows_f"]), torch.from_numpy(g["masks"])
    with emulated_device_ops():
        out = _engine(fc).forward(fl * (1 - m[:, :-1]), m[:, :-1].contiguous())
    assert (out - torch.from_numpy(g["pred_f"])).abs().max() < 1e-3


def test_flow_completion_bidirectional_batched_graph(models):
    """forward_bidirect_flow stacks the forward and the time-flipped backward sequence along the batch axis (the
    reference runs the net twice, recurrent_flow_completion.py:312-337): both directions must match the goldens."""
    fc = models[1]
    g = load_golden("fc_64x96.npz")
    fl = (torch.from_numpy(g["flows_f"]), torch.from_numpy(g["flows_b"]))
    m = torch.from_numpy(g["masks"])
    with emulated_device_ops():
        (pf, pb), edges = fc.forward_bidirect_flow(fl, m)
        cf, cb = fc.combine_flow(fl, (pf, pb), m)
    assert edges == [None, None]
    assert (pf - torch.from_numpy(g["pred_f"])).abs().max() < 1e-3
    assert (pb - torch.from_numpy(g["pred_b"])).abs().max() < 1e-3
    assert (cf - torch.from_numpy(g["comb_f"])).abs().max() < 1e-3


def test_generator_cached_encoder_features(models):
    """encode_frames() once + forward(enc_feat=slice) == forward() (the encoder is per-frame)."""
    gen = models[2]
    g = load_golden("gen_64x96.npz")
    fr, mk, mu = (torch.from_numpy(g[k]) for k in ("frames", "masks_in", "masks_upd"))
    fl = (torch.from_numpy(g["flows_f"]), torch.from_numpy(g["flows_b"]))
    with emulated_device_ops():
        feat = gen.encode_frames(fr * (1 - mk), mk, mu)
        ids = list(range(fr.shape[1]))
        out = gen(fr * (1 - mk), fl, mk, mu, int(g["lt"]), enc_feat=feat[ids])
    assert (out - torch.from_numpy(g["out"])).abs().max() < 1e-3


def test_generator_graph(models):
    gen = models[2]
    g = load_golden("gen_64x96.npz")
    fr, mk, mu = (torch.from_numpy(g[k]) for k in ("frames", "masks_in", "masks_upd"))
    with emulated_device_ops():
        eng = _engine(gen)
        out = eng.forward(fr * (1 - mk), (torch.from_numpy(g["flows_f"]), torch.from_numpy(g["flows_b"])), mk, mu,
                          int(g["lt"]), "bilinear", 2)
        pi, pm = eng.img_propagation(fr * (1 - mk), torch.from_numpy(g["ip_flows_f"]), torch.from_numpy(g["ip_flows_b"]), mk,
                                     "nearest")
    assert (out - torch.from_numpy(g["out"])).abs().max() < 1e-3
    assert torch.equal(pi, torch.from_numpy(g["ip_frames"])) and torch.equal(pm, torch.from_numpy(g["ip_masks"]))


def test_clip_pipeline_graph(models):
    """Whole clip path (chunked RAFT / completion / propagation, window schedule, device-side blend) under emulation
    vs the golden composited frames."""
    from propainter_amd.pipeline import InferenceConfig, run_clip
    g = load_golden("e2e_128x192.npz")
    cfg = InferenceConfig(raft_iter=int(g["raft_iter"]), subvideo_length=int(g["subvideo_length"]),
                          neighbor_length=int(g["neighbor_length"]), ref_stride=int(g["ref_stride"]), fp16=False)
    with emulated_device_ops():
        comp = run_clip(models, g["frames_u8"], g["masks_u8"], g["masks_u8"], cfg, torch.device("cpu"))
    comp = comp.numpy()
    assert comp.shape == g["comp"].shape
    assert O.psnr(comp, g["comp"]) > 50.0
    assert (comp != g["comp"]).mean() < 0.01


def test_raft_fp16_graph_uses_the_on_the_fly_correlation(models):
    """The fp16 RAFT engine takes the volume-free correlation path (feature pyramid + on-the-fly lookup); under the CPU
    emulation of the device ops its flow must stay within fp16 end-point error of the real reference's golden."""
    raft = models[0]
    g = load_golden("raft_128x192.npz")
    fr = torch.from_numpy(g["frames_u8"]).permute(0, 3, 1, 2).float().div(255)[None] * 2 - 1
    with emulated_device_ops():
        eng = _engine(raft, torch.float16)
        assert eng.corr_otf and not _engine(raft, torch.float32).corr_otf
        b, l_t, c, h, w = fr.shape
        from propainter_amd import hip
        x = hip.nchw_to_nhwc(fr.reshape(b * l_t, c, h, w), out_dtype=torch.float16, cpad=8)
        fmap = eng.encode(eng.fnet, x, True)
        ctx = eng.encode(eng.cnet, x, False)
        up = eng.refine(torch.cat([fmap[:-1], fmap[1:]]).contiguous(), torch.cat([fmap[1:], fmap[:-1]]).contiguous(),
                        torch.cat([ctx[:-1], ctx[1:]]).contiguous(), int(g["iters"]))
    epe = (up[:2].float() - torch.from_numpy(g["flows_f"])).pow(2).sum(1).sqrt()
    assert epe.mean() < 0.05 and epe.max() < 0.5, (epe.mean(), epe.max())
