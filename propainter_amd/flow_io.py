"""Optical-flow files in the reference's on-disk format (``utils/flow_util.py:28-89`` as written by
``scripts/compute_flow.py``): 4-byte tag ``PIEH``, int32 width, int32 height, then h*w*2 **float16** values
(x then y displacement per pixel, row-major).  Lets RAFT output be cached and reused between runs."""
import os

import numpy as np

TAG = b'PIEH'


def flowwrite(flow, filename):
    """flow: (h, w, 2) array (any float dtype) -> file (values stored as float16, like the reference)."""
    flow = np.asarray(flow)
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise ValueError(f"flow must be (h, w, 2), got {flow.shape}")
    d = os.path.dirname(os.path.abspath(filename))
    os.makedirs(d, exist_ok=True)
    with open(filename, 'wb') as f:
        f.write(TAG)
        np.array([flow.shape[1], flow.shape[0]], dtype=np.int32).tofile(f)
        flow.astype(np.float16).tofile(f)


def flowread(filename):
    """-> (h, w, 2) float32."""
    with open(filename, 'rb') as f:
        if f.read(4) != TAG:
            raise IOError(f'Invalid flow file: {filename}, header does not contain PIEH')
        w, h = (int(v) for v in np.fromfile(f, np.int32, 2))
        data = np.fromfile(f, np.float16, w * h * 2)
    if data.size != w * h * 2:
        raise IOError(f'Invalid flow file: {filename}: expected {w * h * 2} values, found {data.size}')
    return data.reshape(h, w, 2).astype(np.float32)


def save_clip_flows(flows_f, flows_b, folder):
    """flows_*: torch / numpy [t-1, 2, h, w] (the RAFT_bi outputs of one clip) -> <folder>/{00000_f.flo, 00000_b.flo, ...}."""
    for name, fl in (("f", flows_f), ("b", flows_b)):
        a = fl.detach().float().cpu().numpy() if hasattr(fl, "detach") else np.asarray(fl, dtype=np.float32)
        for i in range(a.shape[0]):
            flowwrite(np.transpose(a[i], (1, 2, 0)), os.path.join(folder, f"{i:05d}_{name}.flo"))


def load_clip_flows(folder):
    """Inverse of save_clip_flows -> (flows_f, flows_b) float32 numpy [t-1, 2, h, w]."""
    out = []
    for name in ("f", "b"):
        files = sorted(f for f in os.listdir(folder) if f.endswith(f"_{name}.flo"))
        out.append(np.stack([np.transpose(flowread(os.path.join(folder, f)), (2, 0, 1)) for f in files]))
    return out[0], out[1]
