"""State-dict schemas.  The drop-in boundary includes ``load_state_dict(strict=True)`` with the
reference's key names, so every model here is a tree of parameter holders generated from a flat schema
``[(dotted.key, shape, kind)]`` rather than a stack of torch layers: the arithmetic lives in the HIP engine."""
import math

import torch
import torch.nn as nn


class ParamTree(nn.Module):
    """Nested holder; ``add('a.b.0.weight', tensor)`` registers ``a.b.0.weight`` as a parameter (or buffer)."""

    def add(self, dotted, tensor, buffer=False):
        node = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, ParamTree())
            node = node._modules[p]
        if buffer:
            node.register_buffer(parts[-1], tensor)
        else:
            node.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))

    def get(self, dotted):
        node = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node._modules[p]
        leaf = parts[-1]
        if leaf in node._parameters:
            return node._parameters[leaf]
        return node._buffers[leaf]

    def forward(self, *a, **k):  # pragma: no cover - holders are never called
        raise RuntimeError("ParamTree holds parameters only")


def conv_entries(prefix, cout, cin, kh, kw=None, bias=True):
    kw = kh if kw is None else kw
    e = [(f"{prefix}.weight", (cout, cin, kh, kw), "w")]
    if bias:
        e.append((f"{prefix}.bias", (cout,), "b"))
    return e


def linear_entries(prefix, cout, cin):
    return [(f"{prefix}.weight", (cout, cin), "w"), (f"{prefix}.bias", (cout,), "b")]


def norm_entries(prefix, c, running=False):
    e = [(f"{prefix}.weight", (c,), "one"), (f"{prefix}.bias", (c,), "zero")]
    if running:
        e += [(f"{prefix}.running_mean", (c,), "buf_zero"), (f"{prefix}.running_var", (c,), "buf_one"),
              (f"{prefix}.num_batches_tracked", (), "buf_long")]
    return e


def populate(tree, schema, std=None):
    """Creates the tensors of a schema with PyTorch-default-like init (kaiming-uniform(a=sqrt(5)) for weights,
    or N(0, std) when ``std`` is given, as the reference's BaseNetwork.init_weights does)."""
    for name, shape, kind in schema:
        if kind == "w":
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            if std is None:
                bound = 1.0 / math.sqrt(fan_in)
                t = torch.empty(shape).uniform_(-bound, bound)
            else:
                t = torch.empty(shape).normal_(0.0, std)
            tree.add(name, t)
        elif kind == "b":
            tree.add(name, torch.zeros(shape))
        elif kind == "one":
            tree.add(name, torch.ones(shape))
        elif kind == "zero":
            tree.add(name, torch.zeros(shape))
        elif kind == "buf_zero":
            tree.add(name, torch.zeros(shape), buffer=True)
        elif kind == "buf_one":
            tree.add(name, torch.ones(shape), buffer=True)
        elif kind == "buf_long":
            tree.add(name, torch.zeros(shape, dtype=torch.long), buffer=True)
        elif kind == "const":
            raise ValueError("const entries must be added explicitly")
        else:
            raise ValueError(kind)
