"""State-dict schema of the reference checkpoints.

The names are part of the drop-in contract: ``eval.py:124-125`` loads
``state_dict['model']`` (keys prefixed with ``module.``) strictly, so the
engine consumes exactly these 150 tensors.  This module lists them without
instantiating any network, and offers the prefix handling used by the loaders.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Mapping, Tuple

import torch

Shape = Tuple[int, ...]


def _conv_bn(d: "OrderedDict[str, Shape]", prefix: str, cout: int, cin: int, k: int = 3) -> None:
    d[prefix + "conv.weight"] = (cout, cin, k, k)
    d[prefix + "bn.weight"] = (cout,)
    d[prefix + "bn.bias"] = (cout,)
    d[prefix + "bn.running_mean"] = (cout,)
    d[prefix + "bn.running_var"] = (cout,)
    d[prefix + "bn.num_batches_tracked"] = ()


def state_dict_schema() -> "OrderedDict[str, Shape]":
    """name -> shape, in the reference's registration order."""
    d: "OrderedDict[str, Shape]" = OrderedDict()
    fn = "feature_net."
    _conv_bn(d, fn + "conv1.", 8, 3)
    cin = 8
    for li, cout in ((1, 16), (2, 32), (3, 48)):
        p = f"{fn}layer{li}."
        _conv_bn(d, p + "0.conv1.", cout, cin)
        _conv_bn(d, p + "0.conv2.", cout, cout)
        _conv_bn(d, p + "0.downsample.", cout, cin)
        _conv_bn(d, p + "1.conv1.", cout, cout)
        _conv_bn(d, p + "1.conv2.", cout, cout)
        cin = cout
    for name, cout in (("output3", 48), ("output2", 32), ("output1", 16)):
        d[fn + name + ".weight"] = (cout, 48, 3, 3)
        d[fn + name + ".bias"] = (cout,)
    for name, c in (("inner1", 16), ("inner2", 32), ("inner3", 48)):
        d[fn + name + ".weight"] = (48, c, 1, 1)
        d[fn + name + ".bias"] = (48,)

    ev = "iter_mvs.evaluation."
    d[ev + "pixel_view_weight.conv.0.conv.weight"] = (16, 8, 3, 3)
    d[ev + "pixel_view_weight.conv.1.weight"] = (1, 16, 1, 1)
    d[ev + "pixel_view_weight.conv.1.bias"] = (1,)
    for i in range(3):
        p = f"{ev}corr_conv1.{i}."
        d[p + "conv0.conv.weight"] = (8, 8, 3, 3)
        d[p + "conv1.conv.weight"] = (16, 8, 3, 3)
        d[p + "conv2.conv.weight"] = (32, 16, 3, 3)
        d[p + "conv3.weight"] = (32, 16, 3, 3)   # ConvTranspose2d: (in, out, k, k)
        d[p + "conv4.weight"] = (16, 8, 3, 3)
        d[p + "conv5.weight"] = (1, 8, 3, 3)
        d[p + "conv5.bias"] = (1,)

    up = "iter_mvs.update."
    for gate in ("convz", "convr", "convq"):
        d[f"{up}gru.{gate}.weight"] = (32, 43, 3, 3)
        d[f"{up}gru.{gate}.bias"] = (32,)
    d[up + "depth_head.0.weight"] = (32, 32, 3, 3)
    d[up + "depth_head.2.weight"] = (64, 32, 1, 1)
    d[up + "depth_head.4.weight"] = (256, 64, 1, 1)
    d[up + "depth_head.4.bias"] = (256,)
    d[up + "confidence_head.0.weight"] = (32, 32, 3, 3)
    d[up + "confidence_head.2.weight"] = (1, 32, 1, 1)
    d[up + "confidence_head.2.bias"] = (1,)
    d[up + "hidden_init_head.0.weight"] = (64, 32, 3, 3)
    d[up + "hidden_init_head.2.weight"] = (32, 64, 1, 1)
    d[up + "hidden_init_head.2.bias"] = (32,)
    d["iter_mvs.upsample.0.weight"] = (64, 32, 3, 3)
    d["iter_mvs.upsample.2.weight"] = (144, 64, 1, 1)
    return d


def strip_module_prefix(state: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept checkpoints saved from a ``DataParallel`` wrapper (eval.py:119,125)."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}


def check_state_dict(state: Mapping[str, torch.Tensor]) -> None:
    """Raise ``KeyError`` / ``ValueError`` unless ``state`` matches the schema exactly
    (the strict load of eval.py:125)."""
    schema = state_dict_schema()
    missing = [k for k in schema if k not in state]
    extra = [k for k in state if k not in schema]
    if missing or extra:
        raise KeyError(f"state_dict mismatch: missing={missing[:4]}... extra={extra[:4]}...")
    for k, shp in schema.items():
        if tuple(state[k].shape) != tuple(shp):
            raise ValueError(f"{k}: expected shape {shp}, got {tuple(state[k].shape)}")
