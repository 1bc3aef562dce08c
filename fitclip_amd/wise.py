"""WiSE-FT weight-space ensemble on the ROCm device.

Drop-in for the two functions of the reference's `aligner/wise.py` (`wise_state_dict` :10-16, `wise` :19-23; Hydra
target `config/encoder/wise.yaml:6-9`): theta = (1 - w) * theta_1 + w * theta_2 over `named_parameters()`, with the
same `AssertionError`s for mismatching key sets / model types.  The blend itself is `fc_wise`, one HBM-bound axpby
launch per tensor (12 bytes of traffic per parameter), bit-identical to the torch expression.
"""
from __future__ import annotations

import copy
from collections import OrderedDict
from typing import Mapping, TypeVar

import torch
from torch import nn

from . import ops

ModuleT = TypeVar("ModuleT", bound=nn.Module)


def _params(model: nn.Module) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((name, p.detach()) for name, p in model.named_parameters())


def wise_state_dict(model1: ModuleT, model2: ModuleT, weight_for_2: float = 0.5) -> Mapping[str, torch.Tensor]:
    first, second = _params(model1), _params(model2)
    assert set(first) == set(second), "the two models must have the same parameter names"
    blended: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, theta1 in first.items():
        theta2 = second[name]
        assert theta1.shape == theta2.shape, f"shape mismatch for {name}"
        blended[name] = ops.wise_axpby(theta1.contiguous(), theta2.contiguous(), weight_for_2)
    return blended


def wise(model1: ModuleT, model2: ModuleT, weight_for_2: float = 0.5, copy_model1: bool = True) -> ModuleT:
    assert type(model1) is type(model2), "WiSE needs two models of the same class"
    ensemble = copy.deepcopy(model1) if copy_model1 else copy.deepcopy(model2)
    ensemble.load_state_dict(wise_state_dict(model1, model2, weight_for_2=weight_for_2))
    return ensemble
