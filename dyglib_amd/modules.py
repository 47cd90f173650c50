"""Parameter containers with the reference's names/shapes/initialisation (models/modules.py:7-68),
so reference checkpoints (`state_dict`, utils/EarlyStopping.py:65-86) load unchanged.  The math
runs in the HIP library; these classes only own the parameters."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import _capi


class TimeEncoder(nn.Module):
    """cos(w * dt + b), w initialised to 10^-linspace(0,9,dim), b = 0 (models/modules.py:9-25).
    Evaluated inside the fused kernels; no forward() on purpose."""

    def __init__(self, time_dim: int, parameter_requires_grad: bool = True):
        super().__init__()
        self.time_dim = time_dim
        self.w = nn.Linear(1, time_dim)
        self.w.weight = nn.Parameter(torch.from_numpy(1 / 10 ** np.linspace(0, 9, time_dim, dtype=np.float32)).reshape(time_dim, -1))
        self.w.bias = nn.Parameter(torch.zeros(time_dim))
        if not parameter_requires_grad:
            self.w.weight.requires_grad = False
            self.w.bias.requires_grad = False


class MergeLayer(nn.Module):
    """Link predictor fc2(relu(fc1(cat(x1,x2)))) (models/modules.py:42-68).  `forward` keeps the
    reference's signature and returns logits [n, output_dim] computed by PyTorch-ROCm ops (it is the
    caller's module, outside the hot path); `link_probabilities` is the fused HIP head used by the
    evaluation step: sigmoid(forward(x1,x2)).squeeze(-1) in one launch (output_dim must be 1)."""

    def __init__(self, input_dim1: int, input_dim2: int, hidden_dim: int, output_dim: int):
        super().__init__()
        self.fc1 = nn.Linear(input_dim1 + input_dim2, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, output_dim)
        self.act = nn.ReLU()

    def forward(self, input_1: torch.Tensor, input_2: torch.Tensor):
        x = torch.cat([input_1, input_2], dim=1)
        return self.fc2(self.act(self.fc1(x)))

    @torch.no_grad()
    def link_probabilities(self, input_1: torch.Tensor, input_2: torch.Tensor) -> torch.Tensor:
        if self.fc2.out_features != 1 or input_1.shape != input_2.shape:
            raise AssertionError("link_probabilities needs output_dim == 1 and equally shaped inputs")
        if not input_1.is_cuda:
            raise _capi.DygnnError("link_probabilities runs on the GPU only")
        lib = _capi.load()
        a, b = input_1.contiguous().float(), input_2.contiguous().float()
        out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
        _capi.check(lib.dygnn_merge_layer_sigmoid(a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], self.fc1.out_features,
                                                  self.fc1.weight.data_ptr(), self.fc1.bias.data_ptr(),
                                                  self.fc2.weight.data_ptr(), self.fc2.bias.data_ptr(), out.data_ptr(),
                                                  _capi.current_stream_ptr()))
        return out
