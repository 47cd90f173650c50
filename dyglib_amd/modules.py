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


class _MergeFunction(torch.autograd.Function):
    """logits = fc2(relu(fc1(cat(x1, x2)))) for output_dim 1 with its backward pass on the HIP library: one launch forward
    (dygnn_merge_layer_logits), one launch backward (dygnn_merge_layer_backward: two launches on the matrix cores; the hidden layer is recomputed, nothing is kept)."""

    @staticmethod
    def forward(ctx, a, b, w1, b1, w2, b2):
        lib = _capi.load()
        a, b = a.contiguous().float(), b.contiguous().float()
        out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
        _capi.check(lib.dygnn_merge_layer_logits(a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], w1.shape[0], w1.data_ptr(), b1.data_ptr(),
                                                 w2.data_ptr(), b2.data_ptr(), out.data_ptr(), _capi.current_stream_ptr()))
        ctx.save_for_backward(a, b, w1, b1, w2)
        return out.unsqueeze(-1)

    @staticmethod
    def backward(ctx, g):
        a, b, w1, b1, w2 = ctx.saved_tensors
        lib = _capi.load()
        g = g.reshape(-1).contiguous().float()
        da, db = torch.empty_like(a), torch.empty_like(b)
        n, hidden = a.shape[0], w1.shape[0]
        flat = torch.zeros(w1.numel() + b1.numel() + w2.numel() + 1, dtype=torch.float32, device=a.device)      # accumulated into (atomics)
        work = torch.empty(n * hidden, dtype=torch.float32, device=a.device)
        o1, o2, o3 = w1.numel(), w1.numel() + b1.numel(), w1.numel() + b1.numel() + w2.numel()
        _capi.check(lib.dygnn_merge_layer_backward(a.data_ptr(), b.data_ptr(), n, a.shape[1], hidden, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                                   g.data_ptr(), da.data_ptr(), db.data_ptr(), flat.data_ptr(), flat[o1:].data_ptr(), flat[o2:].data_ptr(),
                                                   flat[o3:].data_ptr(), work.data_ptr(), _capi.current_stream_ptr()))
        return da, db, flat[:o1].view_as(w1), flat[o1:o2].view_as(b1), flat[o2:o3].view_as(w2), flat[o3:o3 + 1]


class MergeLayer(nn.Module):
    """Link predictor fc2(relu(fc1(cat(x1,x2)))) (models/modules.py:42-68).  `forward` keeps the
    reference's signature and returns logits [n, output_dim]: on the GPU with output_dim 1 (the link
    predictor of train_link_prediction.py:124) through the HIP library, forward and backward one launch
    each; otherwise (a CPU copy, several outputs) by PyTorch ops.  `link_probabilities` is the fused head
    of the evaluation step: sigmoid(forward(x1,x2)).squeeze(-1) in one launch (output_dim must be 1)."""

    def __init__(self, input_dim1: int, input_dim2: int, hidden_dim: int, output_dim: int):
        super().__init__()
        self.fc1 = nn.Linear(input_dim1 + input_dim2, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, output_dim)
        self.act = nn.ReLU()

    def forward(self, input_1: torch.Tensor, input_2: torch.Tensor):
        # the library's backward (dygnn_merge_layer_backward) holds the hidden layer in LDS: hidden % 4 == 0 and hidden <= 192 — any other
        # link predictor takes the PyTorch ops in BOTH directions (a forward through the library would fail inside loss.backward())
        if (input_1.is_cuda and self.fc2.out_features == 1 and input_1.dim() == 2 and input_1.shape == input_2.shape and input_1.shape[1] % 4 == 0
                and 2 * input_1.shape[1] == self.fc1.in_features and self.fc1.weight.is_cuda and self.fc1.weight.dtype == torch.float32
                and self.fc1.out_features % 4 == 0 and self.fc1.out_features <= 192):
            return _MergeFunction.apply(input_1, input_2, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
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
