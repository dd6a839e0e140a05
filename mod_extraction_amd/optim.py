"""AdamW on one flat fp32 buffer (K12, ``mx_adamw_step``), mirroring ``torch.optim.AdamW`` as the
reference configures it (configs/opt/adam_w.yml: lr 1e-4, betas (0.8, 0.99); torch defaults
eps 1e-8, weight_decay 0.01).

All trainable parameters of the module are re-homed as views into one contiguous buffer, and their
``.grad`` as views into a second one, so that a DDP step is ONE RCCL all-reduce over the flat
gradient followed by ONE kernel launch (the reference's Lightning/DDP path buckets per tensor).
"""
import contextlib
from typing import Dict, Iterable, Iterator, List, Tuple

import torch
from torch import Tensor as T, nn

from . import _hip


class FlatAdamW:
    def __init__(self, params: Iterable[nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.01) -> None:
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        assert all(p.device == dev and p.dtype == torch.float32 for p in self.params)
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), \
            float(weight_decay)
        n = sum(p.numel() for p in self.params)
        self.flat_param = torch.empty(n, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        self.step_count = 0
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + k].view(p.shape)
            p.grad = self.flat_grad[off:off + k].view(p.shape)
            off += k

    @property
    def numel(self) -> int:
        return self.flat_param.numel()

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.flat_grad.zero_()
        self.flat_grad._modex_fresh = False     # in-place gradient writes are armed only by direct_backward() below
        off = 0
        for p in self.params:          # re-attach views if something replaced .grad
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad[off:off + k].data_ptr():
                p.grad = self.flat_grad[off:off + k].view(p.shape)
            off += k

    @contextlib.contextmanager
    def direct_backward(self) -> Iterator[None]:
        """Scope of ONE ``loss.backward()`` that may WRITE its parameter gradients into the flat buffer instead of
        accumulating (``models._direct_grad_views``).  The caller promises that the buffer holds nothing it wants to keep
        -- i.e. this is the first backward after ``zero_grad()`` -- and that neither ``torch.autograd.grad`` nor parameter
        hooks are used for that call (autograd is handed ``None`` for those inputs).  The flag is consumed by the first
        CNN backward inside the scope and is always cleared on exit, so a backward outside the scope (a weight penalty, a
        second sub-batch, a manual add into ``flat_grad``) takes the ordinary accumulate path."""
        self.flat_grad._modex_fresh = True
        try:
            yield
        finally:
            self.flat_grad._modex_fresh = False

    def step(self, grad_scale: float = 1.0) -> None:
        self.step_count += 1
        _hip.call("mx_adamw_step", _hip.ptr(self.flat_param), _hip.ptr(self.flat_grad), _hip.ptr(self.exp_avg),
                  _hip.ptr(self.exp_avg_sq), self.numel, self.step_count, self.lr, self.betas[0], self.betas[1],
                  self.eps, self.weight_decay, float(grad_scale), _hip.stream())

    def step_from_rows(self, part: torch.Tensor, grad_scale: float = 1.0) -> None:
        """``flat_grad = part.sum(0)`` (one gradient row per clip, the order of ``mx_reduce_rows``) and the AdamW step in ONE
        launch -- the TBPTT loop of the effect model takes 83 optimizer steps per batch on 17 473 parameters.  Bit-identical
        to ``mx_reduce_rows`` followed by ``step()``."""
        assert part.dim() == 2 and part.size(1) == self.numel and part.is_contiguous() and part.dtype == torch.float32
        self.step_count += 1
        _hip.call("mx_reduce_rows_adamw_step", _hip.ptr(part), part.size(0), _hip.ptr(self.flat_param), _hip.ptr(self.flat_grad),
                  _hip.ptr(self.exp_avg), _hip.ptr(self.exp_avg_sq), self.numel, self.step_count, self.lr, self.betas[0],
                  self.betas[1], self.eps, self.weight_decay, float(grad_scale), _hip.stream())

    def state_dict(self) -> Dict[str, object]:
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
