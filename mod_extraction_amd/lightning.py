"""Host mirror of mod_extraction/lightning.py: the step logic of ``LFOExtraction``
(lightning.py:65-199) and ``TBPTTLFOEffectModeling`` (lightning.py:202-431) with the same
constructor arguments, batch 4-tuple ``(dry, wet, mod_sig, fx_params)`` and metric names
(``train/l1`` ... ``val/loss``).  ``pytorch_lightning`` is replaced by ``trainer.Trainer`` (a thin
DDP loop); these classes are plain ``nn.Module``s that record their metrics in ``self.logged``.

Everything stays on the device: the reference's per-step ``.detach().float().cpu()`` copies of all
tensors (lightning.py:132-141) are dropped -- ``data_dict`` holds device tensors.
"""
import logging
import os
from collections import defaultdict
from typing import Dict, List, Optional

import torch
from torch import Tensor as T, nn

from . import losses as L
from .models import HiddenStateModel, RandomLFO
from .modulations import find_valid_mod_sig_indices, smoothen, stretch_corners, valid_mod_sig_mask
from .util import linear_interpolate_last_dim

log = logging.getLogger(__name__)
log.setLevel(level=os.environ.get("LOGLEVEL", "INFO"))


class BaseLightingModule(nn.Module):
    default_loss_dict = {"l1": 1.0, "mse": 0.0}

    def __init__(self, loss_dict: Optional[Dict[str, float]] = None) -> None:
        super().__init__()
        if loss_dict is None:
            loss_dict = self.default_loss_dict
        self.loss_dict = dict(loss_dict)
        self._fused_lfo = all(k in ("l1", "fdl1", "sdl1", "mse") for k in self.loss_dict)
        self.loss_funcs = nn.ModuleList([] if self._fused_lfo else
                                        [L.get_loss_func_by_name(name) for name in self.loss_dict])
        self.logged: Dict[str, List[T]] = defaultdict(list)

    def log(self, name: str, value: T) -> None:
        """Stands in for LightningModule.log(on_epoch=True, sync_dist=True): values are kept as
        device scalars; the trainer reduces them to epoch means (and all-reduces across ranks)."""
        self.logged[name].append(value.detach())

    def calc_and_log_losses(self, y_hat: T, y: T, prefix: str, should_log: bool = True) -> T:
        """lightning.py:33-62: weighted sum of the named losses; zero-weight terms are only logged."""
        if self._fused_lfo:
            loss, terms = L.lfo_loss(y_hat, y, self.loss_dict)
        else:
            terms = {name: f(y_hat, y) for name, f in zip(self.loss_dict, self.loss_funcs)}
            loss = None
            for name, w in self.loss_dict.items():
                if w > 0:
                    loss = w * terms[name] if loss is None else loss + w * terms[name]
        if should_log:
            for name in self.loss_dict:
                self.log(f"{prefix}/{name}", terms[name])
            self.log(f"{prefix}/loss", loss)
        return loss


class LFOExtraction(BaseLightingModule):
    def __init__(self,
                 model: nn.Module,
                 sr: float = 44100,
                 use_dry: bool = True,
                 model_smooth_n_frames: int = 4,
                 should_stretch: bool = False,
                 max_n_corners: int = 16,
                 stretch_smooth_n_frames: int = 0,
                 sub_batch_size: Optional[int] = None,
                 loss_dict: Optional[Dict[str, float]] = None) -> None:
        super().__init__(loss_dict)
        self.model = model
        self.sr = sr
        self.use_dry = use_dry
        self.model_smooth_n_frames = model_smooth_n_frames
        self.should_stretch = should_stretch
        self.max_n_corners = max_n_corners
        self.stretch_smooth_n_frames = stretch_smooth_n_frames
        self.sub_batch_size = sub_batch_size

    @staticmethod
    def center_crop_mod_sig(mod_sig: T, size: int) -> T:
        if size == mod_sig.size(-1):
            return mod_sig
        assert size < mod_sig.size(-1)
        padding = mod_sig.size(-1) - size
        pad_l = padding // 2
        return mod_sig[..., pad_l:pad_l + size]

    def common_step(self, batch, is_training: bool):
        """lightning.py:96-158."""
        prefix = "train" if is_training else "val"
        dry, wet, mod_sig, fx_params = batch
        if isinstance(self.model, RandomLFO):
            mod_sig_hat = self.model(wet.size(0), fx_params).to(wet.device)
        elif self.use_dry:
            assert dry is not None
            mod_sig_hat, _ = self.model(torch.cat([dry, wet], dim=1))
        else:
            mod_sig_hat, _ = self.model(wet)
        mod_sig_hat = mod_sig_hat.squeeze(1)
        if mod_sig is None:
            mod_sig = torch.zeros_like(mod_sig_hat)
        else:
            mod_sig = linear_interpolate_last_dim(mod_sig, mod_sig_hat.size(-1), align_corners=True)
        assert mod_sig.shape == mod_sig_hat.shape
        if self.model_smooth_n_frames > 1:
            if mod_sig_hat.requires_grad:
                # training with smoothing is not used by any shipped config; keep autograd correct
                mod_sig_hat = mod_sig_hat.unfold(-1, self.model_smooth_n_frames, 1).mean(-1)
            else:
                mod_sig_hat = smoothen(mod_sig_hat, self.model_smooth_n_frames)
            mod_sig = self.center_crop_mod_sig(mod_sig, mod_sig_hat.size(-1))
        if self.should_stretch:
            mod_sig_hat = stretch_corners(mod_sig_hat.detach(), max_n_corners=self.max_n_corners,
                                          smooth_n_frames=self.stretch_smooth_n_frames)
            if self.stretch_smooth_n_frames > 1:
                mod_sig = self.center_crop_mod_sig(mod_sig, mod_sig_hat.size(-1))
        assert mod_sig.shape == mod_sig_hat.shape
        loss = self.calc_and_log_losses(mod_sig_hat, mod_sig.contiguous(), prefix)
        data_dict = {"wet": wet.detach(), "mod_sig": mod_sig.detach(), "mod_sig_hat": mod_sig_hat.detach()}
        if dry is not None:
            data_dict["dry"] = dry.detach()
        return loss, data_dict, fx_params

    def sub_batch_size_common_step(self, batch, is_training: bool):
        """lightning.py:160-185."""
        dry, wet, mod_sig, fx_params = batch
        bs = mod_sig.size(0)
        assert bs >= self.sub_batch_size and bs % self.sub_batch_size == 0
        losses, out = [], None
        for s in range(0, bs, self.sub_batch_size):
            e = s + self.sub_batch_size
            sub = (None if dry is None else dry[s:e], wet[s:e], mod_sig[s:e],
                   {k: v[s:e] for k, v in fx_params.items()})
            out = self.common_step(sub, is_training=is_training)
            losses.append(out[0])
        return torch.stack(losses, dim=0).mean(dim=0), out[1], out[2]

    def training_step(self, batch, batch_idx: int = 0) -> T:
        step = self.common_step if self.sub_batch_size is None else self.sub_batch_size_common_step
        return step(batch, is_training=True)[0]

    def validation_step(self, batch, batch_idx: int = 0):
        step = self.common_step if self.sub_batch_size is None else self.sub_batch_size_common_step
        with torch.no_grad():
            return step(batch, is_training=False)
