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
from .modulations import (find_valid_mod_sig_indices, smoothen, smoothen_bwd, smoothen_with_grad, stretch_corners, stretch_corners_bwd,
                          valid_mod_sig_mask)
from .util import linear_interpolate_last_dim, linear_interpolate_last_dim_bwd

log = logging.getLogger(__name__)
log.setLevel(level=os.environ.get("LOGLEVEL", "INFO"))


def stack_dry_wet(dry: T, wet: T) -> T:
    """``tr.cat([dry, wet], dim=1)`` (lightning.py:109,256).  The device batcher renders dry and wet as the two channels of
    ONE (B, 2, N) buffer: then the concatenation already exists and a view of it is returned (no 180 MB copy per step)."""
    B, C, N = dry.shape
    if (C == 1 and wet.shape == dry.shape and dry.dtype == wet.dtype and dry.stride() == wet.stride() == (2 * N, N, 1)
            and wet.data_ptr() == dry.data_ptr() + N * dry.element_size()
            and dry.untyped_storage().data_ptr() == wet.untyped_storage().data_ptr()):
        return torch.as_strided(dry, (B, 2, N), (2 * N, N, 1))
    return torch.cat([dry, wet], dim=1)


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
            mod_sig_hat, _ = self.model(stack_dry_wet(dry, wet))
        else:
            mod_sig_hat, _ = self.model(wet)
        mod_sig_hat = mod_sig_hat.squeeze(1)
        if mod_sig is None:
            mod_sig = torch.zeros_like(mod_sig_hat)
        else:
            mod_sig = linear_interpolate_last_dim(mod_sig, mod_sig_hat.size(-1), align_corners=True)
        assert mod_sig.shape == mod_sig_hat.shape
        if self.model_smooth_n_frames > 1:
            if mod_sig_hat.requires_grad:       # training with smoothing (no shipped config): the kernel and its transpose as one autograd node
                mod_sig_hat = smoothen_with_grad(mod_sig_hat, self.model_smooth_n_frames)
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


def _channel_rows(t: T) -> T:
    """(B, C, T) -> (B C, 1, T): the effect-model losses reduce over clips AND channels alike (losses.py:33-38,61-66: a mean over
    (batch, channel) of per-row ratios; nn.L1Loss: a mean over everything)."""
    return t if t.size(1) == 1 else t.contiguous().view(-1, 1, t.size(-1))


class TBPTTLFOEffectModeling(BaseLightingModule):
    """lightning.py:202-431: frozen LFO-net -> smooth / stretch / crop -> discard invalid LFOs -> LSTM
    effect model trained with truncated BPTT (1024-sample warm-up, then one optimizer step per
    1024-sample chunk).  ``automatic_optimization`` is False like in the reference: ``training_step``
    receives the optimizer and runs the 83 inner steps itself; under DDP every inner step is one
    all-reduce of the 70 KB flat gradient, and a rank without any valid LFO still takes part with
    zero gradients (the reference would return None there and dead-lock DDP)."""
    default_loss_dict = {"l1": 1.0, "esr": 0.0, "dc": 0.0}

    def __init__(self,
                 warmup_n_samples: int,
                 step_n_samples: int,
                 effect_model: HiddenStateModel,
                 lfo_model: Optional[nn.Module] = None,
                 lfo_model_weights_path: Optional[str] = None,
                 freeze_lfo_model: bool = True,
                 param_model: Optional[nn.Module] = None,
                 sr: float = 44100,
                 use_dry: bool = True,
                 model_smooth_n_frames: int = 8,
                 should_stretch: bool = True,
                 max_n_corners: int = 16,
                 stretch_smooth_n_frames: int = 0,
                 discard_invalid_lfos: bool = True,
                 loss_dict: Optional[Dict[str, float]] = None) -> None:
        super().__init__(loss_dict)
        assert warmup_n_samples > 0
        # param_model (lightning.py:344-347,371-375): any nn.Module wet (B, C, n) -> (B, P); its output is repeated over time and
        # concatenated to the LFO as extra latent channels (the effect model then has latent_dim = 1 + P, i.e. the general LSTM
        # of lstm_generic.py, an autograd node -- the step below runs such models through `_general_train_chunk`).
        # freeze_lfo_model: false (lightning.py:258,344-366): the extractor is re-run inside every TBPTT step and trained through the
        # effect model -- CNN -> moving average -> stretch_corners -> resampling -> LSTM, every stage with a backward kernel
        # (common_step below, `relearn`).
        from .effect_losses import GRAD_NAMES
        for name, w in self.loss_dict.items():
            if w > 0 and name not in GRAD_NAMES:
                raise NotImplementedError(f"effect-model loss '{name}' has no gradient kernel (supported: {GRAD_NAMES})")
        # only nn.L1Loss weighted (every shipped config): the BPTT kernel evaluates its gradient itself
        self._fused_l1 = all(w <= 0 or name == "l1" for name, w in self.loss_dict.items())
        self._mrstft = None
        self._extra_losses = {}           # loss modules outside effect_loss_terms (mrstft, ...), built once, by name
        self.warmup_n_samples, self.step_n_samples = warmup_n_samples, step_n_samples
        self.effect_model = effect_model
        self.lfo_model_weights_path = lfo_model_weights_path
        self.freeze_lfo_model = freeze_lfo_model
        self.param_model = param_model
        self.sr, self.use_dry = sr, use_dry
        self.model_smooth_n_frames = model_smooth_n_frames
        self.should_stretch, self.max_n_corners = should_stretch, max_n_corners
        self.stretch_smooth_n_frames = stretch_smooth_n_frames
        self.discard_invalid_lfos = discard_invalid_lfos
        if lfo_model is not None:
            if lfo_model_weights_path is not None:
                log.info("Loading LFO model weights")
                assert os.path.isfile(lfo_model_weights_path)
                lfo_model.load_state_dict(torch.load(lfo_model_weights_path, map_location="cpu"))
            if freeze_lfo_model:
                log.info("Freezing LFO model")
                lfo_model.eval()
                for p in lfo_model.parameters():
                    p.requires_grad = False
        else:
            log.info("Using ground truth mod_sig")
        self.lfo_model = lfo_model
        self.automatic_optimization = False
        self.use_gt_mod_sig = lfo_model is None

    def train(self, mode: bool = True):
        super().train(mode)
        if self.lfo_model is not None and self.freeze_lfo_model:
            self.lfo_model.eval()               # frozen extractor stays in eval mode (lightning.py:243-244)
        return self

    center_crop_mod_sig = staticmethod(LFOExtraction.center_crop_mod_sig)

    def extract_mod_sig(self, wet: T, mod_sig: Optional[T] = None, fx_params=None):
        """lightning.py:254-272."""
        with torch.no_grad():
            if self.lfo_model is None:
                assert mod_sig is not None and mod_sig.ndim == 2
                mod_sig_hat = mod_sig
            elif isinstance(self.lfo_model, RandomLFO):
                mod_sig_hat = self.lfo_model(wet.size(0), fx_params).squeeze(1).to(wet.device)
            else:
                mod_sig_hat, _ = self.lfo_model(wet)
                mod_sig_hat = mod_sig_hat.squeeze(1)
            if mod_sig is not None and mod_sig.size(-1) != mod_sig_hat.size(-1):
                mod_sig = linear_interpolate_last_dim(mod_sig, mod_sig_hat.size(-1), align_corners=True)
            return mod_sig_hat, mod_sig

    def smooth_stretch_crop_mod_sig(self, mod_sig_hat: T, mod_sig: Optional[T] = None):
        """lightning.py:284-300."""
        orig = mod_sig_hat.size(-1)
        if self.model_smooth_n_frames > 1:
            mod_sig_hat = smoothen(mod_sig_hat, self.model_smooth_n_frames)
            if mod_sig is not None:
                mod_sig = self.center_crop_mod_sig(mod_sig, mod_sig_hat.size(-1))
        if self.should_stretch:
            mod_sig_hat = stretch_corners(mod_sig_hat, max_n_corners=self.max_n_corners,
                                          smooth_n_frames=self.stretch_smooth_n_frames)
            if self.stretch_smooth_n_frames > 1 and mod_sig is not None:
                mod_sig = self.center_crop_mod_sig(mod_sig, mod_sig_hat.size(-1))
        return mod_sig_hat, mod_sig, orig - mod_sig_hat.size(-1)

    def _prepare_all_rows(self, batch):
        """lightning.py:310-326 for EVERY clip of the batch, without a host synchronisation: extractor forward, smooth /
        stretch / crop, and the validity verdict of each row as a device tensor (None when nothing is discarded)."""
        dry, wet, mod_sig, fx_params = batch
        assert dry.size(-1) == wet.size(-1) >= self.warmup_n_samples + self.step_n_samples
        lfo_in = stack_dry_wet(dry, wet) if self.use_dry else wet
        mod_sig_hat, mod_sig = self.extract_mod_sig(lfo_in, mod_sig, fx_params)
        mod_sig_hat, mod_sig, removed = self.smooth_stretch_crop_mod_sig(mod_sig_hat, mod_sig)
        n_frames = mod_sig_hat.size(-1)
        n_samples = int((n_frames / (n_frames + removed)) * dry.size(-1))
        dry = self.center_crop_mod_sig(dry, n_samples)
        wet = self.center_crop_mod_sig(wet, n_samples)
        valid = valid_mod_sig_mask(mod_sig_hat) if self.discard_invalid_lfos else None
        return dry, wet, mod_sig_hat, mod_sig, valid

    def _select_rows(self, dry, wet, mod_sig_hat, mod_sig, keep):
        """lightning.py:327-337: drop the clips without a valid LFO (``keep``: host index tensor or None = all rows)
        and resample the LFOs to the audio rate."""
        self.last_kept = dry.size(0) if keep is None else int(keep.numel())      # clips that train this batch
        if keep is not None and keep.numel() == 0:
            log.info("No valid LFO signals found")
            return None
        if keep is not None and keep.numel() < dry.size(0):
            # (pinned: a copy from pageable memory holds the HOST until the stream has reached it, i.e. until the whole
            #  previous batch has run -- and the device then idles while the host catches up)
            if dry.is_cuda:
                keep = keep.pin_memory()
            keep = keep.to(dry.device, non_blocking=True)
            dry, wet, mod_sig_hat = dry[keep], wet[keep], mod_sig_hat[keep]
            if mod_sig is not None:
                mod_sig = mod_sig[keep]
        dry, wet = dry.contiguous(), wet.contiguous()
        lfo_sr = linear_interpolate_last_dim(mod_sig_hat, dry.size(-1), align_corners=True).unsqueeze(1)
        return dry, wet, mod_sig_hat, mod_sig, lfo_sr

    def prepare(self, batch):
        """lightning.py:310-337: everything before the LSTM loop.  Returns None if no clip has a valid
        LFO, else (dry, wet, mod_sig_hat, mod_sig, lfo_at_sample_rate (B',1,n'))."""
        dry, wet, mod_sig_hat, mod_sig, valid = self._prepare_all_rows(batch)
        keep = None if valid is None else torch.nonzero(valid.cpu()).view(-1)
        return self._select_rows(dry, wet, mod_sig_hat, mod_sig, keep)

    def prepare_ahead(self, batch):
        """``prepare`` for a data module's ``set_ahead_fn``: everything before the LSTM loop depends only on the batch
        and on the FROZEN extractor, so it can run one batch ahead on the side stream (the reference gets the same
        overlap from its DataLoader workers for the rendering; the extractor forward is prefetched on top).
        It must not block the host -- the caller still has the 83 optimizer steps of the CURRENT batch to enqueue -- so
        the row verdicts of ``discard_invalid_lfos`` travel to pinned host memory asynchronously behind an event, and
        the gather of the surviving rows happens in ``finish_prepare`` when the batch is consumed."""
        dry, wet, mod_sig_hat, mod_sig, valid = self._prepare_all_rows(batch)
        prep = {"dry": dry, "wet": wet, "mod_sig_hat": mod_sig_hat, "mod_sig": mod_sig, "valid_host": None, "ready": None}
        if valid is not None:
            host = torch.empty(valid.shape, dtype=valid.dtype, pin_memory=True)
            host.copy_(valid, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(valid.device))
            prep["valid_host"], prep["ready"], prep["valid"] = host, ev, valid
        return prep

    def finish_prepare(self, prep):
        """Second half of a prefetched ``prepare``: wait for the (long finished) verdict copy, gather the valid rows on
        the consuming stream.  Returns what ``prepare`` returns."""
        keep = None
        if prep["valid_host"] is not None:
            prep["ready"].synchronize()
            keep = torch.nonzero(prep["valid_host"]).view(-1)
        return self._select_rows(prep["dry"], prep["wet"], prep["mod_sig_hat"], prep["mod_sig"], keep)

    def common_step(self, batch, is_training: bool, optimizer=None, world_size: int = 1, prep=None):
        """lightning.py:302-419."""
        from .effect_losses import effect_loss_grad, effect_loss_terms
        from .trainer import allreduce_flat_grad
        prefix = "train" if is_training else "val"
        prep = self.prepare(batch) if prep is None else self.finish_prepare(prep)
        n_chunks_max = (batch[0].size(-1) - self.warmup_n_samples) // self.step_n_samples
        if prep is None:
            if is_training and world_size > 1:          # stay in lock-step with the other ranks
                for _ in range(n_chunks_max):
                    optimizer.zero_grad()
                    optimizer.step(grad_scale=allreduce_flat_grad(optimizer.flat_grad, world_size))
            return None
        dry, wet, mod_sig_hat, mod_sig, lfo_sr = prep
        em, W, S = self.effect_model, self.warmup_n_samples, self.step_n_samples
        n = dry.size(-1)
        B = dry.size(0)
        # lightning.py:344-349: an UNFROZEN extractor is re-run inside every training step (on the full-length, unfiltered
        # input -- the reference does not re-apply its validity filter there, so it only works when no clip was dropped)
        relearn = (is_training and self.lfo_model is not None and not self.freeze_lfo_model
                   and not isinstance(self.lfo_model, RandomLFO))
        # effect models outside the fused LSTM-64 kernels, and every step with a param_model, go through autograd
        general = bool(getattr(em, "generic", False)) or self.param_model is not None
        lfo_in = None
        if relearn:
            if B != batch[0].size(0):
                raise ValueError("freeze_lfo_model: false re-extracts the LFO of EVERY clip inside the step (lightning.py:344-349): "
                                 "it cannot be combined with clips dropped by discard_invalid_lfos")
            lfo_in = stack_dry_wet(batch[0], batch[1]) if self.use_dry else batch[1]
        from .models import LSTM_NPARAM
        if relearn and not general:
            g_off = (em.lstm.weight_ih_l0.grad.data_ptr() - optimizer.flat_grad.data_ptr()) // 4
            assert em.fc.bias.grad.data_ptr() == optimizer.flat_grad.data_ptr() + 4 * (g_off + LSTM_NPARAM - 1), \
                "the effect model's parameters must be contiguous in the flat gradient (state-dict order)"
            lstm_grad = optimizer.flat_grad[g_off:g_off + LSTM_NPARAM]
        em.clear_hidden()
        with torch.no_grad():
            param_latent = None
            if general:
                if self.param_model is not None:
                    param_latent = self.param_model(wet).unsqueeze(-1)            # lightning.py:344-347
                chunks = [em(dry[:, :, :W], self._with_params(lfo_sr[:, :, :W], param_latent))]
            else:
                chunks = [em.run_chunk(dry[:, :, :W], lfo_sr[:, :, :W])[0]]          # warm-up, no loss
            if is_training:
                em.detach_hidden()
            if is_training and not general:
                stash = torch.empty((B, S, 384), device=dry.device, dtype=torch.float32)
                w_l1 = float(self.loss_dict.get("l1", 0.0))
            done = 0
            for start in range(W, n, S):
                end = start + S
                if end > n:
                    break
                x, lat, tgt = dry[:, :, start:end], lfo_sr[:, :, start:end], wet[:, :, start:end]
                if general and is_training:
                    y, lfo_sr, mod_sig_hat = self._general_train_chunk(x, tgt, wet, lfo_sr, mod_sig_hat, start, end, lfo_in if relearn
                                                                       else None, optimizer, world_size)
                    done += 1
                elif general:
                    y = em(x, self._with_params(lat, param_latent))
                elif relearn:
                    # lightning.py:344-384 with the extractor in the graph: CNN -> moving average -> resampling -> this chunk
                    # of the LFO -> LSTM -> loss; backward in the opposite order, every stage on its own kernel
                    optimizer.zero_grad()
                    with torch.enable_grad():
                        hat, _ = self.lfo_model(lfo_in)
                    hs = smoothen(hat.detach().squeeze(1), self.model_smooth_n_frames)
                    hst = stretch_corners(hs, max_n_corners=self.max_n_corners, smooth_n_frames=self.stretch_smooth_n_frames) \
                        if self.should_stretch else hs
                    n_f = hst.size(-1)
                    lfo_sr = linear_interpolate_last_dim(hst, n, align_corners=True).unsqueeze(1)
                    lat = lfo_sr[:, :, start:end]
                    y, h0, c0 = em.run_chunk(x, lat, stash)
                    if self._fused_l1:
                        dlat = em.bptt_chunk_dlfo(x, lat, y, stash, h0, c0, lstm_grad, wet=tgt, loss_scale=w_l1 / (B * S))
                    else:
                        dy = effect_loss_grad(y, tgt, self.loss_dict, mrstft=self._loss_module("mrstft") if "mrstft" in self.loss_dict else None)
                        dlat = em.bptt_chunk_dlfo(x, lat, y, stash, h0, c0, lstm_grad, dy=dy)
                    d_hs = linear_interpolate_last_dim_bwd(dlat[:, 0, :], n_f, n, start)
                    if self.should_stretch:
                        d_hs = stretch_corners_bwd(hs, d_hs, max_n_corners=self.max_n_corners, smooth_n_frames=self.stretch_smooth_n_frames)
                    d_hs = smoothen_bwd(d_hs, self.model_smooth_n_frames)     # transpose of the moving average
                    hat.backward(d_hs.view_as(hat))
                    optimizer.step(grad_scale=allreduce_flat_grad(optimizer.flat_grad, world_size))
                    em.detach_hidden()
                    mod_sig_hat = hst
                    done += 1
                elif is_training:
                    y, h0, c0 = em.run_chunk(x, lat, stash)
                    # no zero_grad(): the BPTT launch OVERWRITES the whole flat gradient (one fill kernel less per step)
                    if self._fused_l1 and world_size == 1 and optimizer.numel == LSTM_NPARAM:
                        # one process, only the LSTM trained: row sum + AdamW in one launch (bit-identical to the two below)
                        optimizer.step_from_rows(em.bptt_l1_chunk(x, lat, y, tgt, stash, h0, c0, w_l1 / (B * S), None))
                        em.detach_hidden()
                        done += 1
                        chunks.append(y)
                        continue
                    if self._fused_l1:
                        em.bptt_l1_chunk(x, lat, y, tgt, stash, h0, c0, w_l1 / (B * S), optimizer.flat_grad)
                    else:       # lightning.py:380-382 with any loss_dict: d loss / d y from the loss kernels, then BPTT
                        dy = effect_loss_grad(y, tgt, self.loss_dict, mrstft=self._loss_module("mrstft") if "mrstft" in self.loss_dict else None)
                        em.bptt_chunk(x, lat, y, dy, stash, h0, c0, optimizer.flat_grad)
                    optimizer.step(grad_scale=allreduce_flat_grad(optimizer.flat_grad, world_size))
                    em.detach_hidden()
                    done += 1
                else:
                    y = em.run_chunk(x, lat)[0]
                chunks.append(y)
            if is_training and world_size > 1:          # ranks may have cropped differently: pad the step count
                for _ in range(n_chunks_max - done):
                    optimizer.zero_grad()
                    optimizer.step(grad_scale=allreduce_flat_grad(optimizer.flat_grad, world_size))
            wet_hat = torch.cat(chunks, dim=-1)
            m = wet_hat.size(-1)
            dry_c, wet_c, wet_hat = dry[:, :, W:m], wet[:, :, W:m].contiguous(), wet_hat[:, :, W:m].contiguous()
            wet_c = wet_c.expand_as(wet_hat).contiguous() if wet_c.shape != wet_hat.shape else wet_c
            terms = effect_loss_terms(_channel_rows(wet_hat), _channel_rows(wet_c))
            for name in self.loss_dict:
                if name not in terms:
                    terms[name] = self._loss_module(name)(_channel_rows(wet_hat), _channel_rows(wet_c))
            loss = None
            for name, w in self.loss_dict.items():
                self.log(f"{prefix}/{name}", terms[name])
                if w > 0:
                    loss = w * terms[name] if loss is None else loss + w * terms[name]
            self.log(f"{prefix}/loss", loss)
        data_dict = {"dry": dry_c, "wet": wet_c, "wet_hat": wet_hat, "mod_sig_hat": mod_sig_hat}
        if mod_sig is not None:
            data_dict["mod_sig"] = mod_sig
        return loss, data_dict, batch[3]

    @staticmethod
    def _with_params(lfo: T, param_latent: Optional[T]) -> T:
        """lightning.py:345-347,374-375: the param_model's vector repeated over the chunk, after the LFO channel."""
        if param_latent is None:
            return lfo
        return torch.cat([lfo, param_latent.repeat(1, 1, lfo.size(-1))], dim=1)

    def _general_train_chunk(self, x, tgt, wet, lfo_sr, mod_sig_hat, start, end, lfo_in, optimizer, world_size):
        """One TBPTT training step (lightning.py:358-384) with the effect model as an autograd node: any LSTM size, a
        param_model in the graph (re-evaluated on every step like the reference does), and -- ``lfo_in`` given -- the unfrozen
        extractor re-run and trained through the LFO it produces (the backward chain of the fused path: resampling window ->
        stretch_corners -> moving average -> CNN)."""
        from .effect_losses import effect_loss_grad
        from .trainer import allreduce_flat_grad
        em, n = self.effect_model, wet.size(-1)
        optimizer.zero_grad()
        with torch.enable_grad():
            hat = None
            if lfo_in is not None:
                hat, _ = self.lfo_model(lfo_in)
                hs = smoothen(hat.detach().squeeze(1), self.model_smooth_n_frames)
                hst = stretch_corners(hs, max_n_corners=self.max_n_corners, smooth_n_frames=self.stretch_smooth_n_frames) \
                    if self.should_stretch else hs
                lfo_sr = linear_interpolate_last_dim(hst, n, align_corners=True).unsqueeze(1)
                mod_sig_hat = hst
            lat_lfo = lfo_sr[:, :, start:end]
            if hat is not None:
                lat_lfo = lat_lfo.detach().clone().requires_grad_(True)
            p = self.param_model(wet).unsqueeze(-1) if self.param_model is not None else None       # lightning.py:371-373
            y = em(x, self._with_params(lat_lfo, p))
        tgt = tgt.expand_as(y) if tgt.shape != y.shape else tgt
        dy = effect_loss_grad(_channel_rows(y.detach()), _channel_rows(tgt), self.loss_dict,
                              mrstft=self._loss_module("mrstft") if "mrstft" in self.loss_dict else None)
        y.backward(dy.view_as(y))
        if hat is not None:
            d_hs = linear_interpolate_last_dim_bwd(lat_lfo.grad[:, 0, :].contiguous(), hst.size(-1), n, start)
            if self.should_stretch:
                d_hs = stretch_corners_bwd(hs, d_hs, max_n_corners=self.max_n_corners, smooth_n_frames=self.stretch_smooth_n_frames)
            d_hs = smoothen_bwd(d_hs, self.model_smooth_n_frames)
            hat.backward(d_hs.view_as(hat))
        optimizer.step(grad_scale=allreduce_flat_grad(optimizer.flat_grad, world_size))
        em.detach_hidden()
        return y.detach(), lfo_sr, mod_sig_hat

    def _loss_module(self, name: str):
        """One module per loss name for the lifetime of the step object (the MR-STFT module owns window / twiddle tables on
        the device: building it per batch re-uploaded them); `mrstft` is the same object for the gradient and for logging."""
        mod = self._extra_losses.get(name)
        if mod is None:
            mod = self._extra_losses[name] = L.get_loss_func_by_name(name)
            if name == "mrstft":
                self._mrstft = mod
        return mod

    def training_step(self, batch, batch_idx: int = 0, optimizer=None, world_size: int = 1, prep=None):
        assert optimizer is not None, "manual optimisation: pass the FlatAdamW optimizer"
        result = self.common_step(batch, is_training=True, optimizer=optimizer, world_size=world_size, prep=prep)
        return None if result is None else result[0]

    def validation_step(self, batch, batch_idx: int = 0):
        return self.common_step(batch, is_training=False)
