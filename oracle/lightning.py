"""Oracle restatement of the step logic in mod_extraction/lightning.py and of the batch synthesis in
data_modules.py / datasets.py (TEST INFRASTRUCTURE ONLY; torch + numpy + oracle C on the CPU).

* ``synth_batch``        datasets.py:365-398,428-482 + data_modules.py:419-458: LFO labels, flanger /
                         chorus (fx.py) and phaser (pedalboard restatement) rendering of a batch
* ``lfo_common_step``    lightning.py:96-158 (LFOExtraction.common_step) + 33-62 (weighted losses)
* ``lfo_train_step``     one optimisation step with torch.optim.AdamW (configs/opt/adam_w.yml)
"""
import math
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor as T

from . import fx as ofx, losses as olosses, modulations as omod, util as outil


def synth_batch(params: Dict[str, Any], kinds: Sequence[str], src: np.ndarray, n_samples: int, sr: float,
                max_min_ms: Dict[str, float], max_lfo_ms: float = 10.0,
                mod_override: Optional[np.ndarray] = None) -> Tuple[T, T, T]:
    """params: per-clip tensors/lists as produced by SyntheticFxBatcher.sample_params (host side);
    src (B, >= n + lead) source audio.  Returns dry (B,1,N), wet (B,1,N), mod_sig (B, N//100) (the oracle's own LFO labels).  ``mod_override`` (B, N//100): LFO values to feed the flanger/chorus instead
    of the oracle's own (device cos differs from the host's in the last ulp; with the device's LFO as input
    the delay-line output must match bit-for-bit)."""
    B, N = len(kinds), n_samples
    n_lfo, lfo_sr = N // 100, sr // 100
    dry = np.empty((B, N), np.float32)
    wet = np.empty((B, N), np.float32)
    mod = np.empty((B, n_lfo), np.float32)
    for b, kind in enumerate(kinds):
        rate, phase = float(params["rate_hz"][b]), float(params["phase"][b])
        if kind == "phaser":
            lead = int(params["lead"][b])
            full = omod.make_mod_signal(lead + N, sr, rate, math.pi / 2, "cos")           # datasets.py:442
            mod[b] = outil.linear_interpolate_last_dim(full[lead:lead + N], n_lfo).numpy()
            x = src[b:b + 1, :lead + N]
            y = ofx.phaser_np(x, [rate], [float(params["depth"][b])], [float(params["centre_frequency_hz"][b])],
                              [float(params["feedback"][b])], [float(params["mix"][b])], sr)
            dry[b], wet[b] = x[0, lead:], y[0, lead:]
        else:
            m = omod.make_mod_signal(n_lfo, lfo_sr, rate, phase, params["shape"][b], float(params["exp"][b]))
            mod[b] = m.numpy()
            Mm, Ml = ofx.delay_samples(max_min_ms[kind], sr), ofx.delay_samples(max_lfo_ms, sr)
            one = {k: torch.tensor([float(params[k][b])]) for k in ("feedback", "min_delay_width", "width", "depth", "mix")}
            p = ofx.derive_params(1, Mm, Ml, **one)
            dry[b] = src[b, :N]
            lfo = mod[b:b + 1] if mod_override is None else np.ascontiguousarray(mod_override[b:b + 1], np.float32)
            up = outil.linear_interpolate_last_dim_np(lfo, N)                             # data_modules.py:455
            wet[b] = ofx.flanger_np(dry[b:b + 1], up, p, Mm + Ml)[0]
    return torch.from_numpy(dry).unsqueeze(1), torch.from_numpy(wet).unsqueeze(1), torch.from_numpy(mod)


def center_crop(x: T, size: int) -> T:
    if size == x.size(-1):
        return x
    pad = x.size(-1) - size
    lo = pad // 2
    return x[..., lo:lo + size]


def lfo_common_step(model, dry: Optional[T], wet: T, mod_sig: Optional[T], loss_dict: Dict[str, float],
                    use_dry: bool = True, model_smooth_n_frames: int = 0, should_stretch: bool = False,
                    max_n_corners: int = 16, stretch_smooth_n_frames: int = 0,
                    masks=None) -> Tuple[T, Dict[str, T], T]:
    x = torch.cat([dry, wet], dim=1) if use_dry else wet
    y_hat, _ = model(x, masks) if masks is not None else model(x)
    y_hat = y_hat.squeeze(1)
    y = torch.zeros_like(y_hat) if mod_sig is None else outil.linear_interpolate_last_dim(mod_sig, y_hat.size(-1))
    if model_smooth_n_frames > 1:
        y_hat = y_hat.unfold(-1, model_smooth_n_frames, 1).mean(-1)
        y = center_crop(y, y_hat.size(-1))
    if should_stretch:
        y_hat = omod.stretch_corners(y_hat.detach(), max_n_corners, stretch_smooth_n_frames)
        if stretch_smooth_n_frames > 1:
            y = center_crop(y, y_hat.size(-1))
    terms = {k: olosses.get_loss_func_by_name(k)(y_hat, y) for k in loss_dict}
    loss = sum(w * terms[k] for k, w in loss_dict.items() if w > 0)
    return loss, terms, y_hat


def lfo_train_step(model, opt: torch.optim.Optimizer, dry: T, wet: T, mod_sig: T, loss_dict: Dict[str, float],
                   masks=None) -> Tuple[float, Dict[str, float]]:
    opt.zero_grad()
    loss, terms, _ = lfo_common_step(model, dry, wet, mod_sig, loss_dict, masks=masks)
    loss.backward()
    opt.step()
    return float(loss.detach()), {k: float(v.detach()) for k, v in terms.items()}


def tbptt_common_step(effect_model, opt: Optional[torch.optim.Optimizer], dry: T, wet: T, mod_sig_hat: T,
                      warmup_n_samples: int, step_n_samples: int, loss_dict: Dict[str, float],
                      is_training: bool = True, model_smooth_n_frames: int = 8, should_stretch: bool = True,
                      max_n_corners: int = 16, stretch_smooth_n_frames: int = 0,
                      discard_invalid_lfos: bool = True):
    """lightning.py:302-419 (TBPTTLFOEffectModeling.common_step) given the extractor output
    ``mod_sig_hat`` (B, frames) (lightning.py:318 -- the frozen LFO-net, or the ground truth when
    lfo_model is None).  Returns None when no LFO is valid, else a dict with the batch loss terms,
    wet_hat, the processed LFOs, the kept clip indices and the number of optimizer steps taken."""
    n_frames_orig = mod_sig_hat.size(-1)
    # smooth_stretch_crop_mod_sig (lightning.py:284-300)
    if model_smooth_n_frames > 1:
        mod_sig_hat = omod.smoothen(mod_sig_hat, model_smooth_n_frames)
    if should_stretch:
        mod_sig_hat = omod.stretch_corners(mod_sig_hat, max_n_corners, stretch_smooth_n_frames)
    n_frames = mod_sig_hat.size(-1)
    removed = n_frames_orig - n_frames
    n_samples = int((n_frames / (n_frames + removed)) * dry.size(-1))                # lightning.py:321
    dry, wet = center_crop(dry, n_samples), center_crop(wet, n_samples)
    kept = list(range(dry.size(0)))
    if discard_invalid_lfos:
        kept = omod.find_valid_mod_sig_indices(mod_sig_hat)
        if not kept:
            return None
        dry, wet, mod_sig_hat = dry[kept], wet[kept], mod_sig_hat[kept]
    lfo_sr = outil.linear_interpolate_last_dim(mod_sig_hat, dry.size(-1)).unsqueeze(1)   # lightning.py:337
    effect_model.clear_hidden()
    W, S = warmup_n_samples, step_n_samples
    chunks = [effect_model(dry[:, :, :W], lfo_sr[:, :, :W])]
    steps = 0
    if is_training:
        effect_model.detach_hidden()
        opt.zero_grad()
    for start in range(W, dry.size(-1), S):
        end = start + S
        if end > dry.size(-1):
            break
        y = effect_model(dry[:, :, start:end], lfo_sr[:, :, start:end])
        chunks.append(y)
        if is_training:
            tgt = wet[:, :, start:end]
            terms = {k: olosses.get_loss_func_by_name(k)(y, tgt) for k in loss_dict}
            loss = sum(w * terms[k] for k, w in loss_dict.items() if w > 0)
            loss.backward()
            opt.step()
            effect_model.detach_hidden()
            opt.zero_grad()
            steps += 1
    wet_hat = torch.cat(chunks, dim=-1).detach()
    m = wet_hat.size(-1)
    wet_c, wet_hat = wet[:, :, W:m], wet_hat[:, :, W:m]
    terms = {k: olosses.get_loss_func_by_name(k)(wet_hat, wet_c) for k in loss_dict}
    loss = sum(w * terms[k] for k, w in loss_dict.items() if w > 0)
    return {"loss": loss, "terms": terms, "wet_hat": wet_hat, "wet": wet_c, "mod_sig_hat": mod_sig_hat,
            "kept": kept, "steps": steps, "n_samples": n_samples}
