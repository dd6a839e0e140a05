"""Host mirror of the hot-path part of mod_extraction/data_modules.py + datasets.py: the batch
contract ``(dry, wet, mod_sig, fx_params)`` (lightning.py:100,310) and the effect-parameter /
LFO sampling recipes (datasets.py:365-398, 428-482; data_modules.py:419-458), with the effect
rendering moved onto the device:

* flanger / chorus  -> ``mx_flanger_fwd`` (the reference runs fx.py on the CPU in
  ``on_before_batch_transfer`` or pre-renders to disk)
* phaser            -> ``mx_phaser_fwd`` (the reference calls pedalboard in DataLoader workers)
* LFO labels        -> ``mx_lfo_synth``

Audio comes from recorded files when the configured directories exist (``datasets.FileChunkSource``: the
reference's random non-silent chunk search, datasets.py:151-213, feeding whole batches of chunks to the
device renderer), otherwise from a synthetic clip source (peak-normalised uniform noise, seeded per
rank -- the IDMT / EGFx datasets are not part of this repository, and the benchmark is synthetic).
Data-module classes keep the reference names and accept the reference ``init_args`` so the shipped
YAML configs instantiate.
"""
import math
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor as T

from . import _hip, fx, util
from .modulations import SHAPE_IDS, make_mod_signals

LFO_SHAPES = ["cos", "rect_cos", "inv_rect_cos", "tri", "saw", "rsaw"]

# parameter ranges of the shipped configs
FLANGER_FX = {"max_min_delay_ms": 1.0, "max_lfo_delay_ms": 10.0, "feedback": (0.0, 0.7),
              "min_delay_width": (0.0, 1.0), "width": (0.25, 1.0), "depth": (0.25, 1.0),
              "mix": (0.25, 1.0)}                               # configs/train_lfo_flanger.yml:52-71
CHORUS_FX = dict(FLANGER_FX, max_min_delay_ms=30.0, min_delay_width=(0.367, 1.0))   # configs/data/gen_idmt_ch.yml:34-51
PHASER_FX = {"rate_hz": (0.5, 3.0), "depth": (0.2, 1.0), "centre_frequency_hz": (70.0, 18000.0),
             "feedback": (0.0, 0.7), "mix": (0.2, 1.0)}        # configs/train_lfo_phaser.yml:33-48
MOD_SIG = {"rate_hz": (0.5, 3.0), "phase": (0.0, 2 * math.pi), "shapes": LFO_SHAPES, "exp": 1.0}


def _rng(v: Any) -> Tuple[float, float]:
    if isinstance(v, dict):
        return float(v["min"]), float(v["max"])
    return float(v[0]), float(v[1])


def _vec(x: Any, dtype: torch.dtype = torch.float32) -> T:
    """util.sample_* return a scalar for n = 1, as the reference's do: the batch-wide draws want a 1-element tensor then"""
    return x.to(dtype) if isinstance(x, torch.Tensor) else torch.tensor([x], dtype=dtype)


def _fx_from_config(cfg: Optional[Dict[str, Any]], default: Dict[str, Any]) -> Dict[str, Any]:
    out = dict(default)
    if cfg:
        for k, v in cfg.items():
            out[k] = _rng(v) if isinstance(v, (dict, list, tuple)) else v
    return out


class SyntheticFxBatcher:
    """Builds one training batch entirely on the device.

    ``kinds`` lists the effect of each slot of the interleave, clip ``i`` gets ``kinds[i % len(kinds)]``
    (datasets.py:79-83): "flanger", "chorus" (fx.py with 1 ms / 30 ms base delay) or "phaser".
    Host RNG: ``torch`` global generator and numpy global RNG (scipy ``loguniform``), as in the
    reference.  ``rng_order="batch"`` (default) draws each parameter once per batch (one vectorised call);
    ``rng_order="reference"`` consumes the host RNG streams in the reference's own order -- per item
    rate / phase / shape (datasets.py:367-372) resp. rate, depth, centre, feedback, mix, crop offset for a phaser
    item (datasets.py:429-431,460-465,444), then the five (B,) effect draws of
    ``FlangerCPUDataModule.on_before_batch_transfer`` (data_modules.py:421-445) -- so a seeded run reproduces the
    reference's parameter stream value for value (tests/test_param_stream.py, golden from the real ``util``).
    """

    def __init__(self, batch_size: int, n_samples: int, sr: float, kinds: Sequence[str], device: torch.device,
                 flanger_fx: Optional[Dict] = None, chorus_fx: Optional[Dict] = None,
                 phaser_fx: Optional[Dict] = None, mod_sig: Optional[Dict] = None, audio_seed: int = 43,
                 peak_db: float = -1.0, fixed_lead: Optional[int] = None, overlap: bool = False,
                 chunk_source: Optional[Any] = None, rng_order: str = "batch") -> None:
        assert rng_order in ("batch", "reference")
        self.rng_order = rng_order
        self.B, self.N, self.sr, self.device = batch_size, n_samples, float(sr), device
        self.chunk_source = chunk_source                             # datasets.FileChunkSource or None (synthetic)
        self.kinds = [kinds[i % len(kinds)] for i in range(batch_size)]
        self.fl = _fx_from_config(flanger_fx, FLANGER_FX)
        self.ch = _fx_from_config(chorus_fx, CHORUS_FX)
        self.ph = _fx_from_config(phaser_fx, PHASER_FX)
        self.ms = dict(MOD_SIG)
        if mod_sig:
            for k, v in mod_sig.items():
                self.ms[k] = _rng(v) if isinstance(v, dict) else v
        self.n_lfo = n_samples // 100                               # datasets.py:382
        self.lfo_sr = self.sr // 100
        # "dry": no effect -- the wet slot carries the untouched chunk and the LFO is a label only (datasets.py:365-398 as
        # RandomAudioChunkAndModSigDataModule uses it, data_modules.py:331-371)
        self.kind_id = torch.tensor([{"flanger": 0, "chorus": 1, "phaser": 2, "dry": 3}[k] for k in self.kinds])
        self.rows_fx = torch.nonzero(self.kind_id < 2).view(-1).to(torch.int32).to(device)
        self.rows_ph = torch.nonzero(self.kind_id == 2).view(-1).to(torch.int32).to(device)
        self._is_ph_dev = (self.kind_id == 2).to(device)
        self.has_ph = bool((self.kind_id == 2).any())
        self.has_fx = bool((self.kind_id < 2).any())
        self.has_dry = bool((self.kind_id == 3).any())
        self.fixed_lead = fixed_lead
        # phaser: the reference renders n + sr/rate samples and crops (datasets.py:428-449)
        self.max_lead = int(self.sr / self.ph["rate_hz"][0] + 0.5) if self.has_ph else 0
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(audio_seed)
        self.peak = 10.0 ** (peak_db / 20.0)
        mm = torch.tensor([fx.delay_samples((self.ch if k == "chorus" else self.fl)["max_min_delay_ms"], self.sr)
                           for k in self.kinds], dtype=torch.float32)
        ml = torch.tensor([fx.delay_samples((self.ch if k == "chorus" else self.fl)["max_lfo_delay_ms"], self.sr)
                           for k in self.kinds], dtype=torch.float32)
        self.max_min_delay = mm.to(device)
        self.max_lfo_delay = ml.to(device)
        self.max_delay = (mm + ml).to(torch.int32).to(device)
        self.max_delay_max = int((mm + ml).max())
        # overlap=True: batches are rendered one step ahead on a side HIP stream into two alternating
        # buffer sets, so the (latency-bound) effect kernels run concurrently with the previous train step
        self.overlap = bool(overlap) and device.type == "cuda"
        n_sets = 2 if self.overlap else 1
        self._src = [torch.empty((batch_size, n_samples + self.max_lead), device=device, dtype=torch.float32)
                     for _ in range(n_sets)]
        self._audio = [torch.empty((batch_size, 2, n_samples), device=device, dtype=torch.float32)
                       for _ in range(n_sets)]
        self._src_cur, self.audio = self._src[0], self._audio[0]
        self.rows_nonph = torch.nonzero(self.kind_id != 2).view(-1).to(torch.int32).to(device)
        self._noise_seed, self._noise_counter = int(audio_seed), 0
        # recorded audio is gathered on the host into a pinned staging buffer and copied over in one piece
        self._host = None
        self._host_ev = None                  # recorded after each async copy out of the pinned staging buffer
        if chunk_source is not None:
            self._host = torch.empty((batch_size, n_samples + self.max_lead), dtype=torch.float32)
            if device.type == "cuda":
                self._host = self._host.pin_memory()
        self._side = torch.cuda.Stream(device=device) if self.overlap else None
        self._pending = None
        self._slot = 0
        # optional work on each freshly rendered batch that does not depend on the trainable weights (the frozen LFO
        # extractor of the effect-modelling step): with overlap it runs one batch ahead on the side stream too
        self.ahead_fn = None
        self.last_ahead = None

    # ---- host-side parameter draws -------------------------------------------------------------
    def _uniform(self, lo: float, hi: float) -> T:
        return _vec(util.sample_uniform(lo, hi, n=self.B))

    def _choose(self, per_kind: Dict[str, Tuple[float, float]]) -> T:
        """one vectorised draw per effect kind, merged by slot"""
        out = torch.zeros(self.B)
        for kid, name in enumerate(("flanger", "chorus")):
            if name in per_kind and bool((self.kind_id == kid).any()):      # no draw for a kind the batch lacks
                lo, hi = per_kind[name]
                v = self._uniform(lo, hi)
                out = torch.where(self.kind_id == kid, v, out)
        return out

    def _sample_params_reference_order(self) -> Dict[str, Any]:
        """The reference's RNG consumption order (see the class docstring)."""
        B = self.B
        rate, phase, shapes = torch.zeros(B, dtype=torch.float64), torch.zeros(B, dtype=torch.float64), []
        ph = {k: torch.zeros(B, dtype=torch.float64) for k in ("depth", "centre", "feedback", "mix")}
        lead = torch.zeros(B, dtype=torch.int64)
        rate_n = torch.zeros(B, dtype=torch.int64)
        for i, kind in enumerate(self.kinds):
            if kind == "phaser":
                rate[i] = util.sample_log_uniform(*self.ph["rate_hz"])                      # datasets.py:429-432
                rate_n[i] = int((self.sr / float(rate[i])) + 0.5)                           # datasets.py:433
                ph["depth"][i] = util.sample_uniform(*self.ph["depth"])                     # datasets.py:460-465
                ph["centre"][i] = util.sample_log_uniform(*self.ph["centre_frequency_hz"])
                ph["feedback"][i] = util.sample_uniform(*self.ph["feedback"])
                ph["mix"][i] = util.sample_uniform(*self.ph["mix"])
                lead[i] = util.randint(0, int(rate_n[i]) + 1) if self.fixed_lead is None else int(self.fixed_lead)
                phase[i] = math.pi / 2                                                       # datasets.py:442
                shapes.append("cos")
            else:
                rate[i] = util.sample_log_uniform(*self.ms["rate_hz"])                      # datasets.py:367-372
                phase[i] = util.sample_uniform(*self.ms["phase"])
                shapes.append(util.choice(list(self.ms["shapes"])))
        is_ph = self.kind_id == 2
        p: Dict[str, Any] = {}
        for name in ("feedback", "min_delay_width", "width", "depth", "mix"):               # data_modules.py:421-445
            p[name] = self._choose({"flanger": self.fl[name], "chorus": self.ch[name]}) if self.has_fx else torch.zeros(B)
        centre = torch.full((B,), 440.0)
        if self.has_ph:
            p["depth"] = torch.where(is_ph, ph["depth"].float(), p["depth"])
            p["feedback"] = torch.where(is_ph, ph["feedback"].float(), p["feedback"])
            p["mix"] = torch.where(is_ph, ph["mix"].float(), p["mix"])
            centre = torch.where(is_ph, ph["centre"].float(), centre)
            p["proc_extra"] = rate_n
        p.update(rate_hz=rate.float(), phase=phase.float(), shape=shapes, exp=torch.full((B,), float(self.ms["exp"])),
                 centre_frequency_hz=centre, lead=lead.to(torch.int32))
        return p

    def sample_params(self) -> Dict[str, Any]:
        if self.rng_order == "reference":
            return self._sample_params_reference_order()
        B = self.B
        rate = _vec(util.sample_log_uniform(*self.ms["rate_hz"], n=B))
        phase = self._uniform(*self.ms["phase"])
        shape_idx = _vec(util.randint(0, len(self.ms["shapes"]), n=B), torch.int64)
        shapes = [self.ms["shapes"][int(i)] for i in shape_idx]
        is_ph = self.kind_id == 2
        p: Dict[str, Any] = {}
        if self.has_fx:
            for name in ("feedback", "min_delay_width", "width", "depth", "mix"):   # data_modules.py:421-445 order
                p[name] = self._choose({"flanger": self.fl[name], "chorus": self.ch[name]})
        else:
            for name in ("feedback", "min_delay_width", "width", "depth", "mix"):
                p[name] = torch.zeros(B)
        lead = torch.zeros(B, dtype=torch.int32)
        centre = torch.full((B,), 440.0)
        if self.has_ph:
            ph_rate = _vec(util.sample_log_uniform(*self.ph["rate_hz"], n=B))   # datasets.py:429-432
            ph_depth = self._uniform(*self.ph["depth"])                           # datasets.py:460-465 order
            centre = _vec(util.sample_log_uniform(*self.ph["centre_frequency_hz"], n=B))
            ph_fb = self._uniform(*self.ph["feedback"])
            ph_mix = self._uniform(*self.ph["mix"])
            rate = torch.where(is_ph, ph_rate, rate)
            phase = torch.where(is_ph, torch.full((B,), math.pi / 2), phase)      # datasets.py:442
            shapes = ["cos" if is_ph[i] else s for i, s in enumerate(shapes)]
            p["depth"] = torch.where(is_ph, ph_depth, p["depth"])
            p["feedback"] = torch.where(is_ph, ph_fb, p["feedback"])
            p["mix"] = torch.where(is_ph, ph_mix, p["mix"])
            rate_n = (self.sr / rate.double() + 0.5).to(torch.int64)              # datasets.py:433
            if self.fixed_lead is None:
                u = torch.rand(B, dtype=torch.float64)
                lead = torch.minimum((u * (rate_n + 1).double()).to(torch.int64), rate_n)   # randint(0, rate_n + 1)
            else:
                lead = torch.full((B,), int(self.fixed_lead), dtype=torch.int64)
            lead = torch.where(is_ph, lead, torch.zeros_like(lead)).to(torch.int32)
            p["proc_extra"] = torch.where(is_ph, rate_n, torch.zeros_like(rate_n))   # the reference renders n + sr/rate
        p.update(rate_hz=rate, phase=phase, shape=shapes, exp=torch.full((B,), float(self.ms["exp"])),
                 centre_frequency_hz=centre, lead=lead)
        return p

    # ---- device-side rendering -------------------------------------------------------------------
    def to_device(self, p: Dict[str, Any]) -> Tuple[Dict[str, T], T]:
        """The host-sampled parameters on the device.  These copies come from pageable host memory, so each of them blocks
        the HOST until the stream it is issued on has reached it: issue them on a stream that is not waiting for anything
        (``_launch_ahead`` does so BEFORE it makes the side stream wait for the train step in flight -- with the copies behind
        that wait the host was held until the step had finished, once per step, and the main stream then idled ~0.85 ms
        while the host caught up)."""
        dev = self.device
        d = {k: v.to(dev) for k, v in p.items() if isinstance(v, torch.Tensor)}
        shape_id = torch.tensor([SHAPE_IDS[s] for s in p["shape"]], dtype=torch.int32, device=dev)
        return d, shape_id

    def render(self, p: Dict[str, Any], dev_params: Optional[Tuple[Dict[str, T], T]] = None) -> Tuple[T, T, T, Dict[str, Any]]:
        dev, B, N = self.device, self.B, self.N
        d, shape_id = self.to_device(p) if dev_params is None else dev_params
        if self.chunk_source is not None:
            # recorded audio: one non-silent chunk per clip (phaser clips: n + sr/rate samples, datasets.py:433-436)
            extra = p.get("proc_extra", torch.zeros(B, dtype=torch.int64))
            if self._host_ev is not None:
                self._host_ev.synchronize()       # the previous async H2D copy must have read the staging buffer
            self.chunk_source.fill(self._host, (extra + N).clamp(max=self._host.size(1)))
            self._src_cur.copy_(self._host, non_blocking=True)
            if dev.type == "cuda":
                self._host_ev = torch.cuda.Event()
                self._host_ev.record(torch.cuda.current_stream(dev))
            self.audio[:, 0, :].copy_(self._src_cur[:, :N])
        else:
            # synthetic dry audio: uniform noise at `peak`, written ONCE where it is read (csrc/noise.hip):
            # clips without a phaser straight into the batch's dry channel, phaser clips into the staging row the phaser takes its
            # lead-in from (N + lead samples of it; the phaser writes the cropped dry clip itself)
            self._noise_counter += 1
            st = _hip.stream()
            if self.rows_nonph.numel():
                _hip.call("mx_uniform_rows", _hip.ptr(self.audio), 2 * N, _hip.ptr(self.rows_nonph), int(self.rows_nonph.numel()),
                          None, N, N, self._noise_seed, self._noise_counter, -self.peak, self.peak, st)
            if self.has_ph:
                _hip.call("mx_uniform_rows", _hip.ptr(self._src_cur), self._src_cur.stride(0), _hip.ptr(self.rows_ph),
                          int(self.rows_ph.numel()), _hip.ptr(d["lead"].to(torch.int32).contiguous()), N, N + self.max_lead,
                          self._noise_seed, self._noise_counter, -self.peak, self.peak, st)
        # LFO labels at n_samples // 100 points
        mod = make_mod_signals(self.n_lfo, self.lfo_sr, d["rate_hz"], d["phase"], shape_id, d["exp"])
        if self.has_ph:
            half_pi = torch.full((B,), math.pi / 2, device=dev)
            mod_ph = make_mod_signals(N, self.sr, d["rate_hz"], half_pi, None, None, d["lead"], n_out=self.n_lfo)
            mod = torch.where(self._is_ph_dev.unsqueeze(1), mod_ph, mod)
        dry, wet = self.audio[:, 0, :], self.audio[:, 1, :]
        if self.has_dry:
            wet.copy_(dry)                           # effect rows are overwritten below
        if self.has_fx:
            consts = {"lfo_scale": (d["width"] * self.max_lfo_delay).contiguous(),
                      "min_delay": (d["min_delay_width"] * self.max_min_delay).contiguous(),
                      "feedback": d["feedback"], "depth": d["depth"], "mix": d["mix"],
                      "one_minus_mix": (1.0 - d["mix"]).contiguous()}
            fx.flanger_forward(dry, mod, consts, self.max_delay, self.max_delay_max, rows=self.rows_fx, out=wet)
        if self.has_ph:
            fx.phaser_forward(self._src_cur, d, d["lead"], self.sr, N, rows=self.rows_ph, out=wet, dry_out=dry)
        fx_params = dict(d)
        fx_params["shape"] = p["shape"]
        return self.audio[:, 0:1, :], self.audio[:, 1:2, :], mod, fx_params

    @property
    def src(self) -> T:
        """(B, N + longest lead) source audio of the batch rendered last (tests / smoke hand it to the oracle): rows without a
        phaser are read back from the dry channel they were generated into."""
        if self.chunk_source is None and self.rows_nonph.numel():
            idx = self.rows_nonph.long()
            self._src_cur[idx, :self.N] = self.audio[idx, 0, :]
        return self._src_cur

    def use_side_stream(self, stream) -> None:
        """render ahead on ``stream`` (e.g. one confined to its own CUs, streams.cu_partition) instead of the private one"""
        if self.overlap:
            self._side = stream

    def _launch_ahead(self) -> None:
        """render the next batch on the side stream into the idle buffer set"""
        main = torch.cuda.current_stream(self.device)
        slot = self._slot
        self._slot ^= 1
        self._src_cur, self.audio = self._src[slot], self._audio[slot]
        p = self.sample_params()
        with torch.cuda.stream(self._side), torch.no_grad():
            dev_params = self.to_device(p)      # host-blocking copies: before the wait below (see to_device)
        self._side.wait_stream(main)            # the idle set was last read two steps ago; also orders the RNG state
        with torch.cuda.stream(self._side), torch.no_grad():
            batch = self.render(p, dev_params)
            extra = self.ahead_fn(batch) if self.ahead_fn is not None else None
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._pending = (batch, extra, ev)

    @staticmethod
    def _tensors(obj):
        if isinstance(obj, torch.Tensor):
            yield obj
        elif isinstance(obj, dict):
            for v in obj.values():
                yield from SyntheticFxBatcher._tensors(v)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                yield from SyntheticFxBatcher._tensors(v)

    def next_batch(self) -> Tuple[T, T, T, Dict[str, Any]]:
        if not self.overlap:
            with torch.no_grad():
                batch = self.render(self.sample_params())
                self.last_ahead = self.ahead_fn(batch) if self.ahead_fn is not None else None
                return batch
        if self._pending is None:
            self._launch_ahead()
        batch, extra, ev = self._pending
        self.last_ahead = extra
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ev)
        for t in (batch[2], *self._tensors(batch[3]), *self._tensors(extra)):
            if t.is_cuda:
                t.record_stream(main)           # allocated on the side stream, consumed on the main stream
        self._launch_ahead()                    # overlaps with the train step the caller is about to enqueue
        return batch


# ---- data modules with the reference's class names ---------------------------------------------
class _SyntheticDataModule:
    """Common part: per-rank batch size (``batch_size`` is per process, as under Lightning DDP),
    ``train_batch()`` / ``val_batch()`` producing the 4-tuple on the device."""

    kinds: Sequence[str] = ("flanger",)

    def __init__(self, batch_size: int, n_samples: int = 88200, sr: float = 44100,
                 train_num_examples_per_epoch: int = 8000, val_num_examples_per_epoch: int = 2000,
                 fx_config: Optional[Dict[str, Any]] = None, rng_order: str = "batch", overlap: bool = True,
                 **ignored: Any) -> None:
        self.batch_size, self.n_samples, self.sr = batch_size, n_samples, sr
        self.rng_order = rng_order
        self.overlap = overlap              # render (and prefetch work, see set_ahead_fn) one batch ahead on a side stream
        self._ahead_fn = None
        self.train_num_examples_per_epoch = train_num_examples_per_epoch
        self.val_num_examples_per_epoch = val_num_examples_per_epoch
        self.fx_config = fx_config or {}
        self.ignored_args = ignored          # num_workers, ... (dataset directories and chunk-search settings are used
        self._batcher: Optional[SyntheticFxBatcher] = None      # when the directories exist, see _chunk_source)
        self._val_batcher: Optional[SyntheticFxBatcher] = None

    _DS_KEYS = ("ext", "silence_fraction_allowed", "silence_threshold_energy", "n_retries", "check_dataset",
                "end_buffer_n_samples", "should_peak_norm", "peak_norm_db")

    def _chunk_source(self, which: str) -> Optional[Any]:
        """datasets.FileChunkSource over `<which>_dir` (train_dir / val_dir of the reference's init_args) if that
        directory exists, else None (synthetic clips)."""
        d = self.ignored_args.get(f"{which}_dir") or self.ignored_args.get(f"dry_{which}_dir")
        if not d or not os.path.isdir(d):
            return None
        from . import datasets
        kw = {k: self.ignored_args[k] for k in self._DS_KEYS if k in self.ignored_args}
        n_ex = self.train_num_examples_per_epoch if which == "train" else self.val_num_examples_per_epoch
        ds = datasets.RandomAudioChunkDataset(d, self.n_samples, self.sr, num_examples_per_epoch=n_ex, **kw)
        return datasets.FileChunkSource(ds)

    def setup(self, device: torch.device, rank: int = 0, seed: int = 43) -> None:
        fl = self.fx_config.get("flanger")
        common = dict(flanger_fx=fl, chorus_fx=fl if "chorus" in self.kinds and fl else None,
                      phaser_fx=self.fx_config.get("pedalboard_phaser"), mod_sig=self.fx_config.get("mod_sig"),
                      audio_seed=seed + rank, rng_order=self.rng_order)
        self._batcher = SyntheticFxBatcher(self.batch_size, self.n_samples, self.sr, self.kinds, device,
                                           chunk_source=self._chunk_source("train"), overlap=self.overlap, **common)
        self._batcher.ahead_fn = self._ahead_fn
        val_src = self._chunk_source("val")
        if val_src is not None:
            self._val_batcher = SyntheticFxBatcher(self.batch_size, self.n_samples, self.sr, self.kinds, device,
                                                   chunk_source=val_src, **common)

    def set_ahead_fn(self, fn) -> None:
        """``fn(batch)`` is evaluated on every TRAINING batch right after it is rendered -- one batch ahead on the side
        stream when ``overlap`` is on; ``take_ahead()`` returns its result for the batch handed out last."""
        self._ahead_fn = fn
        if self._batcher is not None:
            self._batcher.ahead_fn = fn

    def pause_ahead(self, paused: bool) -> None:
        """Validation hands out batches of the training batcher when there is no validation corpus: the prefetch hook
        must not run for those (``validation_step`` evaluates ``prepare`` itself)."""
        if self._batcher is not None:
            self._batcher.ahead_fn = None if paused else self._ahead_fn

    def use_side_stream(self, stream) -> None:
        for b in (self._batcher, self._val_batcher):
            if b is not None:
                b.use_side_stream(stream)

    def take_ahead(self):
        b = self._batcher
        if b is None:
            return None
        out, b.last_ahead = b.last_ahead, None
        return out

    def train_steps_per_epoch(self) -> int:
        return max(1, self.train_num_examples_per_epoch // self.batch_size)

    def val_steps_per_epoch(self) -> int:
        return max(1, self.val_num_examples_per_epoch // self.batch_size)

    def train_batch(self):
        return self._batcher.next_batch()

    def val_batch(self):
        return (self._val_batcher or self._batcher).next_batch()


class FlangerCPUDataModule(_SyntheticDataModule):
    """data_modules.py:374-458 -- the flanger is rendered on the device, not on the CPU."""
    kinds = ("flanger",)


class PedalboardPhaserDataModule(_SyntheticDataModule):
    """data_modules.py:259-328."""
    kinds = ("phaser",)


class RandomAudioChunkAndModSigDataModule(_SyntheticDataModule):
    """data_modules.py:331-371 (configs/eval_lfo_rand.yml): unprocessed chunks with random LFO labels, batches of
    ``(None, chunk, mod_sig, fx_params)`` -- the input of the random-LFO baseline (models.RandomLFO, use_dry false)."""
    kinds = ("dry",)

    def train_batch(self):
        _, wet, mod, params = super().train_batch()
        return None, wet, mod, params

    def val_batch(self):
        _, wet, mod, params = super().val_batch()
        return None, wet, mod, params


class InterwovenDataModule(_SyntheticDataModule):
    """data_modules.py:20-83 with configs/data/interwoven_idmt_all.yml: flanger, chorus, phaser
    interleaved by ``idx % 3`` (the reference pre-renders flanger/chorus to disk)."""
    kinds = ("flanger", "chorus", "phaser")

    def __init__(self, batch_size: int, num_workers: int = 0, shared_args: Optional[Dict[str, Any]] = None,
                 shared_train_args: Optional[Dict[str, Any]] = None, shared_val_args: Optional[Dict[str, Any]] = None,
                 train_dataset_args: Optional[List[Dict[str, Any]]] = None,
                 val_dataset_args: Optional[List[Dict[str, Any]]] = None, **ignored: Any) -> None:
        shared_args = shared_args or {}
        fx_config: Dict[str, Any] = {}
        for ds in (train_dataset_args or []):
            if "fx_config" in ds:
                fx_config.update(ds["fx_config"])
        super().__init__(batch_size, n_samples=shared_args.get("n_samples", 88200), sr=shared_args.get("sr", 44100),
                         train_num_examples_per_epoch=(shared_train_args or {}).get("num_examples_per_epoch", 8000),
                         val_num_examples_per_epoch=(shared_val_args or {}).get("num_examples_per_epoch", 2000),
                         fx_config=fx_config, num_workers=num_workers, **ignored)
        self.shared_args = shared_args


class RandomAudioChunkDryWetDataModule(_SyntheticDataModule):
    """data_modules.py:177-256 (dry/wet pairs of an unseen effect, configs/train_em_dry_wet.yml).  With
    ``dry_train_dir`` / ``wet_train_dir`` (and the ``_val_`` pair) present on disk the batches are recorded pairs
    (``datasets.RandomAudioChunkDryWetDataset``); otherwise synthetic dry clips and this package's phaser render."""
    kinds = ("phaser",)

    def _pair_dataset(self, which: str):
        dry_d, wet_d = self.ignored_args.get(f"dry_{which}_dir"), self.ignored_args.get(f"wet_{which}_dir")
        if not dry_d or not wet_d or not os.path.isdir(dry_d) or not os.path.isdir(wet_d):
            return None
        from . import datasets
        kw = {k: self.ignored_args[k] for k in self._DS_KEYS if k in self.ignored_args}
        n_ex = self.train_num_examples_per_epoch if which == "train" else self.val_num_examples_per_epoch
        return datasets.RandomAudioChunkDryWetDataset(dry_d, wet_d, self.n_samples, self.sr, num_examples_per_epoch=n_ex, **kw)

    def setup(self, device: torch.device, rank: int = 0, seed: int = 43) -> None:
        self._pairs = {w: self._pair_dataset(w) for w in ("train", "val")}
        self._device = device
        if self._pairs["train"] is None:
            super().setup(device, rank, seed)

    def set_ahead_fn(self, fn) -> None:
        # the step sees (dry, wet, None, None): prefetch on exactly that tuple
        super().set_ahead_fn(None if fn is None else (lambda b: fn((b[0], b[1], None, None))))

    def take_ahead(self):
        return super().take_ahead() if self._pairs.get("train") is None else None

    def _pair_batch(self, ds):
        dry = torch.empty((self.batch_size, 1, self.n_samples), dtype=torch.float32)
        wet = torch.empty_like(dry)
        for i in range(self.batch_size):
            dry[i], wet[i] = ds[i]
        return dry.to(self._device), wet.to(self._device), None, None

    def train_batch(self):
        if self._pairs["train"] is not None:
            return self._pair_batch(self._pairs["train"])
        dry, wet, _, _ = self._batcher.next_batch()
        return dry, wet, None, None

    def val_batch(self):
        ds = self._pairs["val"] or self._pairs["train"]
        if ds is not None:
            return self._pair_batch(ds)
        dry, wet, _, _ = (self._val_batcher or self._batcher).next_batch()
        return dry, wet, None, None
