"""File-backed data path: host mirror of the dataset classes of
mod_extraction/datasets.py that read audio from disk.

What stays on the host, exactly as in the reference (same RNG helpers, same order of draws, same decisions):
  * file discovery and filtering by length / sample rate            (datasets.py:120-149, 204-216)
  * random non-silent chunk search with retries and file fall-back   (datasets.py:151-213)
  * dry / wet pairing by file name                                   (datasets.py:243-314)
  * peak normalisation                                               (datasets.py:215-220; pyloudnorm 0.1.1
    `normalize.peak`, third party and absent: gain = 10^(dB/20) / max|x| is its published definition)
  * the pre-rendered dataset format `<stem>.pt` + `<stem>_dry.wav` + `<stem>_wet.wav`  (datasets.py:504-551)
  * interleaving of several datasets by `idx % n`                    (datasets.py:42-83)
What moves to the device: the effects and LFO labels are NOT rendered per item here; `FileChunkSource` hands whole
batches of chunks to `data_modules.SyntheticFxBatcher`, which renders flanger / chorus / phaser and the LFOs in HBM
(`mx_flanger_fwd`, `mx_phaser_fwd`, `mx_lfo_synth`).

Audio I/O is scipy's RIFF reader (memory mapped; PCM 16/24/32 and float32), scaled like torchaudio.load
(int PCM / 2^(bits-1)).  torchaudio itself is not part of this image.
"""
import logging
import os
from typing import Any, Dict, List, Optional, Tuple, Type

import numpy as np
import torch
from torch import Tensor as T

from . import util

log = logging.getLogger(__name__)


# ---- wav I/O ------------------------------------------------------------------------------------------------
def _wav_open(path: str) -> Tuple[int, np.ndarray]:
    from scipy.io import wavfile
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                     # non-data chunks (LIST, bext, ...)
        sr, data = wavfile.read(path, mmap=True)
    if data.ndim == 1:
        data = data[:, None]
    return int(sr), data                                    # (frames, channels), native dtype


def wav_info(path: str) -> Tuple[int, int, int]:
    """(num_frames, sample_rate, num_channels) -- the three fields of torchaudio.info the reference reads."""
    sr, data = _wav_open(path)
    return int(data.shape[0]), sr, int(data.shape[1])


def wav_load(path: str, frame_offset: int = 0, num_frames: int = -1) -> Tuple[T, int]:
    """(channels, frames) float32 in [-1, 1) like torchaudio.load(normalize=True)."""
    sr, data = _wav_open(path)
    end = data.shape[0] if num_frames < 0 else min(data.shape[0], frame_offset + num_frames)
    seg = np.asarray(data[frame_offset:end])
    if seg.dtype == np.int16:
        out = seg.astype(np.float32) / 32768.0
    elif seg.dtype == np.int32:
        out = (seg.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif seg.dtype == np.uint8:
        out = (seg.astype(np.float32) - 128.0) / 128.0
    else:
        out = seg.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(out.T)), sr


def wav_save(path: str, audio: T, sr: int) -> None:
    """float32 RIFF (what the reference's generation scripts write through torchaudio.save)."""
    from scipy.io import wavfile
    a = audio.detach().cpu().float().numpy()
    wavfile.write(path, int(sr), np.ascontiguousarray(a.T))


def list_files(input_dir: str, ext: str) -> List[str]:
    """datasets.py:222-233: recursive, sorted, dot-files skipped."""
    assert os.path.isdir(input_dir), input_dir
    found = [os.path.join(root, name) for root, _, names in os.walk(input_dir) for name in names
             if name.endswith(ext) and not name.startswith(".")]
    found.sort()
    assert len(found) > 0, f"no *{ext} under {input_dir}"
    return found


def peak_normalize(audio: T, peak_db: float) -> T:
    """pyloudnorm.normalize.peak on a (channels, frames) tensor."""
    assert audio.ndim == 2
    current = float(audio.abs().max())
    gain = (10.0 ** (peak_db / 20.0)) / current
    return audio * gain


# ---- random non-silent chunks -------------------------------------------------------------------------------
class RandomAudioChunkDataset:
    """datasets.py:86-233.  Items are (1, n_samples) float32 host tensors."""

    get_file_paths = staticmethod(list_files)       # datasets.py:230-241, under the reference's name

    def __init__(self, input_dir: str, n_samples: int, sr: float, ext: str = "wav", num_examples_per_epoch: int = 10000,
                 silence_fraction_allowed: float = 0.2, silence_threshold_energy: float = 1e-6, n_retries: int = 10,
                 check_dataset: bool = True, min_suitable_files_fraction: float = 0.5, end_buffer_n_samples: int = 0,
                 should_peak_norm: bool = False, peak_norm_db: float = -1.0) -> None:
        self.input_dir, self.n_samples, self.sr, self.ext = input_dir, n_samples, sr, ext
        self.num_examples_per_epoch = num_examples_per_epoch
        self.silence_fraction_allowed = silence_fraction_allowed
        self.silence_threshold_energy = silence_threshold_energy
        self.n_retries, self.check_dataset = n_retries, check_dataset
        self.min_suitable_files_fraction = min_suitable_files_fraction
        self.end_buffer_n_samples = end_buffer_n_samples
        self.should_peak_norm, self.peak_norm_db = should_peak_norm, peak_norm_db
        self.max_n_consecutive_silent_samples = int(silence_fraction_allowed * n_samples)
        self._frames: Dict[str, int] = {}
        kept, total = [], 0
        for path in list_files(input_dir, ext):
            frames, file_sr, _ = wav_info(path)
            if frames < n_samples or file_sr != sr:          # too short / wrong rate: dropped (datasets.py:125-131)
                continue
            self._frames[path] = frames
            total += frames
            kept.append(path)
        log.info("%d usable files, %.0f s of audio", len(kept), total / sr)
        assert len(kept) > 0
        self.input_paths = kept
        if check_dataset:
            assert self.check_dataset_for_suitable_files(n_samples, min_suitable_files_fraction, end_buffer_n_samples), \
                "Could not find a suitable non-silent audio chunk in the dataset"

    def num_frames(self, path: str) -> int:
        if path not in self._frames:
            self._frames[path] = wav_info(path)[0]
        return self._frames[path]

    def check_dataset_for_suitable_files(self, n_samples: int, min_suitable_files_fraction: float,
                                         end_buffer_n_samples: int = 0) -> bool:
        need = max(1, int(min_suitable_files_fraction * len(self.input_paths)))
        ok = 0
        for path in self.input_paths:
            for _ in range(self.n_retries):
                if self.find_audio_chunk_in_file(path, n_samples, end_buffer_n_samples) is not None:
                    ok += 1
                    break
        return ok >= need

    def check_for_silence(self, audio_chunk: T) -> bool:
        """True if any window of `silence_fraction_allowed * n_samples` samples (hop = window / 4) has a mean
        energy below the threshold (datasets.py:165-172; same torch ops, so the same decisions)."""
        window = self.max_n_consecutive_silent_samples
        hop = window // 4
        means = (audio_chunk ** 2).unfold(dimension=-1, size=window, step=hop).mean(dim=-1)
        return int((means < self.silence_threshold_energy).sum()) > 0

    def find_audio_chunk_in_file(self, file_path: str, n_samples: int,
                                 end_buffer_n_samples: int = 0) -> Optional[Tuple[T, int]]:
        file_n = self.num_frames(file_path)
        if n_samples > file_n - end_buffer_n_samples:
            return None
        start = util.randint(0, file_n - n_samples - end_buffer_n_samples + 1)
        chunk, _ = wav_load(file_path, frame_offset=start, num_frames=n_samples)
        if self.check_for_silence(chunk):
            return None
        return chunk, start

    def search_dataset_for_audio_chunk(self, n_samples: int, end_buffer_n_samples: int = 0) -> Tuple[T, str, int, int]:
        """One random file, up to n_retries random offsets in it, then the next random file of the remaining pool
        (datasets.py:189-213).  Returns (chunk (1, n), path, channel, start)."""
        pool = list(self.input_paths)
        path = util.choice(pool)
        pool.remove(path)
        found, attempts = None, 0
        while found is None:
            found = self.find_audio_chunk_in_file(path, n_samples, end_buffer_n_samples)
            if found is None:
                attempts += 1
            if attempts >= self.n_retries:
                assert pool, "no file of the dataset holds a non-silent chunk"
                path = util.choice(pool)
                pool.remove(path)
                attempts = 0
        chunk, start = found
        ch = 0
        if chunk.size(0) > 1:
            ch = util.randint(0, chunk.size(0))
            chunk = chunk[ch, :].view(1, -1)
        return chunk, path, ch, start

    def peak_normalize(self, audio: T) -> T:
        return peak_normalize(audio, self.peak_norm_db)

    def __len__(self) -> int:
        return self.num_examples_per_epoch

    def __getitem__(self, _: int) -> T:
        chunk, _, _, _ = self.search_dataset_for_audio_chunk(self.n_samples, self.end_buffer_n_samples)
        return self.peak_normalize(chunk) if self.should_peak_norm else chunk


class RandomAudioChunkDryWetDataset(RandomAudioChunkDataset):
    """datasets.py:236-329: chunks of recorded dry / wet pairs (same file name in two directories)."""

    def __init__(self, dry_dir: str, wet_dir: str, n_samples: int, sr: float, ext: str = "wav",
                 num_examples_per_epoch: int = 10000, silence_fraction_allowed: float = 0.1,
                 silence_threshold_energy: float = 1e-6, n_retries: int = 10, check_dataset: bool = True,
                 min_suitable_files_fraction: float = 0.5, end_buffer_n_samples: int = 0, should_peak_norm: bool = False,
                 peak_norm_db: float = -1.0) -> None:
        super().__init__(dry_dir, n_samples, sr, ext, num_examples_per_epoch, silence_fraction_allowed,
                         silence_threshold_energy, n_retries, check_dataset, min_suitable_files_fraction,
                         end_buffer_n_samples, should_peak_norm, peak_norm_db)
        self.dry_dir, self.wet_dir = dry_dir, wet_dir
        wet_by_name = {os.path.basename(p): p for p in list_files(wet_dir, ext)}
        pairs = []
        for dry_p in self.input_paths:
            name = os.path.basename(dry_p)
            assert name in wet_by_name, f"Missing wet file: {name}"
            wet_p = wet_by_name[name]
            d_frames, d_sr, d_ch = wav_info(dry_p)
            w_frames, w_sr, w_ch = wav_info(wet_p)
            if d_sr != w_sr or abs(d_frames - w_frames) > end_buffer_n_samples or d_ch != w_ch:
                continue                                        # datasets.py:291-299
            pairs.append((dry_p, wet_p))
        pairs.sort()
        assert len(pairs) > 0
        self.dry_paths = [p for p, _ in pairs]
        self.wet_paths = sorted(w for _, w in pairs)
        self.input_paths = self.dry_paths
        self.name_to_wet_path = {os.path.basename(d): w for d, w in pairs}

    def __getitem__(self, _: int) -> Tuple[T, T]:
        dry, dry_path, ch, start = self.search_dataset_for_audio_chunk(self.n_samples, self.end_buffer_n_samples)
        wet, _ = wav_load(self.name_to_wet_path[os.path.basename(dry_path)], frame_offset=start, num_frames=self.n_samples)
        if wet.size(0) > 1:
            wet = wet[ch, :].view(1, -1)
        assert dry.shape == wet.shape
        if self.should_peak_norm:
            dry, wet = self.peak_normalize(dry), self.peak_normalize(wet)
        return dry, wet


# ---- pre-rendered datasets ----------------------------------------------------------------------------------
class PreprocessedDataset:
    """datasets.py:504-535: `<stem>.pt` = {"mod_sig", "fx_params"} next to `<stem>_dry.wav` / `<stem>_wet.wav`."""

    def __init__(self, input_dir: str, n_samples: int, sr: float) -> None:
        self.input_dir, self.n_samples, self.sr = input_dir, n_samples, sr
        self.pt_paths = list_files(input_dir, ".pt")
        self.dry_paths = [f"{p[:-3]}_dry.wav" for p in self.pt_paths]
        self.wet_paths = [f"{p[:-3]}_wet.wav" for p in self.pt_paths]

    def __len__(self) -> int:
        return len(self.pt_paths)

    def __getitem__(self, idx: int) -> Tuple[T, T, T, Dict[str, Any]]:
        data = torch.load(self.pt_paths[idx])
        dry, sr_d = wav_load(self.dry_paths[idx])
        wet, sr_w = wav_load(self.wet_paths[idx])
        assert sr_d == self.sr and sr_w == self.sr
        assert dry.size(-1) == self.n_samples and wet.size(-1) == self.n_samples
        return dry, wet, data["mod_sig"], data["fx_params"]


class RandomPreprocessedDataset(PreprocessedDataset):
    """datasets.py:538-551."""

    def __init__(self, num_examples_per_epoch: int, input_dir: str, n_samples: int, sr: float) -> None:
        super().__init__(input_dir, n_samples, sr)
        self.num_examples_per_epoch = num_examples_per_epoch

    def __len__(self) -> int:
        return self.num_examples_per_epoch

    def __getitem__(self, idx: int) -> Tuple[T, T, T, Dict[str, Any]]:
        return super().__getitem__(util.randint(0, len(self.pt_paths)))


def write_preprocessed_item(out_dir: str, stem: str, dry: T, wet: T, mod_sig: T, fx_params: Dict[str, Any], sr: int) -> None:
    """Writer of the format PreprocessedDataset reads (the reference generates it in scripts/scratch.py:145-162)."""
    os.makedirs(out_dir, exist_ok=True)
    torch.save({"mod_sig": mod_sig.detach().cpu(), "fx_params": fx_params}, os.path.join(out_dir, f"{stem}.pt"))
    wav_save(os.path.join(out_dir, f"{stem}_dry.wav"), dry, sr)
    wav_save(os.path.join(out_dir, f"{stem}_wet.wav"), wet, sr)


# ---- interleaving -------------------------------------------------------------------------------------------
def get_dataset_class(name: str) -> Type:
    table = {"random_audio_chunk": RandomAudioChunkDataset, "random_audio_chunk_dry_wet": RandomAudioChunkDryWetDataset,
             "preproc": PreprocessedDataset, "random_preproc": RandomPreprocessedDataset}
    if name not in table:
        raise ValueError(f"Unknown dataset name: {name}")       # on-device effect datasets: data_modules.FileFxDataModule
    return table[name]


class InterwovenDataset:
    """datasets.py:42-83: item `idx` comes from dataset `idx % n` (n_copies weights a dataset)."""

    def __init__(self, dataset_args: List[Dict[str, Any]], common_args: Dict[str, Any]) -> None:
        self.dataset_names, self.dataset_weightings, self.datasets = [], [], []
        for ds_args in dataset_args:
            ds_args = dict(ds_args)
            assert "dataset_name" in ds_args
            name = ds_args.pop("dataset_name")
            n_copies = ds_args.pop("n_copies", 1)
            for k, v in common_args.items():
                ds_args.setdefault(k, v)
            self.dataset_names.append(name)
            self.dataset_weightings.append(n_copies)
            for _ in range(n_copies):
                self.datasets.append(get_dataset_class(name)(**ds_args))
        self.size = len(self.datasets[0])
        assert all(len(d) == self.size for d in self.datasets)

    def __len__(self) -> int:
        return self.size

    def __getitem__(self, idx: int) -> Any:
        return self.datasets[idx % len(self.datasets)][idx]


# ---- batches of chunks for the on-device effect renderer ------------------------------------------------------
class FileChunkSource:
    """Feeds `SyntheticFxBatcher` with recorded audio instead of noise: one non-silent chunk per clip, found with the
    reference's search, `n_samples + lead_i` long for phaser clips (the reference searches for `n_samples +
    sr / rate` samples, renders all of them and crops at a random offset, datasets.py:428-449; the filter is causal,
    so the first `offset + n_samples` samples are all that the crop depends on)."""

    def __init__(self, dataset: RandomAudioChunkDataset) -> None:
        self.ds = dataset

    def fill(self, dst: T, need: T) -> None:
        """dst: (B, n_samples + max_lead) pinned/host fp32, zero padded; need (B,) int: samples wanted per clip."""
        dst.zero_()
        for i in range(dst.size(0)):
            n_i = int(need[i])
            chunk, _, _, _ = self.ds.search_dataset_for_audio_chunk(n_i, self.ds.end_buffer_n_samples)
            if self.ds.should_peak_norm:
                chunk = self.ds.peak_normalize(chunk)
            dst[i, :n_i] = chunk[0]
