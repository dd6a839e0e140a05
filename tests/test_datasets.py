"""CPU: the file-backed data path against the decisions of the REAL reference
`mod_extraction.datasets` (tests/golden/datasets_decisions.json, produced by tests/golden/make_golden_datasets.py
on the corpus of tests/golden/wav_fixture.py): surviving files, the random non-silent chunk search under a seed
(file, channel, offset, sample checksum), silence verdicts, dry/wet pairing, pre-rendered + interleaved datasets.
Index / path decisions are exact; checksums compare float64 sums of identical float32 samples (exact)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    from wav_fixture import make_corpus
    root = str(tmp_path_factory.mktemp("corpus"))
    return root, make_corpus(root)


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(HERE, "golden", "datasets_decisions.json")))


KW = dict(n_samples=4410, sr=44100, silence_fraction_allowed=0.1, silence_threshold_energy=1e-6, n_retries=4,
          check_dataset=True)


def test_wav_io_round_trip(tmp_path):
    from mod_extraction_amd import datasets as ds
    x = (torch.rand(2, 3000) * 2 - 1) * 0.7
    p = str(tmp_path / "x.wav")
    ds.wav_save(p, x, 44100)
    assert ds.wav_info(p) == (3000, 44100, 2)
    y, sr = ds.wav_load(p, frame_offset=100, num_frames=500)
    assert sr == 44100 and torch.equal(y, x[:, 100:600])
    from scipy.io import wavfile
    wavfile.write(str(tmp_path / "i.wav"), 22050, np.array([0, 16384, -32768, 32767], dtype=np.int16))
    z, sr = ds.wav_load(str(tmp_path / "i.wav"))
    assert sr == 22050 and z.tolist() == [[0.0, 0.5, -1.0, 32767 / 32768]]


def test_chunk_search_matches_reference(corpus, gold):
    from mod_extraction_amd import datasets as ds
    root, dirs = corpus
    rel = lambda p: os.path.relpath(p, root)
    torch.manual_seed(7)
    d = ds.RandomAudioChunkDataset(dirs["dry"], **KW)
    assert [rel(p) for p in d.input_paths] == gold["kept"]             # too short / wrong rate / dot-file dropped
    for want in gold["picks"]:
        chunk, path, ch, start = d.search_dataset_for_audio_chunk(4410, 0)
        assert [rel(path), ch, start, float(chunk.double().sum()), list(chunk.shape)] == want
    torch.manual_seed(8)
    for want in gold["picks_long"]:
        chunk, path, ch, start = d.search_dataset_for_audio_chunk(3 * 4410, 100)
        assert [rel(path), ch, start, float(chunk.double().sum())] == want
    g = torch.Generator().manual_seed(3)
    loud = torch.rand((1, 4410), generator=g) - 0.5
    hole = loud.clone()
    hole[:, 1000:1500] = 0.0
    assert [d.check_for_silence(loud), d.check_for_silence(hole), d.check_for_silence(loud * 1e-4)] == gold["silence"]
    assert len(d) == 10000 and d[0].shape == (1, 4410)


def test_dry_wet_pairs_match_reference(corpus, gold):
    from mod_extraction_amd import datasets as ds
    root, dirs = corpus
    rel = lambda p: os.path.relpath(p, root)
    torch.manual_seed(9)
    dw = ds.RandomAudioChunkDryWetDataset(dirs["dry"], dirs["wet"], **KW)
    assert [[rel(a), rel(b)] for a, b in zip(dw.dry_paths, dw.wet_paths)] == gold["pairs"]
    for i, (sd, sw) in enumerate(gold["pair_items"]):
        dry, wet = dw[i]
        assert dry.shape == wet.shape == (1, 4410)
        assert float(dry.double().sum()) == sd and float(wet.double().sum()) == sw


def test_preprocessed_and_interwoven_match_reference(corpus, gold):
    from mod_extraction_amd import datasets as ds
    _, dirs = corpus
    torch.manual_seed(10)
    inter = ds.InterwovenDataset([{"dataset_name": "preproc", "input_dir": dirs["pre_a"]},
                                  {"dataset_name": "random_preproc", "input_dir": dirs["pre_b"],
                                   "num_examples_per_epoch": 3, "n_copies": 2}], {"n_samples": 4410, "sr": 44100})
    assert len(inter) == gold["interwoven_len"]
    for i, (tag, sd, sm) in enumerate(gold["interwoven"]):
        dry, wet, mod, fxp = inter[i]
        assert fxp["tag"] == tag and float(dry.double().sum()) == sd and float(mod.double().sum()) == sm
    with pytest.raises(ValueError):
        ds.get_dataset_class("nope")


def test_peak_normalize_and_writer(tmp_path):
    from mod_extraction_amd import datasets as ds
    x = torch.tensor([[0.1, -0.25, 0.2]])
    y = ds.peak_normalize(x, -1.0)                                      # pyloudnorm.normalize.peak: max|y| = 10^(-1/20)
    assert abs(float(y.abs().max()) - 10 ** (-1 / 20)) < 1e-7 and torch.allclose(y / y[0, 1], x / x[0, 1])
    ds.write_preprocessed_item(str(tmp_path), "it", x.repeat(1, 1470), -x.repeat(1, 1470), torch.rand(44),
                               {"rate_hz": 1.0}, 44100)
    p = ds.PreprocessedDataset(str(tmp_path), 4410, 44100)
    dry, wet, mod, fxp = p[0]
    assert torch.equal(dry, x.repeat(1, 1470)) and torch.equal(wet, -dry) and mod.shape == (44,) and fxp["rate_hz"] == 1.0


def test_file_chunk_source_fills_requested_lengths(corpus):
    from mod_extraction_amd import datasets as ds
    _, dirs = corpus
    torch.manual_seed(11)
    d = ds.RandomAudioChunkDataset(dirs["dry"], **KW)
    src = ds.FileChunkSource(d)
    buf = torch.full((5, 4410 + 3000), 7.0)
    need = torch.tensor([4410, 4410 + 3000, 4410 + 17, 4410, 4410 + 1500])
    src.fill(buf, need)
    for i in range(5):
        n = int(need[i])
        assert float(buf[i, :n].abs().max()) > 0 and bool((buf[i, n:] == 0).all())
        assert not d.check_for_silence(buf[i:i + 1, :n])


def test_dry_wet_data_module_reads_recorded_pairs(corpus):
    """RandomAudioChunkDryWetDataModule with directories on disk: batches are the recorded pairs (host side only)."""
    from mod_extraction_amd import data_modules
    _, dirs = corpus
    torch.manual_seed(12)
    dm = data_modules.RandomAudioChunkDryWetDataModule(batch_size=3, n_samples=4410, sr=44100, dry_train_dir=dirs["dry"],
                                                       wet_train_dir=dirs["wet"], dry_val_dir=dirs["dry"], wet_val_dir=dirs["wet"],
                                                       silence_fraction_allowed=0.1, n_retries=4)
    dm.setup(torch.device("cpu"))
    dry, wet, mod, fxp = dm.train_batch()
    assert dry.shape == wet.shape == (3, 1, 4410) and mod is None and fxp is None
    assert torch.allclose(wet, torch.tanh(2.0 * dry), atol=1e-6)       # the corpus' wet files are tanh(2 dry)
    assert dm.val_batch()[0].shape == (3, 1, 4410)
