"""CPU: the model classes accept every constructor argument of the reference (mod_extraction/models.py:127-145,311-316) and
keep its module tree / state-dict keys outside the shipped shapes too (the arithmetic of those shapes is covered on the GPU by
tests/test_gpu_generic_cnn.py and tests/test_gpu_generic_lstm.py)."""
import pytest
import torch

from oracle import models as om


def test_spectral2dcnn_class_defaults_build_and_mirror_the_reference_keys():
    from mod_extraction_amd import models as am
    mine, ref = am.Spectral2DCNN(), om.Spectral2DCNN()             # pool (3,1), five blocks, temp dilations 1..16, in_ch 1
    assert mine.generic and mine.pool_size == (3, 1) and mine.temp_dilations == [1, 2, 4, 8, 16]
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert mine.state_dict()[k].shape == v.shape, k
    mine.load_state_dict(ref.state_dict(), strict=True)
    no_ln = am.Spectral2DCNN(use_ln=False, out_channels=[4, 4], n_mels=18)
    assert [k for k in no_ln.state_dict() if k.startswith("cnn.")] == ["cnn.0.weight", "cnn.0.bias", "cnn.2.weight", "cnn.3.weight",
                                                                       "cnn.3.bias", "cnn.5.weight"]
    assert not am.Spectral2DCNN(in_ch=2, n_mels=256, out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1)).generic


def test_lstm_effect_model_sizes_and_keys():
    from mod_extraction_amd import models as am
    for args in [(1, 1, 64, 1), (1, 1, 32, 3), (2, 2, 48, 1), (1, 3, 16, 1)]:
        mine, ref = am.LSTMEffectModel(*args), om.LSTMEffectModel(*args)
        assert mine.generic == (args != (1, 1, 64, 1))
        assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    with pytest.raises(ValueError):
        am.LSTMEffectModel(2, 3, 8, 1)


def test_product_ops_refuse_host_tensors():
    """No CPU fallback anywhere on the general paths either: a host tensor is an error, not a silent torch evaluation."""
    from mod_extraction_amd import _hip, models as am
    net = am.Spectral2DCNN(n_samples=6000, n_mels=9, out_channels=[2], temp_dilations=[1])
    with pytest.raises((_hip.HipLibraryError, RuntimeError)):
        net(torch.zeros(1, 1, 6000))
    em = am.LSTMEffectModel(1, 1, 8, 2)
    with pytest.raises((_hip.HipLibraryError, RuntimeError)):
        em(torch.zeros(1, 1, 16), torch.zeros(1, 2, 16))
