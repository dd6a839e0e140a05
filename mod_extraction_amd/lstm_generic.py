"""``LSTMEffectModel`` (mod_extraction/models.py:311-339) for any ``in_ch`` / ``out_ch`` / ``n_hidden`` / ``latent_dim`` as ONE
autograd node per call: the two recurrences are ``csrc/lstm_generic.hip``, everything else -- the input projection, the output
layer, every parameter gradient and d loss / d latent -- is the general fp32 matrix-core GEMM (``mx_sgemm_f32``) over all
steps of the chunk at once.  The shipped LSTM-64 (1 audio + 1 LFO channel) keeps its fused kernels (``models.LSTMEffectModel``
dispatches); this path serves other sizes and the ``param_model`` variant of the TBPTT step (lightning.py:344-347,371-375).

The incoming state (h0, c0) is a constant of the node: the reference detaches it between TBPTT chunks
(lightning.py:353,383), and no gradient flows into a previous call.  x (the audio) receives no gradient.
"""
from typing import Tuple

import torch
from torch import Tensor as T

from . import _hip
from .tcn import _sgemm


def _empty(shape, dev) -> T:
    return torch.empty(shape, device=dev, dtype=torch.float32)


def _reduce_rows(part: T, R: int, C: int) -> T:
    out = _empty((C,), part.device)
    _hip.call("mx_reduce_rows", _hip.ptr(part), R, C, 0, _hip.ptr(out), _hip.stream())
    return out


class GenericLSTM(torch.autograd.Function):
    """(x (B, in_ch, T), latent (B, latent_dim, T), h0, c0 (B, Hn), weight_ih, weight_hh, bias_ih, bias_hh, fc.weight, fc.bias)
    -> (y (B, max(out_ch, in_ch), T), h1, c1)."""

    @staticmethod
    def forward(ctx, x: T, latent: T, h0: T, c0: T, w_ih: T, w_hh: T, b_ih: T, b_hh: T, w_fc: T, b_fc: T):
        dev, st = x.device, _hip.stream()
        B, in_ch, Tn = x.shape
        lat_dim, Hn, out_ch = latent.size(1), w_hh.size(1), w_fc.size(0)
        D, G = lat_dim + in_ch, 4 * Hn
        assert w_ih.shape == (G, D) and w_hh.shape == (G, Hn) and w_fc.shape == (out_ch, Hn)
        assert out_ch == in_ch or out_ch == 1 or in_ch == 1, "fc output and x do not broadcast (models.py:338)"
        x = x.contiguous().float()
        u = torch.cat([latent.float(), x], dim=1).transpose(1, 2).contiguous()           # (B, T, D): LFO first, audio second (models.py:331)
        w = [p.detach().contiguous().float() for p in (w_ih, w_hh, b_ih, b_hh, w_fc, b_fc)]
        h0, c0 = h0.detach().contiguous().float(), c0.detach().contiguous().float()
        zin = _empty((B, Tn, G), dev)
        _sgemm(_hip.ptr(u), D, 1, Tn * D, _hip.ptr(w[0]), 1, D, 0, _hip.ptr(zin), G, 1, Tn * G, Tn, G, D, B)
        stash, h1, c1 = _empty((B, Tn, 6, Hn), dev), _empty((B, Hn), dev), _empty((B, Hn), dev)
        _hip.call("mx_lstmg_fwd", _hip.ptr(zin), _hip.ptr(w[2]), _hip.ptr(w[3]), _hip.ptr(w[1]), _hip.ptr(h0), _hip.ptr(c0), B, Tn, Hn,
                  _hip.ptr(stash), _hip.ptr(h1), _hip.ptr(c1), st)
        del zin
        fc = _empty((B, Tn, out_ch), dev)
        hs = stash.data_ptr() + 4 * 5 * Hn                                                # h_t of every step: (B, T, Hn) at step stride 6 Hn
        _sgemm(hs, 6 * Hn, 1, Tn * 6 * Hn, _hip.ptr(w[4]), 1, Hn, 0, _hip.ptr(fc), out_ch, 1, Tn * out_ch, Tn, out_ch, Hn, B)
        Co = max(out_ch, in_ch)
        y = _empty((B, Co, Tn), dev)
        _hip.call("mx_lstmg_out_fwd", _hip.ptr(fc), _hip.ptr(w[5]), _hip.ptr(x), B, Tn, out_ch, in_ch, _hip.ptr(y), st)
        ctx.dims = (B, Tn, D, Hn, out_ch, Co, lat_dim)
        ctx.save_for_backward(u, stash, y, h0, c0, w[0], w[1], w[4])
        ctx.mark_non_differentiable(h1, c1)
        return y, h1, c1

    @staticmethod
    def backward(ctx, dy: T, _dh1, _dc1):
        u, stash, y, h0, c0, w_ih, w_hh, w_fc = ctx.saved_tensors
        B, Tn, D, Hn, out_ch, Co, lat_dim = ctx.dims
        G = 4 * Hn
        dev, st = y.device, _hip.stream()
        dy = dy.contiguous().float()
        dpre = _empty((B, Tn, out_ch), dev)
        _hip.call("mx_lstmg_out_bwd", _hip.ptr(dy), _hip.ptr(y), B, Tn, out_ch, Co, _hip.ptr(dpre), st)
        hs, s6 = stash.data_ptr() + 4 * 5 * Hn, 6 * Hn
        # output layer: dW_fc[o][j] = sum dpre[b][t][o] h[b][t][j] (one partial per clip), db_fc, and d loss / d h_t
        part = _empty((B, out_ch, Hn), dev)
        _sgemm(_hip.ptr(dpre), 1, out_ch, Tn * out_ch, hs, s6, 1, Tn * s6, _hip.ptr(part), Hn, 1, out_ch * Hn, out_ch, Hn, Tn, B)
        d_wfc = _reduce_rows(part, B, out_ch * Hn).view(out_ch, Hn)
        d_bfc = _reduce_rows(dpre, B * Tn, out_ch)
        dhfc = _empty((B, Tn, Hn), dev)
        _sgemm(_hip.ptr(dpre), out_ch, 1, Tn * out_ch, _hip.ptr(w_fc), Hn, 1, 0, _hip.ptr(dhfc), Hn, 1, Tn * Hn, Tn, Hn, out_ch, B)
        dgate = _empty((B, Tn, G), dev)
        _hip.call("mx_lstmg_bwd", _hip.ptr(stash), _hip.ptr(dhfc), _hip.ptr(w_hh), _hip.ptr(c0), B, Tn, Hn, _hip.ptr(dgate), st)
        # dW_hh[g][j] = sum over steps of dgate[b][t][g] h[b][t-1][j]: steps 1.. against the stash shifted by one, step 0 against h0
        part = _empty((B, G, Hn), dev)
        _sgemm(_hip.ptr(dgate), 1, Tn * G, 0, _hip.ptr(h0), Hn, 1, 0, _hip.ptr(part), Hn, 1, 0, G, Hn, B, 1)       # (K = clips)
        d_whh0 = part[0].clone()
        if Tn > 1:
            _sgemm(dgate.data_ptr() + 4 * G, 1, G, Tn * G, hs, s6, 1, Tn * s6, _hip.ptr(part), Hn, 1, G * Hn, G, Hn, Tn - 1, B)
            d_whh = _reduce_rows(part, B, G * Hn).view(G, Hn)
            _hip.call("mx_reduce_rows", _hip.ptr(d_whh0), 1, G * Hn, 1, _hip.ptr(d_whh), st)
        else:
            d_whh = d_whh0
        part = _empty((B, G, D), dev)
        _sgemm(_hip.ptr(dgate), 1, G, Tn * G, _hip.ptr(u), D, 1, Tn * D, _hip.ptr(part), D, 1, G * D, G, D, Tn, B)
        d_wih = _reduce_rows(part, B, G * D).view(G, D)
        d_b = _reduce_rows(dgate, B * Tn, G)
        d_lat = None
        if ctx.needs_input_grad[1]:
            du = _empty((B, Tn, D), dev)
            _sgemm(_hip.ptr(dgate), G, 1, Tn * G, _hip.ptr(w_ih), D, 1, 0, _hip.ptr(du), D, 1, Tn * D, Tn, D, G, B)
            d_lat = du[:, :, :lat_dim].transpose(1, 2).contiguous()
        return None, d_lat, None, None, d_wih, d_whh, d_b, d_b.clone(), d_wfc, d_bfc


def run(x: T, latent: T, h0: T, c0: T, params) -> Tuple[T, T, T]:
    return GenericLSTM.apply(x, latent, h0, c0, *params)
