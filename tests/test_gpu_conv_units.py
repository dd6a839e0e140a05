"""GPU parity, kernel by kernel: the fused conv block forward, its data gradient and its weight
gradient against torch's CPU Conv2d / LayerNorm / MaxPool2d / PReLU autograd (fp32), for every
temporal dilation the model family uses and for full (345) and short (88) frame counts."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
PITCH = 352


def to_planes(x, dev):
    """(B,C,H,W) cpu -> (B,C,H,352) device, zero padded."""
    out = torch.zeros(x.shape[:-1] + (PITCH,), dtype=x.dtype)
    out[..., :x.size(-1)] = x
    return out.to(dev).contiguous()


def ref_block(x_in, slope_prev, w, b, T, first):
    """torch CPU reference of one block up to the pooled pre-activation; returns (p, z, xhat)."""
    x = x_in if first else F.prelu(x_in, slope_prev)
    xhat = F.layer_norm(x, x.shape[-2:], eps=1e-5)
    z = F.conv2d(xhat, w, b, dilation=(1, T), padding="same")
    return F.max_pool2d(z, (2, 1)), z, xhat


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("T", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("W,H,cin", [(345, 8, 64), (88, 16, 64), (345, 12, 2), (88, 6, 2)])
def test_block_fwd_dgrad_wgrad(dev, T, W, H, cin):
    from mod_extraction_amd import _hip, models as am
    torch.manual_seed(100 * T + W + H)
    B = 2
    first = cin == 2
    x_in = torch.randn(B, cin, H, W) * (1.0 if first else 0.7) + 0.1
    slope_prev = None if first else torch.rand(cin) * 0.4 + 0.05
    w = (torch.randn(64, cin, 5, 13) / np.sqrt(cin * 65)).requires_grad_(True)
    b = (torch.randn(64) * 0.1).requires_grad_(True)
    x_req = x_in.clone().requires_grad_(True)
    p_r, z_r, xhat_r = ref_block(x_req, slope_prev, w, b, T, first)
    xhat_r.retain_grad()
    Gc = torch.randn_like(p_r)
    (p_r * Gc).sum().backward()

    st = _hip.stream()
    x_d = to_planes(x_in, dev)
    sl_d = slope_prev.to(dev) if slope_prev is not None else None
    stats = torch.empty((B, cin, 2), device=dev)
    _hip.call("mx_plane_stats", _hip.ptr(x_d), _hip.ptr(sl_d), B, cin, H, W, 1e-5, _hip.ptr(stats), st)
    wt = am._pack(w.detach().to(dev), 0)
    p = torch.empty((B, 64, H // 2, PITCH), device=dev)
    amax = torch.empty((B, 64, H // 2, PITCH), device=dev, dtype=torch.uint8)
    _hip.call("mx_conv_block_fwd", _hip.ptr(x_d), _hip.ptr(stats), _hip.ptr(sl_d), _hip.ptr(wt),
              _hip.ptr(b.detach().to(dev)), B, cin, H, W, T, 1 if first else 0, _hip.ptr(p), _hip.ptr(amax), st)
    assert rel(p.cpu()[..., :W], p_r.detach()) < 5e-6
    am_r = (z_r[:, :, 1::2] > z_r[:, :, 0::2]).to(torch.uint8)
    mism = amax.cpu()[..., :W] != am_r
    # a mismatch is only acceptable where the two pooled rows are equal to fp32 rounding
    if mism.any():
        gap = (z_r[:, :, 1::2] - z_r[:, :, 0::2]).abs()[mism]
        assert float(gap.max()) < 1e-5

    # use the oracle's argmax for the gradient checks so that routing is identical on both sides
    amax_d = to_planes(am_r, dev)
    G_d = to_planes(Gc, dev)
    rows = B * H
    rps = 3
    n_slabs = -(-rows // rps)
    part = torch.empty(n_slabs * 65 * 64 * cin, device=dev)
    dW = torch.empty((64, cin, 5, 13), device=dev)
    _hip.call("mx_conv_block_wgrad", _hip.ptr(G_d), _hip.ptr(amax_d), _hip.ptr(x_d), _hip.ptr(stats), _hip.ptr(sl_d),
              B, cin, H, W, T, rps, _hip.ptr(part), _hip.ptr(dW), st)
    assert rel(dW.cpu(), w.grad) < 1e-5, ("wgrad", rel(dW.cpu(), w.grad))
    bsum = torch.empty(B * 64, device=dev)
    _hip.call("mx_plane_sum", _hip.ptr(G_d), B * 64, H // 2, W, _hip.ptr(bsum), st)
    assert rel(bsum.view(B, 64).sum(0).cpu(), b.grad) < 1e-5
    if not first:
        wt_f = am._pack(w.detach().to(dev), 1)
        dxhat = torch.empty((B, 64, H, PITCH), device=dev)
        _hip.call("mx_conv_block_dgrad", _hip.ptr(G_d), _hip.ptr(amax_d), _hip.ptr(wt_f), B, H, W, T, _hip.ptr(dxhat), st)
        assert rel(dxhat.cpu()[..., :W], xhat_r.grad) < 1e-5, ("dgrad", rel(dxhat.cpu()[..., :W], xhat_r.grad))
        assert bool((dxhat[..., W:] == 0).all())
        ds_part = torch.empty(B * 64, device=dev)
        gsum = torch.empty(B * 64, device=dev)
        gmax = torch.zeros(1, device=dev, dtype=torch.int32)
        _hip.call("mx_ln_prelu_bwd", _hip.ptr(x_d), _hip.ptr(dxhat), _hip.ptr(stats), _hip.ptr(sl_d), B, 64, H, W,
                  _hip.ptr(ds_part), _hip.ptr(gsum), _hip.ptr(gmax), None, st)
        assert rel(dxhat.cpu()[..., :W], x_req.grad) < 1e-5, ("ln_prelu_bwd", rel(dxhat.cpu()[..., :W], x_req.grad))
        # by-products of the same pass: per-plane sums and the bit pattern of max|G|
        g_cpu = dxhat.cpu()[..., :W]
        assert rel(gsum.view(B, 64).cpu(), g_cpu.sum(dim=(2, 3))) < 1e-5
        assert float(gmax.view(torch.float32).cpu()) == float(g_cpu.abs().max())


@pytest.mark.parametrize("T", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("W,H", [(345, 8), (88, 16), (17, 2), (351, 4)])      # (17, 2): one row pair, every halo row outside the image
def test_block_f16x3_kernels(dev, T, W, H):
    """The split-fp16 family, kernel by kernel, against torch CPU autograd at the fp32 tolerance (1e-5 of the tensor's
    max): operand preparation, forward (LDS-DMA kernel for T <= 4, register-staged for T >= 8), data gradient, and BOTH
    weight-gradient kernels -- dense MFMA on the routed gradient, sparse MFMA on the pooled gradient + argmax index
    words -- which must also agree with each other."""
    from mod_extraction_amd import _hip, models as am
    torch.manual_seed(7 * T + W + H)
    B = 3
    x_in = torch.randn(B, 64, H, W) * 0.7 + 0.1
    slope_prev = torch.rand(64) * 0.4 + 0.05
    w = (torch.randn(64, 64, 5, 13) / np.sqrt(64 * 65)).requires_grad_(True)
    b = (torch.randn(64) * 0.1).requires_grad_(True)
    x_req = x_in.clone().requires_grad_(True)
    p_r, z_r, xhat_r = ref_block(x_req, slope_prev, w, b, T, False)
    xhat_r.retain_grad()
    Gc = torch.randn_like(p_r)
    (p_r * Gc).sum().backward()

    st = _hip.stream()
    x_d, sl_d = to_planes(x_in, dev), slope_prev.to(dev)
    stats = torch.empty((B, 64, 2), device=dev)
    _hip.call("mx_plane_stats", _hip.ptr(x_d), _hip.ptr(sl_d), B, 64, H, W, 1e-5, _hip.ptr(stats), st)
    x_hi = torch.empty((B, H, 4, PITCH, 16), device=dev, dtype=torch.float16)
    x_lo = torch.empty_like(x_hi)
    _hip.call("mx_conv_prep_fwd_f16", _hip.ptr(x_d), _hip.ptr(stats), _hip.ptr(sl_d), B, H, W, _hip.ptr(x_hi), _hip.ptr(x_lo), st)
    # the pair carries the normalised input to fp32 accuracy, channel-block major
    xhat_pair = (x_hi.float() + x_lo.float()).permute(0, 2, 4, 1, 3).reshape(B, 64, H, PITCH)
    assert rel(xhat_pair.cpu()[..., :W], xhat_r.detach()) < 2e-6
    w_hi, w_lo = am._pack_f16(w.detach().to(dev), 0)
    p = torch.empty((B, 64, H // 2, PITCH), device=dev)
    amax = torch.empty((B, 64, H // 2, PITCH), device=dev, dtype=torch.uint8)
    # with the slope of the PReLU that follows, the epilogue also leaves the next block's LayerNorm statistics as per-row sums
    sl_out = (torch.rand(64) * 0.4 + 0.05).to(dev)
    st_part = torch.empty((B, H // 2, 64, 2), device=dev)
    _hip.call("mx_conv_block_fwd_f16", _hip.ptr(x_hi), _hip.ptr(x_lo), _hip.ptr(w_hi), _hip.ptr(w_lo),
              _hip.ptr(b.detach().to(dev)), B, H, W, T, _hip.ptr(p), _hip.ptr(amax), _hip.ptr(sl_out), _hip.ptr(st_part), st)
    assert rel(p.cpu()[..., :W], p_r.detach()) < 1e-5
    stats_swept, stats_fused = torch.empty((B, 64, 2), device=dev), torch.empty((B, 64, 2), device=dev)
    _hip.call("mx_plane_stats", _hip.ptr(p), _hip.ptr(sl_out), B, 64, H // 2, W, 1e-5, _hip.ptr(stats_swept), st)
    _hip.call("mx_plane_stats_finish", _hip.ptr(st_part), _hip.ptr(b.detach().to(dev)), _hip.ptr(sl_out), B, 64, H // 2, W, 1e-5,
              _hip.ptr(stats_fused), st)
    y_r = torch.where(p_r.detach() > 0, p_r.detach(), sl_out.cpu().view(1, 64, 1, 1) * p_r.detach())
    mean_r, var_r = y_r.double().mean(dim=(2, 3)), y_r.double().var(dim=(2, 3), unbiased=False)
    # mean against the plane's standard deviation, rstd relatively: both at fp32 rounding level
    assert float(((stats_fused[..., 0].cpu().double() - mean_r).abs() / var_r.sqrt()).max()) < 2e-6
    assert float((stats_fused[..., 1].cpu().double() * (var_r + 1e-5).sqrt() - 1).abs().max()) < 2e-6
    assert float((stats_fused - stats_swept).abs().max() / stats_swept.abs().max()) < 2e-6
    p2 = torch.empty_like(p)
    _hip.call("mx_conv_block_fwd_f16", _hip.ptr(x_hi), _hip.ptr(x_lo), _hip.ptr(w_hi), _hip.ptr(w_lo),
              _hip.ptr(b.detach().to(dev)), B, H, W, T, _hip.ptr(p2), _hip.ptr(amax), None, None, st)
    assert torch.equal(p2, p)                                                     # the optional outputs change nothing else
    am_r = (z_r[:, :, 1::2] > z_r[:, :, 0::2]).to(torch.uint8)
    assert float((amax.cpu()[..., :W] != am_r).float().mean()) < 1e-4          # ties aside, the same argmax
    # gradient operand from the REFERENCE's argmax, so that all gradients below are comparable element by element
    G_d, amax_d = to_planes(Gc, dev), to_planes(am_r, dev)
    dz_hi = torch.empty((B, H, 4, PITCH, 16), device=dev, dtype=torch.float16)
    dz_lo = torch.empty_like(dz_hi)
    ws = torch.empty(1, device=dev, dtype=torch.int32)
    scale = torch.empty(2, device=dev)
    Hp = H // 2
    _hip.call("mx_conv_prep_dgrad_f16", _hip.ptr(G_d), _hip.ptr(amax_d), B, H, W, _hip.ptr(ws), 0, _hip.ptr(scale),
              _hip.ptr(dz_hi), _hip.ptr(dz_lo), st)
    S = float(scale[0])
    assert S == 2.0 ** round(np.log2(S)) and 512.0 <= float(Gc.abs().max()) * S < 1024.0       # power of two, in range
    # dense weight gradient
    rps = 3
    n_slabs = -(-(B * H) // rps)
    part = torch.empty(n_slabs * 65 * 64 * 64, device=dev)
    dW_dense = torch.empty((64, 64, 5, 13), device=dev)
    _hip.call("mx_conv_block_wgrad_f16", _hip.ptr(dz_hi), _hip.ptr(dz_lo), _hip.ptr(x_hi), _hip.ptr(x_lo), _hip.ptr(scale),
              B, H, T, rps, _hip.ptr(part), _hip.ptr(dW_dense), st)
    assert rel(dW_dense.cpu(), w.grad) < 1e-5, ("dense wgrad", rel(dW_dense.cpu(), w.grad))
    # the POOLED operand of both sparse kernels: channels-last pair of G * S, index words of the data gradient, planar
    # index words of the weight gradient -- one pass over G
    gc_hi = torch.empty((B, Hp, 4, PITCH, 16), device=dev, dtype=torch.float16)
    gc_lo = torch.empty_like(gc_hi)
    gc_idx = torch.empty((B, Hp, 4, PITCH), device=dev, dtype=torch.int32)
    gidx = torch.empty((B, 64, Hp, 22, 2), device=dev, dtype=torch.int16)
    _hip.call("mx_conv_prep_gpool_cl_f16", _hip.ptr(G_d), _hip.ptr(amax_d), _hip.ptr(scale), B, H, W, _hip.ptr(gc_hi),
              _hip.ptr(gc_lo), _hip.ptr(gc_idx), _hip.ptr(gidx), st)
    gc = (gc_hi.float() + gc_lo.float()).permute(0, 2, 4, 1, 3).reshape(B, 64, Hp, PITCH)
    assert rel(gc.cpu()[..., :W] / S, Gc) < 2e-6 and bool((gc[..., W:] == 0).all())
    # planar index words: per (k-step of 16 positions, lane half hh) the 8 two-bit fields of the lane's compressed
    # elements; element j = 2 gq + e sits at position 16 ks + 8 (gq >> 1) + 4 hh + 2 (gq & 1) + e, field = 2 e + argmax
    am_np = amax_d.cpu().numpy().astype(np.int64) & 1
    want_idx = np.zeros((B, 64, Hp, 22, 2), dtype=np.int64)
    for hh_ in range(2):
        for j_ in range(8):
            gq_, e_ = j_ >> 1, j_ & 1
            pos_ = 16 * np.arange(22) + 8 * (gq_ >> 1) + 4 * hh_ + 2 * (gq_ & 1) + e_
            want_idx[..., hh_] |= (2 * e_ + am_np[..., pos_]) << (2 * j_)
    assert np.array_equal(gidx.cpu().numpy().astype(np.int64) & 0xFFFF, want_idx)
    # same call without the weight-gradient words
    gd_hi, gd_lo, gd_idx = torch.empty_like(gc_hi), torch.empty_like(gc_lo), torch.empty_like(gc_idx)
    _hip.call("mx_conv_prep_gpool_cl_f16", _hip.ptr(G_d), _hip.ptr(amax_d), _hip.ptr(scale), B, H, W, _hip.ptr(gd_hi),
              _hip.ptr(gd_lo), _hip.ptr(gd_idx), None, st)
    assert torch.equal(gd_hi, gc_hi) and torch.equal(gd_lo, gc_lo) and torch.equal(gd_idx, gc_idx)
    # sparse weight gradient
    rps2 = 2
    n_slabs2 = -(-(B * Hp) // rps2)
    part2 = torch.empty(n_slabs2 * 65 * 64 * 64, device=dev)
    dW_sp = torch.empty((64, 64, 5, 13), device=dev)
    _hip.call("mx_conv_block_wgrad_sp_f16", _hip.ptr(gc_hi), _hip.ptr(gc_lo), _hip.ptr(gidx), _hip.ptr(x_hi), _hip.ptr(x_lo),
              _hip.ptr(scale), B, H, W, T, rps2, _hip.ptr(part2), _hip.ptr(dW_sp), st)
    assert rel(dW_sp.cpu(), w.grad) < 1e-5, ("sparse wgrad", rel(dW_sp.cpu(), w.grad))
    assert rel(dW_sp, dW_dense) < 2e-6                                          # same sums, different order
    # data gradient
    wf_hi, wf_lo = am._pack_f16(w.detach().to(dev), 1)
    dxhat = torch.empty((B, 64, H, PITCH), device=dev)
    _hip.call("mx_conv_block_dgrad_f16", _hip.ptr(dz_hi), _hip.ptr(dz_lo), _hip.ptr(wf_hi), _hip.ptr(wf_lo), _hip.ptr(scale),
              B, H, W, T, _hip.ptr(dxhat), st)
    assert rel(dxhat.cpu()[..., :W], xhat_r.grad) < 1e-5
    assert bool((dxhat[..., W:] == 0).all())
    # sparse data gradient: the same pooled operand + fragment-packed weights
    sc2 = torch.empty(2, device=dev)
    _hip.call("mx_conv_prep_dgrad_f16", _hip.ptr(G_d), _hip.ptr(amax_d), B, H, W, _hip.ptr(ws), 0, _hip.ptr(sc2), None, None, st)
    assert torch.equal(sc2, scale)                                                                 # scale-only call
    ws_hi = torch.empty(4 * 3 * 2 * 13 * 2 * 64 * 16, device=dev, dtype=torch.float16)
    ws_lo = torch.empty_like(ws_hi)
    _hip.call("mx_conv_pack_weights_sp_f16", _hip.ptr(w.detach().to(dev).contiguous()), _hip.ptr(ws_hi), _hip.ptr(ws_lo), st)
    dx_sp = torch.empty((B, 64, H, PITCH), device=dev)
    _hip.call("mx_conv_block_dgrad_sp_f16", _hip.ptr(gc_hi), _hip.ptr(gc_lo), _hip.ptr(gc_idx), _hip.ptr(ws_hi), _hip.ptr(ws_lo),
              _hip.ptr(scale), B, H, W, T, _hip.ptr(dx_sp), None, None, None, None, st)
    assert rel(dx_sp.cpu()[..., :W], xhat_r.grad) < 1e-5, ("sparse dgrad", rel(dx_sp.cpu()[..., :W], xhat_r.grad))
    assert bool((dx_sp[..., W:] == 0).all())
    assert rel(dx_sp, dxhat) < 5e-6                                              # same sums, different order
    # the same kernel with the LayerNorm-backward statistics taken in its epilogue: identical dxhat, and per (plane,
    # row, position half) the sums of dxhat and dxhat * xhat
    dx_ln = torch.empty((B, 64, H, PITCH), device=dev)
    ln_part = torch.full((B, 64, H, 2, 2), float("nan"), device=dev)
    _hip.call("mx_conv_block_dgrad_sp_f16", _hip.ptr(gc_hi), _hip.ptr(gc_lo), _hip.ptr(gc_idx), _hip.ptr(ws_hi), _hip.ptr(ws_lo),
              _hip.ptr(scale), B, H, W, T, _hip.ptr(dx_ln), _hip.ptr(x_hi), _hip.ptr(x_lo), _hip.ptr(ln_part), None, st)
    assert torch.equal(dx_ln, dx_sp)
    d64, x64 = dx_sp.double().cpu(), xhat_pair.double().cpu()
    halves = [slice(0, 192), slice(192, PITCH)]                                  # position tiles 0..5 (incl. the shared 5) / 6..10
    want = torch.stack([torch.stack([d64[..., sl].sum(-1), (d64[..., sl] * x64[..., sl]).sum(-1)], -1) for sl in halves], -2)
    got = ln_part.double().cpu()
    assert bool(torch.isfinite(got).all())
    # the middle tile (positions 160..191) is computed by the waves of position half 0 or 1 depending on the wave: only
    # the sum over both halves is pinned
    tot_scale = float(want.abs().sum((2, 3)).max()) + 1e-30
    assert float((got.sum(3) - want.sum(3)).abs().max()) / tot_scale < 1e-6
    # LayerNorm + PReLU backward from those partials == its own two-sweep statistics
    ds_a, gs_a = torch.empty(B * 64, device=dev), torch.empty(B * 64, device=dev)
    ds_b, gs_b = torch.empty(B * 64, device=dev), torch.empty(B * 64, device=dev)
    g_a, g_b = dx_sp.clone(), dx_sp.clone()
    _hip.call("mx_ln_prelu_bwd", _hip.ptr(x_d), _hip.ptr(g_a), _hip.ptr(stats), _hip.ptr(sl_d), B, 64, H, W, _hip.ptr(ds_a),
              _hip.ptr(gs_a), None, None, st)
    _hip.call("mx_ln_prelu_bwd", _hip.ptr(x_d), _hip.ptr(g_b), _hip.ptr(stats), _hip.ptr(sl_d), B, 64, H, W, _hip.ptr(ds_b),
              _hip.ptr(gs_b), None, _hip.ptr(ln_part), st)
    assert rel(g_b, g_a) < 2e-6 and rel(ds_b, ds_a) < 2e-6 and rel(gs_b, gs_a) < 1e-5
    assert rel(g_b.cpu()[..., :W], x_req.grad) < 1e-5, ("ln_prelu_bwd from partials", rel(g_b.cpu()[..., :W], x_req.grad))
    # ... and the same pass written straight into the pooled operand of the block below (dL/dp never in fp32): the data gradient
    # also leaves max|dxhat| and max|xhat|, from which mx_ln_bwd_finish BOUNDS max|G| for the f16x3 scale
    gx = torch.zeros(2, device=dev, dtype=torch.int32)
    _hip.call("mx_conv_block_dgrad_sp_f16", _hip.ptr(gc_hi), _hip.ptr(gc_lo), _hip.ptr(gc_idx), _hip.ptr(ws_hi), _hip.ptr(ws_lo),
              _hip.ptr(scale), B, H, W, T, _hip.ptr(dx_ln), _hip.ptr(x_hi), _hip.ptr(x_lo), _hip.ptr(ln_part), _hip.ptr(gx), st)
    assert torch.equal(dx_ln, dx_sp)
    mx = gx.view(torch.float32).cpu()
    assert float(mx[0]) == float(dx_sp.abs().max()) and float(mx[1]) == float(xhat_pair[..., :W].abs().max())
    m12 = torch.empty((B, 64, 2), device=dev)
    bound_ws = torch.empty(1, device=dev, dtype=torch.int32)
    sc3 = torch.empty(2, device=dev)
    _hip.call("mx_ln_bwd_finish", _hip.ptr(ln_part), _hip.ptr(stats), _hip.ptr(sl_d), _hip.ptr(gx), B, 64, H, W, _hip.ptr(m12),
              _hip.ptr(bound_ws), _hip.ptr(sc3), st)
    S3, g_max = float(sc3[0]), float(g_b.abs().max())
    assert S3 == 2.0 ** round(np.log2(S3)) and float(sc3[1]) == 1.0 / S3
    assert 16.0 <= g_max * S3 < 1024.0, (g_max * S3,)          # a true bound, and within 2^6 of the maximum itself
    am2 = (torch.rand(B, 64, H, PITCH, device=dev) < 0.5).to(torch.uint8)
    shp = (B, H, 4, PITCH, 16)
    fh, fl = torch.empty(shp, device=dev, dtype=torch.float16), torch.empty(shp, device=dev, dtype=torch.float16)
    fi = torch.empty((B, H, 4, PITCH), device=dev, dtype=torch.int32)
    fp = torch.empty((B, 64, H, 22, 2), device=dev, dtype=torch.int16)
    part2 = torch.empty((B, 64, H, 6, 2), device=dev)
    ds_c, gs_c = torch.empty(B * 64, device=dev), torch.empty(B * 64, device=dev)
    _hip.call("mx_ln_prelu_bwd_gpool_f16", _hip.ptr(x_d), _hip.ptr(dx_sp), _hip.ptr(am2), _hip.ptr(stats), _hip.ptr(sl_d),
              _hip.ptr(m12), _hip.ptr(sc3), B, H, W, _hip.ptr(fh), _hip.ptr(fl), _hip.ptr(fi), _hip.ptr(fp), _hip.ptr(part2),
              _hip.ptr(ds_c), _hip.ptr(gs_c), st)
    # the two-pass route on the same scale: G (g_b) -> pooled operand
    rh, rl, ri, rp = torch.empty_like(fh), torch.empty_like(fl), torch.empty_like(fi), torch.empty_like(fp)
    _hip.call("mx_conv_prep_gpool_cl_f16", _hip.ptr(g_b), _hip.ptr(am2), _hip.ptr(sc3), B, 2 * H, W, _hip.ptr(rh), _hip.ptr(rl),
              _hip.ptr(ri), _hip.ptr(rp), st)
    assert torch.equal(fi, ri) and torch.equal(fp, rp)
    fused_pair, split_pair = fh.float() + fl.float(), rh.float() + rl.float()
    assert rel(fused_pair, split_pair) < 2e-6
    back = fused_pair.permute(0, 2, 4, 1, 3).reshape(B, 64, H, PITCH) / S3
    assert rel(back.cpu()[..., :W], x_req.grad) < 1e-5 and bool((back[..., W:] == 0).all())
    assert rel(ds_c, ds_b) < 2e-6 and rel(gs_c, gs_b) < 1e-5


def test_fused_plane_stats_large_offset_tiny_variance(dev):
    """the LayerNorm statistics the forward epilogue leaves (fp32 row sums) must not cancel for planes with
    |mean| >> std -- near-constant planes of silent clips or dead channels, where out ~ bias.  The row sums are taken of
    PReLU(out) - PReLU(bias), so the variance keeps its digits; checked against an fp64 evaluation of the stored plane and
    against the fp64 sweep (mx_plane_stats), for positive and negative biases of magnitude ~20 over a std of ~1e-3, and
    for an exactly constant plane (zero weights: rstd = 1 / sqrt(eps))."""
    from mod_extraction_amd import _hip, models as am
    torch.manual_seed(11)
    B, H, W, T = 2, 8, 345, 1
    st = _hip.stream()
    x_d = to_planes(torch.randn(B, 64, H, W), dev)
    sl_d = (torch.rand(64) * 0.4 + 0.05).to(dev)
    stats = torch.empty((B, 64, 2), device=dev)
    _hip.call("mx_plane_stats", _hip.ptr(x_d), _hip.ptr(sl_d), B, 64, H, W, 1e-5, _hip.ptr(stats), st)
    x_hi = torch.empty((B, H, 4, PITCH, 16), device=dev, dtype=torch.float16)
    x_lo = torch.empty_like(x_hi)
    _hip.call("mx_conv_prep_fwd_f16", _hip.ptr(x_d), _hip.ptr(stats), _hip.ptr(sl_d), B, H, W, _hip.ptr(x_hi), _hip.ptr(x_lo), st)
    bias = (20.0 + torch.randn(64)) * torch.where(torch.arange(64) % 2 == 0, 1.0, -1.0)
    sl_out = (torch.rand(64) * 0.4 + 0.05).to(dev)
    for wscale in (1e-4, 0.0):
        w = torch.randn(64, 64, 5, 13) * wscale
        w_hi, w_lo = am._pack_f16(w.to(dev), 0)
        p = torch.empty((B, 64, H // 2, PITCH), device=dev)
        amax = torch.empty((B, 64, H // 2, PITCH), device=dev, dtype=torch.uint8)
        st_part = torch.empty((B, H // 2, 64, 2), device=dev)
        _hip.call("mx_conv_block_fwd_f16", _hip.ptr(x_hi), _hip.ptr(x_lo), _hip.ptr(w_hi), _hip.ptr(w_lo), _hip.ptr(bias.to(dev)),
                  B, H, W, T, _hip.ptr(p), _hip.ptr(amax), _hip.ptr(sl_out), _hip.ptr(st_part), st)
        fused, swept = torch.empty((B, 64, 2), device=dev), torch.empty((B, 64, 2), device=dev)
        _hip.call("mx_plane_stats_finish", _hip.ptr(st_part), _hip.ptr(bias.to(dev)), _hip.ptr(sl_out), B, 64, H // 2, W, 1e-5,
                  _hip.ptr(fused), st)
        _hip.call("mx_plane_stats", _hip.ptr(p), _hip.ptr(sl_out), B, 64, H // 2, W, 1e-5, _hip.ptr(swept), st)
        pv = p.cpu()[..., :W].double()
        y = torch.where(pv > 0, pv, sl_out.cpu().double().view(1, 64, 1, 1) * pv)
        mean_r, var_r = y.mean(dim=(2, 3)), y.var(dim=(2, 3), unbiased=False)
        if wscale:
            assert float(var_r.sqrt().max()) < 0.1 and float(mean_r.abs().min()) > 0.5       # the regime under test
        assert float(((fused[..., 0].cpu().double() - mean_r).abs() / mean_r.abs()).max()) < 2e-7
        assert float((fused[..., 1].cpu().double() * (var_r + 1e-5).sqrt() - 1).abs().max()) < 1e-4
        assert float((fused[..., 1] / swept[..., 1] - 1).abs().max()) < 1e-4
