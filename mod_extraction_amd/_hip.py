"""ctypes binding of libmodex_hip.so (the C ABI declared in include/modex_hip.h).

No fallback: if the shared library is missing, or a call returns a non-zero status, this raises.
Device pointers are borrowed from torch tensors; launches go to torch's current HIP stream.
"""
import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MODEX_HIP_LIB: load another build of the same library (kernel experiments); there is no non-HIP fallback
SO_PATH = os.environ.get("MODEX_HIP_LIB") or os.path.join(_HERE, "_lib", "libmodex_hip.so")
ABI_VERSION = 16

_ERR = {-1: "MX_ERR_ARG (bad argument)", -2: "MX_ERR_UNSUPPORTED (size not supported)",
        -3: "MX_ERR_LAUNCH (HIP launch error)"}

_P, _I64, _I32, _F32, _F64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_double

# name -> argtypes; must list every symbol include/modex_hip.h declares (tests check this)
SIGNATURES = {
    "mx_abi_version": [],
    "mx_lfo_synth": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _F32, _P, _P],
    "mx_uniform_rows": [_P, _I64, _P, _I64, _P, _I64, _I64, ctypes.c_uint64, ctypes.c_uint32, _F32, _F32, _P],
    "mx_interp_linear": [_P, _I64, _I64, _I64, _P, _P],
    "mx_interp_linear_bwd": [_P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P],
    "mx_flanger_fwd": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _I64, _I64, _I64,
                       _P, _I64, _P, _P, _P, _P],
    "mx_lds_roundtrip_probe": [_I64, _P, _P],
    "mx_lstm_step_probe": [_I32, _I64, _P, _P],
    "mx_lstm_bwd_dgate": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _F32, _P, _P, _I64, _I64, _P],
    "mx_lstm_dlfo": [_P, _P, _I64, _I64, _P, _I64, _P],
    "mx_phaser_cascade_probe": [_I64, _P, _P],
    "mx_phaser_fwd": [_P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _F64, _I32, _P, _I64, _P, _P, _I64, _P],
    "mx_logmel_fwd": [_P, _I64, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _F32, _I32, _I32,
                      _I32, _I32, _P, _P],
    "mx_conv_pack_weights": [_P, _I64, _I64, _I32, _P, _P],
    "mx_plane_stats": [_P, _P, _I64, _I64, _I64, _I64, _F32, _P, _P],
    "mx_plane_stats_finish": [_P, _P, _P, _I64, _I64, _I64, _I64, _F32, _P, _P],
    "mx_conv_block_fwd": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I32, _I32, _P, _P, _P],
    "mx_conv_block_dgrad": [_P, _P, _P, _I64, _I64, _I64, _I32, _P, _P],
    "mx_conv_pack_weights_f16": [_P, _I32, _P, _P, _P],
    "mx_conv_prep_fwd_f16": [_P, _P, _P, _I64, _I64, _I64, _P, _P, _P],
    "mx_conv_prep_dgrad_f16": [_P, _P, _I64, _I64, _I64, _P, _I32, _P, _P, _P, _P],
    "mx_conv_block_fwd_f16": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _P, _P, _P, _P, _P],
    "mx_conv_block_dgrad_f16": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _P, _P],
    "mx_conv_pack_weights_kvec_f16": [_P, _P, _P, _P],
    "mx_conv_prep_fwd_kvec_f16": [_P, _P, _I64, _I64, _I64, _P, _P, _P],
    "mx_conv_block1_fwd_f16": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P],
    "mx_conv_block_wgrad_sp_f16": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I64, _P, _P, _P],
    "mx_conv_pack_weights_sp_f16": [_P, _P, _P, _P],
    "mx_conv_prep_gpool_cl_f16": [_P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P],
    "mx_conv_block_dgrad_sp_f16": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _P, _P, _P, _P, _P, _P],
    "mx_conv_block1_wgrad_f16": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P],
    "mx_conv_block1_wgrad_pair_f16": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P],
    "mx_conv_block_wgrad_f16": [_P, _P, _P, _P, _P, _I64, _I64, _I32, _I64, _P, _P, _P],
    "mx_conv_block_wgrad": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I32, _I64, _P, _P, _P],
    "mx_ln_bwd_finish": [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P],
    "mx_ln_prelu_bwd_gpool_f16": [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P],
    "mx_ln_prelu_bwd": [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P],
    "mx_ln_prelu_bwd_pair": [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P],
    "mx_reduce_rows": [_P, _I64, _I64, _I32, _P, _P],
    "mx_plane_sum": [_P, _I64, _I64, _I64, _P, _P],
    "mx_head_fwd": [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P],
    "mx_head_bwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P],
    "mx_lfo_loss": [_P, _P, _I64, _I64, _F32, _F32, _F32, _F32, _P, _P, _P, _P],
    "mx_smoothen": [_P, _I64, _I64, _I64, _P, _P],
    "mx_find_corners": [_P, _I64, _I64, _P, _P, _P],
    "mx_stretch_corners": [_P, _I64, _I64, _I64, _P, _P],
    "mx_stretch_corners_bwd": [_P, _P, _I64, _I64, _I64, _P, _P],
    "mx_check_mod_sig": [_P, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P],
    "mx_lstm_fwd": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I64, _P],
    "mx_lstm_bwd_l1": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _F32, _P, _I64, _I64, _P],
    "mx_lstm_bwd": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _I64, _I64, _P],
    "mx_sgemm_f32": [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I32, _P],
    "mx_tcn_im2col": [_P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P],
    "mx_tcn_col2im": [_P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P],
    "mx_tcn_act_fwd": [_P, _P, _P, _P, _I64, _I64, _I64, _P, _P],
    "mx_tcn_act_bwd": [_P, _P, _P, _I64, _I64, _I64, _P, _P, _P],
    "mx_tcn_ln_bwd": [_P, _P, _P, _P, _I64, _I64, _I64, _P, _P],
    "mx_im2col2d": [_P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P],
    "mx_col2im2d": [_P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P],
    "mx_rowln_fwd": [_P, _I64, _I64, _F32, _P, _P, _P],
    "mx_rowln_bwd": [_P, _P, _P, _I64, _I64, _P, _P],
    "mx_row_sums": [_P, _I64, _I64, _P, _P],
    "mx_pool_prelu_fwd": [_P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P],
    "mx_pool_prelu_bwd": [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, _P],
    "mx_binmean_head_fwd": [_P, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, _P, _P],
    "mx_binmean_head_bwd": [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P],
    "mx_lstmg_fwd": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P],
    "mx_lstmg_bwd": [_P, _P, _P, _P, _I64, _I64, _I64, _P, _P],
    "mx_lstmg_out_fwd": [_P, _P, _P, _I64, _I64, _I64, _I64, _P, _P],
    "mx_lstmg_out_bwd": [_P, _P, _I64, _I64, _I64, _I64, _P, _P],
    "mx_chan_stats": [_P, _I64, _I64, _I64, _P, _P],
    "mx_chan_norm_fwd": [_P, _P, _I64, _I64, _I64, _P, _P],
    "mx_chan_norm_bwd": [_P, _P, _P, _I64, _I64, _I64, _I32, _P, _P],
    "mx_film_fwd": [_P, _P, _I64, _I64, _I64, _P, _P],
    "mx_film_bwd": [_P, _P, _P, _I64, _I64, _I64, _P, _P, _P],
    "mx_prelu_res_fwd": [_P, _P, _P, _I64, _I64, _I64, _P, _P],
    "mx_prelu_res_bwd": [_P, _P, _P, _I64, _I64, _I64, _P, _P, _P],
    "mx_effect_loss_sums": [_P, _I64, _P, _I64, _I64, _I64, _P, _P],
    "mx_effect_loss_grad": [_P, _I64, _P, _I64, _I64, _I64, _F32, _F32, _F32, _F32, _F32, _I32, _P, _I64, _P],
    "mx_mrstft_loss": [_P, _I64, _P, _I64, _I64, _I64, _I32, _P, _P, _P, _P, _F32, _F32, _F32, _P, _P, _P, _P, _P,
                       _I64, _P],
    "mx_adamw_step": [_P, _P, _P, _P, _I64, _I64, _F32, _F32, _F32, _F32, _F32, _F32, _P],
    "mx_reduce_rows_adamw_step": [_P, _I64, _P, _P, _P, _P, _I64, _I64, _F32, _F32, _F32, _F32, _F32, _F32, _P],
}

# measurement twins (same signatures): the dependent chain of the sample-recurrent kernels without global traffic
PROBE_TWINS = ("mx_flanger_fwd", "mx_phaser_fwd", "mx_lstm_fwd", "mx_lstm_bwd_l1")
for _name in PROBE_TWINS:
    SIGNATURES[_name + "_probe"] = SIGNATURES[_name]

_lib: Optional[ctypes.CDLL] = None


class HipLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.isfile(SO_PATH):
            raise HipLibraryError(
                f"{SO_PATH} not found: build it with `python -m mod_extraction_amd.build` "
                "(there is no CPU fallback)")
        lib = ctypes.CDLL(SO_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        if lib.mx_abi_version() != ABI_VERSION:
            raise HipLibraryError(f"ABI mismatch: library {lib.mx_abi_version()} != binding {ABI_VERSION}")
        _lib = lib
    return _lib


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("mod_extraction_amd ops need tensors on a HIP device (no CPU fallback)")
    if not t.is_contiguous():
        raise HipLibraryError("non-contiguous tensor passed to a HIP kernel")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional per-entry-point device timing with HIP events on the launch stream (bench.py).

    ``with KernelTimer({"mx_conv_block_fwd"}) as kt: ...`` records an event pair around every call of
    the selected C-ABI entry points (they launch on torch's current stream, so ``torch.cuda.Event``
    brackets exactly the kernel(s) of that call); ``kt.results()`` synchronises and returns
    ``{tag: [ms, ...]}`` where tag = ``name`` or ``name#<key>`` if ``key_fn(name, args)`` is given."""

    active: Optional["KernelTimer"] = None

    def __init__(self, names, key_fn=None) -> None:
        self.names, self.key_fn = set(names), key_fn
        self.pairs = []

    def __enter__(self) -> "KernelTimer":
        KernelTimer.active = self
        return self

    def __exit__(self, *exc) -> None:
        KernelTimer.active = None

    def results(self):
        torch.cuda.synchronize()
        out = {}
        for tag, a, b in self.pairs:
            out.setdefault(tag, []).append(a.elapsed_time(b))
        return out


class probe_twins:
    """bench.py / tools only: inside this context the four sample-recurrent entry points are routed to their
    ``*_probe`` twins (serial-floor measurement; results are meaningless).  Host-side routing of THIS binding -- the
    shared library itself has no mode switch."""
    on = False

    def __enter__(self):
        probe_twins.on = True
        return self

    def __exit__(self, *exc):
        probe_twins.on = False


def call(name: str, *args) -> None:
    kt = KernelTimer.active
    if probe_twins.on and name in PROBE_TWINS:
        fn_name = name + "_probe"
    else:
        fn_name = name
    if kt is not None and name in kt.names:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = getattr(load(), fn_name)(*args)
        b.record()
        kt.pairs.append((name if kt.key_fn is None else f"{name}#{kt.key_fn(name, args)}", a, b))
    else:
        rc = getattr(load(), fn_name)(*args)
    if rc != 0:
        raise HipLibraryError(f"{name} failed: {_ERR.get(rc, rc)}")
