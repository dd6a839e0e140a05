"""CU-partitioned HIP streams for the effect-modelling step.

The truncated-BPTT recurrence occupies one workgroup (one CU) per clip and is bound by the latency of a step; the batch
render and the frozen extractor of the NEXT batch run concurrently on a side stream.  Left to the dispatcher the two
share CUs: the matrix-core convolutions of the extractor slow every concurrent LSTM launch by ~13 %.
``hipExtStreamCreateWithCUMask`` gives each stream its own, disjoint set of CUs.

WHAT A MASK BIT IS (measured, tools/probe/probe_xcc.py -> profiles/r03/xcc_mask_probe.txt; round 2 had this wrong):
on this 8-XCD part the driver deals the mask bits round-robin over the XCDs -- bit i belongs to XCD i % 8 -- so ONE
32-bit word is 4 CUs ON EVERY XCD, not one XCD: a stream with word 0 alone runs on 32 distinct CUs, 4 per XCD.  The
default split (5 words / 3 words) is therefore 20 CUs of each XCD for the recurrence and the other 12 of each XCD for
the prefetch work: disjoint CUs, shared L2 slices and clocks.  A whole XCD cannot be carved out through this API: a mask
that leaves an XCD without any CU is not honoured for that XCD (it then runs on all 32 of its CUs -- the probe's
"every 8th bit" mask landed on all 256 CUs), which is also why round 2's "masks that split CUs within the XCDs" showed
no gain: those masks were not in force.

Measured on config 4 (128 clips, MI355X, tools/exp_cumask.py), ms per batch:
    no masks 81.2 | 4 + 4 words 75.6 | 5 + 3 words 73.2 | 6 + 2 words 72.3
Not for the LFO-extraction step: there the main stream is throughput-bound on all 256 CUs, and masking only the (light)
side stream made the step 14 % SLOWER (75.9 -> 86.5 ms): a masked queue next to torch's default queue serialises the
two.  With BOTH streams created through hipExtStreamCreateWithCUMask (tools/exp_cumask_headline.py): main on every CU +
side on one word 74.6 ms against 75.1 with the default streams -- 0.7 %, not adopted.
"""
import atexit
import ctypes
import os
from typing import Dict, List, Optional, Tuple

import torch

_cache: Dict[Tuple[int, int], Tuple[torch.cuda.Stream, torch.cuda.Stream]] = {}
_raw: List[Tuple[int, int]] = []            # (device index, hipStream_t) of every masked stream this module created


def _destroy_masked_streams() -> None:
    """The masked streams are created behind torch's back (ExternalStream does not own them), so nobody destroys them:
    left alive until HIP's own teardown they crashed the process at exit (config 4 under the profiler: rc 139 after the
    result line was printed, profiles/r03/notes.txt).  Drain and destroy them while the runtime is still up."""
    if not _raw:
        return
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
        hip.hipSetDevice.argtypes = [ctypes.c_int]
        for idx, handle in _raw:
            hip.hipSetDevice(idx)
            hip.hipStreamSynchronize(ctypes.c_void_p(handle))
            hip.hipStreamDestroy(ctypes.c_void_p(handle))
    except Exception:                        # interpreter shutdown: never turn a clean exit into a traceback
        pass
    _raw.clear()
    _cache.clear()


atexit.register(_destroy_masked_streams)


def _masked_stream(hip, words, device) -> torch.cuda.Stream:
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    _raw.append((device.index if device.index is not None else torch.cuda.current_device(), st.value))
    return torch.cuda.ExternalStream(st.value, device=device)


def cu_partition(device: torch.device, side_words: Optional[int] = None, main_workgroups: int = 0
                 ) -> Optional[Tuple[torch.cuda.Stream, torch.cuda.Stream]]:
    """(main, side) streams on disjoint sets of CUs (see the module docstring for what a mask word is), or None when
    the device cannot be partitioned (CU count not a multiple of 32, fewer than 4 mask words, ``MODEX_CU_PARTITION=0``)
    or when ``main_workgroups`` -- the number of one-CU workgroups the main stream must keep resident at once, i.e. the
    clips per batch -- exceeds the main stream's CUs (a second wave of workgroups would cost more than the partition
    gains).  The pair is created once per device and kept.  ``side_words`` defaults to ``MODEX_SIDE_WORDS`` (or the
    older name ``MODEX_SIDE_XCDS``) or 3."""
    if device.type != "cuda" or os.environ.get("MODEX_CU_PARTITION", "1") == "0":
        return None
    if side_words is None:
        side_words = int(os.environ.get("MODEX_SIDE_WORDS") or os.environ.get("MODEX_SIDE_XCDS") or "3")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    n_cu = torch.cuda.get_device_properties(idx).multi_processor_count
    words = n_cu // 32
    if n_cu % 32 or words < 4 or not 0 < side_words < words:
        return None
    if main_workgroups > 32 * (words - side_words):
        return None
    key = (idx, side_words)
    if key in _cache:
        return _cache[key]
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    full = 0xFFFFFFFF
    with torch.cuda.device(idx):
        main = _masked_stream(hip, [full] * (words - side_words) + [0] * side_words, torch.device("cuda", idx))
        side = _masked_stream(hip, [0] * (words - side_words) + [full] * side_words, torch.device("cuda", idx))
    _cache[key] = (main, side)
    return main, side
