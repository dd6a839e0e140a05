"""XCD-partitioned HIP streams for the effect-modelling step.

The truncated-BPTT recurrence occupies one workgroup (one CU) per clip and is bound by the latency of a step; the batch
render and the frozen extractor of the NEXT batch run concurrently on a side stream.  Left to the dispatcher the two
share XCDs -- their L2 slices and, it appears, their clocks: the matrix-core convolutions of the extractor slow every
concurrent LSTM launch by ~13 %.  ``hipExtStreamCreateWithCUMask`` confines each stream to whole XCDs (32 CUs = one mask
word on gfx950): measured on config 4 (128 clips, MI355X, tools/exp_cumask.py), ms per batch:
    no masks 81.2 | main 4 XCDs + side 4 XCDs 75.6 | main 5 + side 3 73.2 | main 6 + side 2 72.3 | masks sharing XCDs 81.2
The default gives the recurrence 5 of 8 XCDs and the prefetch work 3 (robust when the prefetch work grows).
Not for the LFO-extraction step: there the main stream is throughput-bound on all 256 CUs, and masking only the (light)
side stream made the step 14 % SLOWER (75.9 -> 86.5 ms, 1, 2 or 4 side XCDs alike): a masked queue next to torch's default
queue serialises the two.  With BOTH streams created through hipExtStreamCreateWithCUMask (tools/exp_cumask_headline.py):
main on every CU + side on one XCD 74.6 ms against 75.1 with the default streams -- 0.7 %, not adopted; main on 7 XCDs + side
on the 8th 80.5; side on 8 CUs of every XCD 79.1.
"""
import ctypes
import os
from typing import Dict, Optional, Tuple

import torch

_cache: Dict[Tuple[int, int], Tuple[torch.cuda.Stream, torch.cuda.Stream]] = {}


def _masked_stream(hip, words, device) -> torch.cuda.Stream:
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(st.value, device=device)


def xcd_partition(device: torch.device, side_xcds: Optional[int] = None) -> Optional[Tuple[torch.cuda.Stream, torch.cuda.Stream]]:
    """(main, side) streams on disjoint sets of XCDs, or None when the device cannot be partitioned (not a multiple of
    32 CUs, fewer than 4 XCDs, or ``MODEX_CU_PARTITION=0``).  The pair is created once per device and kept.
    ``side_xcds`` defaults to ``MODEX_SIDE_XCDS`` or 3."""
    if device.type != "cuda" or os.environ.get("MODEX_CU_PARTITION", "1") == "0":
        return None
    if side_xcds is None:
        side_xcds = int(os.environ.get("MODEX_SIDE_XCDS", "3"))
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, side_xcds)
    if key in _cache:
        return _cache[key]
    n_cu = torch.cuda.get_device_properties(idx).multi_processor_count
    words = n_cu // 32
    if n_cu % 32 or words < 4 or not 0 < side_xcds < words:
        return None
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    full = 0xFFFFFFFF
    with torch.cuda.device(idx):
        main = _masked_stream(hip, [full] * (words - side_xcds) + [0] * side_xcds, torch.device("cuda", idx))
        side = _masked_stream(hip, [0] * (words - side_xcds) + [full] * side_xcds, torch.device("cuda", idx))
    _cache[key] = (main, side)
    return main, side

