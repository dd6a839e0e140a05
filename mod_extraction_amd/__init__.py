"""mod_extraction_amd -- MI355X-native (gfx950) hot path of christhetree/mod_extraction.

Python host code mirrors the reference's module surface (``fx``, ``modulations``, ``util``,
``models``, ``losses``, ``lightning``); all arithmetic runs in hand-written HIP kernels behind the
C ABI of ``include/modex_hip.h`` (``_lib/libmodex_hip.so``).  There is no CPU fallback: calling
any op without the built library or without a GPU raises.
"""
__version__ = "0.1.0"
