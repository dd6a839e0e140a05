// abi.hip -- ABI version of libmodex_hip.so (see include/modex_hip.h).  The library keeps no caller-observable mutable state:
// every entry point is a function of its arguments and the stream it is given.  The only statics are (i) per-DEVICE latches of
// an idempotent driver call (hipFuncAttributeMaxDynamicSharedMemorySize, common.h:mx_set_dyn_lds) and (ii) A/B knobs read once
// from the environment (MODEX_MFMA_SHAPE, MODEX_PATCH_RING, MODEX_BLOCK1_PERSIST, MODEX_LSTM_KQ: kernel variants that compute the
// same results; DESIGN.md names each).
#include "common.h"
MX_EXPORT int mx_abi_version(void) { return 16; }
