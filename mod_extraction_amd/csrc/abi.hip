// abi.hip -- ABI version of libmodex_hip.so (see include/modex_hip.h).  The library keeps no mutable global state: every
// entry point is a pure function of its arguments and the stream it is given.
#include "common.h"
MX_EXPORT int mx_abi_version(void) { return 10; }
