// abi.hip -- ABI version of libmodex_hip.so (see include/modex_hip.h).
#include "common.h"
MX_EXPORT int mx_abi_version(void) { return 1; }
