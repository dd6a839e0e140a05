// abi.hip -- ABI version of libmodex_hip.so (see include/modex_hip.h) and the measurement switch of bench.py.
#include "common.h"
int g_mx_probe = 0;
MX_EXPORT int mx_abi_version(void) { return 2; }
// mode != 0: the sample-recurrent kernels (flanger, phaser, LSTM forward / backward) run their dependent chain with
// NO global-memory traffic inside the loop (inputs are constants, outputs are dropped): the "serial floor" that
// bench.py reports next to each kernel's real duration.  Results are meaningless in that mode.
MX_EXPORT int mx_set_probe_mode(int32_t mode)
{
    g_mx_probe = mode;
    return MX_OK;
}
