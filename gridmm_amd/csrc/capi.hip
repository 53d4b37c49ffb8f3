// ABI bookkeeping for libgridmm_hip.so (the kernels live in the sibling .hip files).
#include "common.h"

extern "C" int gridmm_abi_version(void) { return 20; }
