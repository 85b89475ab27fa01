// ABI version of the rcmarl C interface (include/rcmarl.h).
#include "rcmarl_common.h"
RCMARL_EXPORT int rcmarl_abi_version(void) { return 1; }
