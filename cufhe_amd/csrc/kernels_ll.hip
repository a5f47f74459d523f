// kernels_ll.hip -- translation unit of the low-latency blind-rotate kernels (kernels_ll.hip.h), compiled with the
// max-ilp machine-scheduling strategy (cufhe_amd/build.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_ll.hip.h"
