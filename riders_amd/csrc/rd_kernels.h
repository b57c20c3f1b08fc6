// Internal launcher declarations shared by the kernel translation units and rd_api.cpp.
// Nothing here is part of the public C ABI (that is include/riders_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "rd_types.h"
#ifdef RD_HALF_F16
#define rd rd_f16          // fp16 build of the kernels (see rd_common.h)
#endif
#include "rd_kernels_decl.h"
