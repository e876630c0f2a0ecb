#pragma once
#include <hip/hip_runtime.h>
#include "../../include/xrt_hip.h"
namespace xrt {
hipError_t geosource_shine_launch(const xrt_hip_geosource& G, const xrt_hip_beam& out,
                                  hipStream_t st);
hipError_t geosource_probe_launch(const xrt_hip_geosource& G, int64_t n, int32_t* flag,
                                  hipStream_t st);
}
