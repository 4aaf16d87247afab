// peaq_fb.hip -- advanced mode: 40-band filter-bank ear model (fbearmodel.c).
#include <hip/hip_runtime.h>

#include "peaq_device.h"
#include "peaq_kernels.h"
#include "peaq_wave.h"

namespace peaq {

hipError_t launch_fb_frontend(const FbFrontArgs&, unsigned, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_fb_backend(const FbBackendArgs&, unsigned, hipStream_t) { return hipErrorNotSupported; }

}  // namespace peaq
