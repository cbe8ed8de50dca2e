// lev_wide.hip -- wide-band kernel (one workgroup per pair) -- placeholder until implemented.
#include <hip/hip_runtime.h>

#include "ta_internal.h"

namespace ta {
bool lev_wide_fits(uint32_t) { return false; }
hipError_t lev_wide_launch(const LevParams &, bool, hipStream_t, uint32_t *, uint32_t *, uint32_t *, uint32_t *) {
    return hipErrorNotSupported;
}
}  // namespace ta
