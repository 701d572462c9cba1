// hope_rs.hip -- Reeds-Shepp feasibility search kernel (placeholder wiring; real kernel follows)
#include "hope_dev.h"
#include "hope_internal.h"

namespace hope {
size_t rs_lds_bytes(int max_obst) { return (size_t)max_obst * 64 + 4096; }
hipError_t launch_rs_search(const RsParams& p, hipStream_t stream) {
    (void)p; (void)stream;
    return hipSuccess;
}
}  // namespace hope
