// 128x128 interrogation windows, first pass: the two-threads-per-line kernel of xcorr_big.hpp (64-point
// in-register codelets, planar LDS transposes).  Shifted 128x128 passes are only reachable with a
// 256-pixel first pass and run the generic-size kernel (xcorr_generic.hip).
#include "xcorr_big.hpp"
namespace tpiv {
hipError_t launch_xcorr_big128_pass1(const PassParams& p, int n_cu, hipStream_t stream) {
    return launch_xcorr_big128(p, n_cu, stream);
}
hipError_t launch_xcorr_cand_ws128(const PassParams& p, int n_cu, hipStream_t stream) {
    return launch_xcorr_big128(p, n_cu, stream, true);
}
hipError_t launch_peak_debug_ws128(const PassParams& p, const float* maps, int n_maps, hipStream_t stream) {
    return launch_peak_debug_big(p, maps, n_maps, stream);
}
}  // namespace tpiv
