// Tile kernels for 32x32 interrogation windows (see xcorr_tile.hpp).
#include "xcorr_tile.hpp"
namespace tpiv {
hipError_t launch_xcorr_ws32_cws(const PassParams& p, int n_cu, hipStream_t stream);      // xcorr_ws32c.hip
hipError_t launch_xcorr_ws32(const PassParams& p, int mode, int n_cu, hipStream_t stream) {
    switch (mode) {
        case MODE_PASS1: return launch_tile<32, MODE_PASS1>(p, n_cu, stream);
        case MODE_DWS: return launch_tile<32, MODE_DWS>(p, n_cu, stream);
        case MODE_CWS: return launch_xcorr_ws32_cws(p, n_cu, stream);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_xcorr_cand_ws32(const PassParams& p, int n_cu, hipStream_t stream) {
    return launch_xcorr_tile_cand_ws<32>(p, n_cu, stream);
}
hipError_t launch_peak_debug_ws32(const PassParams& p, const float* maps, int n_maps, int planar, hipStream_t stream) {
    return launch_peak_debug<32>(p, maps, n_maps, planar, stream);
}
}  // namespace tpiv
