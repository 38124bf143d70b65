// Tile kernels for 64x64 interrogation windows (see xcorr_tile.hpp).
#include "xcorr_tile.hpp"
namespace tpiv {
hipError_t launch_xcorr_ws64(const PassParams& p, int mode, int n_cu, hipStream_t stream) {
    return launch_xcorr_tile_ws<64>(p, mode, n_cu, stream);
}
hipError_t launch_xcorr_cand_ws64(const PassParams& p, int n_cu, hipStream_t stream) {
    return launch_xcorr_tile_cand_ws<64>(p, n_cu, stream);
}
hipError_t launch_peak_debug_ws64(const PassParams& p, const float* maps, int n_maps, int planar, hipStream_t stream) {
    return launch_peak_debug<64>(p, maps, n_maps, planar, stream);
}
}  // namespace tpiv
