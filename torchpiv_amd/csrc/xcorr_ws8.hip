// Tile kernels for 8x8 interrogation windows (see xcorr_tile.hpp).
#include "xcorr_tile.hpp"
namespace tpiv {
hipError_t launch_xcorr_ws8(const PassParams& p, int mode, int n_cu, hipStream_t stream) {
    return launch_xcorr_tile_ws<8>(p, mode, n_cu, stream);
}
hipError_t launch_xcorr_cand_ws8(const PassParams& p, int n_cu, hipStream_t stream) {
    return launch_xcorr_tile_cand_ws<8>(p, n_cu, stream);
}
hipError_t launch_peak_debug_ws8(const PassParams& p, const float* maps, int n_maps, int planar, hipStream_t stream) {
    return launch_peak_debug<8>(p, maps, n_maps, planar, stream);
}
}  // namespace tpiv
