// Tile kernels for 8x8 interrogation windows (see xcorr_tile.hpp).
#include "xcorr_tile.hpp"
namespace tpiv {
hipError_t launch_xcorr_ws8(const PassParams& p, int mode, int n_cu, hipStream_t stream) {
    return launch_xcorr_tile_ws<8>(p, mode, n_cu, stream);
}
}  // namespace tpiv
