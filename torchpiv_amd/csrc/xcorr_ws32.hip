// Tile kernels for 32x32 interrogation windows (see xcorr_tile.hpp).
#include "xcorr_tile.hpp"
namespace tpiv {
hipError_t launch_xcorr_ws32(const PassParams& p, int mode, int n_cu, hipStream_t stream) {
    return launch_xcorr_tile_ws<32>(p, mode, n_cu, stream);
}
}  // namespace tpiv
