// Tile kernels for 128x128 interrogation windows: first-generation kernel (lane = column,
// whole tile in LDS, see xcorr_kernel.hpp); 128-point lines do not fit the per-lane layout of
// xcorr_tile.hpp.
#include "xcorr_kernel.hpp"
namespace tpiv {
hipError_t launch_xcorr_ws128(const PassParams& p, int mode, int n_cu, hipStream_t stream) {
    return launch_xcorr_ws<128>(p, mode, n_cu, stream);
}
}  // namespace tpiv
