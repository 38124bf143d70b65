// Tile kernels for 128x128 interrogation windows (see xcorr_kernel.hpp).
#include "xcorr_kernel.hpp"
namespace tpiv {
template hipError_t launch_xcorr_ws<128>(const PassParams&, int, int, hipStream_t);
}
