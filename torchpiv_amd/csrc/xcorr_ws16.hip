// Tile kernels for 16x16 interrogation windows (see xcorr_kernel.hpp).
#include "xcorr_kernel.hpp"
namespace tpiv {
template hipError_t launch_xcorr_ws<16>(const PassParams&, int, int, hipStream_t);
}
