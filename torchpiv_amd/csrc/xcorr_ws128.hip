// 128x128 interrogation windows.  Pass 1 (no window shift) runs the second-generation kernel of
// xcorr_big.hpp (two threads per line, 64-point codelets, planar LDS transposes); shifted passes at
// this size (only reachable with a 256-pixel first pass) still run the first-generation kernel
// (lane = column, whole tile in LDS, xcorr_kernel.hpp).
#include "xcorr_big.hpp"
#include "xcorr_kernel.hpp"
namespace tpiv {
hipError_t launch_xcorr_ws128(const PassParams& p, int mode, int n_cu, hipStream_t stream) {
    return launch_xcorr_ws<128>(p, mode, n_cu, stream);
}
hipError_t launch_xcorr_big128_pass1(const PassParams& p, int n_cu, hipStream_t stream) {
    return launch_xcorr_big128(p, n_cu, stream);
}
}  // namespace tpiv
