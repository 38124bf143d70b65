// The shifted-window (CWS) instances of the 32x32 tile kernel (see xcorr_tile.hpp): a translation unit of their own because
// they are compiled with the backend's max-ILP scheduling strategy (Makefile, ILP_UNITS) -- same instructions in another order,
// same bits, 1 % faster here and 5 % SLOWER for the first-pass / candidate instances of xcorr_ws32.hip.
#include "xcorr_tile.hpp"
namespace tpiv {
hipError_t launch_xcorr_ws32_cws(const PassParams& p, int n_cu, hipStream_t stream) {
    return launch_tile<32, MODE_CWS>(p, n_cu, stream);
}
}  // namespace tpiv
