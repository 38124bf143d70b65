// Image ingest on the device: uncompressed BMP files, as PIV cameras write them, are
// uploaded as raw file bytes and unpacked here -- header skip, bottom-up row flip, 4-byte row padding
// strip, palette look-up (8 bit) or OpenCV's fixed-point BGR -> gray (24 / 32 bit).
//
// Reference: PIVDataset.__getitem__ (PIVbackend.py:129-144) decodes on the host with
// cv2.imdecode(np.fromfile(path), IMREAD_GRAYSCALE) and uploads the frame; at > 10^3 pairs/s that host
// decode is the bottleneck, while on the device it is a 4 MB -> 4 MB copy per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "piv_kernels.h"

namespace tpiv {

namespace {

// desc[f] = {file offset in raw, pixel-data offset in the file, row stride, bytes per pixel, flip, unused}
__global__ __launch_bounds__(256) void bmp_unpack_kernel(const uint8_t* __restrict__ raw, const long long* __restrict__ desc,
                                                         const uint8_t* __restrict__ lut, int H, int W,
                                                         uint8_t* __restrict__ out) {
    const int f = blockIdx.z;
    const long long* d = desc + (size_t)f * 6;
    const long long base = d[0] + d[1];
    const long long stride = d[2];
    const int bpp = (int)d[3];
    const bool flip = d[4] != 0;
    const int y = blockIdx.y;
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;        // four output pixels per thread
    if (x4 >= W) return;
    const uint8_t* __restrict__ src = raw + base + (long long)(flip ? H - 1 - y : y) * stride;
    const uint8_t* __restrict__ tab = lut + (size_t)f * 256;
    uint8_t* __restrict__ dst = out + ((size_t)f * H + y) * W + x4;
    uint8_t px[4];
    const int n = W - x4 < 4 ? W - x4 : 4;
    if (bpp == 1) {
        uint32_t w;
        if (n == 4) __builtin_memcpy(&w, src + x4, 4);                 // (unaligned dword loads are fine in global memory)
        else {
            w = 0;
            for (int k = 0; k < n; ++k) w |= (uint32_t)src[x4 + k] << (8 * k);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) px[k] = tab[(w >> (8 * k)) & 0xffu];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < n) {
                const uint8_t* q = src + (long long)(x4 + k) * bpp;
                // OpenCV cvtColor BGR2GRAY, 8 bit: (B*1868 + G*9617 + R*4899 + 8192) >> 14
                px[k] = (uint8_t)(((int)q[0] * 1868 + (int)q[1] * 9617 + (int)q[2] * 4899 + 8192) >> 14);
            } else px[k] = 0;
        }
    }
    if (n == 4 && (W & 3) == 0) {
        uint32_t w;
        __builtin_memcpy(&w, px, 4);
        *reinterpret_cast<uint32_t*>(dst) = w;
    } else {
        for (int k = 0; k < n; ++k) dst[k] = px[k];
    }
}

}  // namespace

hipError_t launch_bmp_unpack(const uint8_t* raw, const long long* desc, const uint8_t* lut, int n_files, int H, int W,
                             uint8_t* out, hipStream_t stream) {
    if (n_files <= 0) return hipSuccess;
    const int tx = 256;
    dim3 grid(((W + 3) / 4 + tx - 1) / tx, H, n_files);
    if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bmp_unpack_kernel, grid, dim3(tx), 0, stream, raw, desc, lut, H, W, out);
    return hipGetLastError();
}

}  // namespace tpiv
