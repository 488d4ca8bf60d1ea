// unpack.hip -- device side of the input pipeline: decoded PNG pixels -> the
// tensors KBNetInferenceDataset.__getitem__ returns (reference src/datasets.py:259-283):
//   image        uint8 H x Wraw x C (C = 1 gray, 3 RGB, 4 RGBA), columns [x_offset, x_offset + W) of each
//                row -- the middle third of an image triplet (load_image_triplet, :22-46) or the whole
//                image -- as float32 N x 3 x H x W, values 0..255 (normalize=False; gray is replicated
//                and alpha dropped, which is what PIL's convert('RGB') does, src/data_utils.py:75)
//   sparse depth uint16 (or uint8) H x W -> float32 N x 1 x H x W = value / 256
//                (data_utils.load_depth, src/data_utils.py:137-141; `z[z <= 0] = 0` is a no-op on unsigned)
// Byte work, HBM bound: 5 bytes in, 16 bytes out per pixel; one thread per 4 pixels of a row.
#include "kbn_common.h"

namespace kbn {

template <int C, typename DepthT>
__global__ __launch_bounds__(256) void unpack_frames_kernel(const unsigned char* __restrict__ img,
                                                            const DepthT* __restrict__ dep, float* __restrict__ image,
                                                            float* __restrict__ depth, int H, int W, int Wraw,
                                                            int x_offset, long long quads) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= quads) return;
    const int wq = (W + 3) >> 2;
    const int xq = (int)(q % wq);
    long long r = q / wq;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    const int x0 = xq * 4;
    const long long HW = (long long)H * W;
    const unsigned char* src = img + (((long long)n * H + y) * Wraw + x_offset + x0) * C;
    float* oi = image ? image + (long long)n * 3 * HW + (long long)y * W + x0 : nullptr;
    float* od = depth ? depth + (long long)n * HW + (long long)y * W + x0 : nullptr;
    const DepthT* sd = dep ? dep + ((long long)n * H + y) * W + x0 : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (x0 + i >= W) break;
        if (oi) {
            const float c0 = (float)src[i * C];
            const float c1 = C >= 3 ? (float)src[i * C + 1] : c0;
            const float c2 = C >= 3 ? (float)src[i * C + 2] : c0;
            oi[i] = c0; oi[HW + i] = c1; oi[2 * HW + i] = c2;
        }
        if (od) od[i] = (float)sd[i] / 256.0f;
    }
}

template <typename DepthT>
static int launch_unpack(const unsigned char* img, const void* dep, float* image, float* depth, int n, int H, int W,
                         int Wraw, int xoff, int C, hipStream_t st) {
    const long long quads = (long long)n * H * ((W + 3) >> 2);
    if (quads > 0x7fffffffLL * 256LL) return KBN_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((quads + 255) / 256)), block(256);
    const DepthT* d = static_cast<const DepthT*>(dep);
    switch (C) {
        case 1: hipLaunchKernelGGL((unpack_frames_kernel<1, DepthT>), grid, block, 0, st, img, d, image, depth, H, W, Wraw, xoff, quads); break;
        case 3: hipLaunchKernelGGL((unpack_frames_kernel<3, DepthT>), grid, block, 0, st, img, d, image, depth, H, W, Wraw, xoff, quads); break;
        case 4: hipLaunchKernelGGL((unpack_frames_kernel<4, DepthT>), grid, block, 0, st, img, d, image, depth, H, W, Wraw, xoff, quads); break;
        default: return KBN_ERR_UNSUPPORTED;
    }
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

}  // namespace kbn

extern "C" int kbn_unpack_frames_forward(const unsigned char* image_u8, const void* depth_raw, float* image,
                                         float* sparse_depth, int n, int height, int width, int raw_width,
                                         int x_offset, int image_channels, int depth_bits, kbn_stream_t stream) {
    using namespace kbn;
    if (n < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if ((image != nullptr) != (image_u8 != nullptr) || (sparse_depth != nullptr) != (depth_raw != nullptr))
        return KBN_ERR_INVALID_ARGUMENT;
    if (!image && !sparse_depth) return KBN_ERR_INVALID_ARGUMENT;
    if (image && (raw_width < width || x_offset < 0 || x_offset + width > raw_width)) return KBN_ERR_INVALID_ARGUMENT;
    if (!image) image_channels = 3;
    if (depth_bits == 16)
        return launch_unpack<unsigned short>(image_u8, depth_raw, image, sparse_depth, n, height, width, raw_width,
                                             x_offset, image_channels, (hipStream_t)stream);
    if (depth_bits == 8)
        return launch_unpack<unsigned char>(image_u8, depth_raw, image, sparse_depth, n, height, width, raw_width,
                                            x_offset, image_channels, (hipStream_t)stream);
    return KBN_ERR_UNSUPPORTED;
}

// The other direction, for data_utils.save_depth (reference src/data_utils.py:154-167): np.uint32(z * 256.0) and PIL's clip of a
// mode 'I' image to the 16 bits of the PNG it writes -- samples = min(trunc(z * 256), 65535); z * 256 is exact in fp32.  Negative or NaN
// depths (numpy leaves their conversion undefined) give 0.  One read, one 2-byte write per pixel.
namespace kbn {
__global__ __launch_bounds__(256) void depth_to_u16_kernel(const float* __restrict__ depth, unsigned short* __restrict__ out, long long total) {
    const long long step = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
        const float v = depth[i] * 256.0f;
        out[i] = (unsigned short)(v >= 65535.0f ? 65535u : (v >= 0.f ? (unsigned)v : 0u));
    }
}
}  // namespace kbn

extern "C" int kbn_depth_to_u16_forward(const float* depth, unsigned short* samples, long long count, kbn_stream_t stream) {
    if (!depth || !samples || count < 1) return KBN_ERR_INVALID_ARGUMENT;
    const int blocks = (int)std::min<long long>(4096, (count + 255) / 256);
    hipLaunchKernelGGL(kbn::depth_to_u16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, depth, samples, count);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}
