// pre_eval.hip -- the stages the reference's run loop executes around the model inside its timed
// region / right after it, as small HBM-bound kernels.
//
//  kbn_preprocess_forward : validity map                       reference src/kbnet.py:899-902
//                           OutlierRemoval.remove_outliers      reference src/net_utils.py:1761-1806
//                           image / 255                         reference src/transforms.py:201-204
//  kbn_eval_accumulate    : GT-masked error sums for MAE / RMSE / iMAE / iRMSE
//                                                               reference src/kbnet.py:932-950,
//                                                               src/eval_utils.py:20-78
#include <math.h>

#include "kbn_common.h"

namespace kbn {

// monotone float -> uint map so that atomicMax orders like the floats do
__device__ __forceinline__ unsigned ordered_bits(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// batch-global max of the sparse depth (torch.max(sparse_depth), src/net_utils.py:1776)
__global__ __launch_bounds__(256) void max_reduce_kernel(const float* __restrict__ x, long long total,
                                                         unsigned* __restrict__ result) {
    float m = -INFINITY;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        atomicMax(result, ordered_bits(m));
    }
}

constexpr int PRE_TW = 64, PRE_TH = 16, PRE_MAXR = 7;

__global__ __launch_bounds__(256) void preprocess_kernel(const float* __restrict__ image, const float* __restrict__ sparse,
                                                         float* __restrict__ out_image, float* __restrict__ out_validity,
                                                         float* __restrict__ out_sparse, const unsigned* __restrict__ maxbits,
                                                         int C, int H, int W, int tilesX, int tilesY, int radius,
                                                         float threshold, int signed_range) {
    __shared__ float tile[(PRE_TH + 2 * PRE_MAXR) * (PRE_TW + 2 * PRE_MAXR)];
    __shared__ float hmin[(PRE_TH + 2 * PRE_MAXR) * PRE_TW];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int n = bid / tilesY;
    const int y0 = ty * PRE_TH, x0 = tx * PRE_TW;
    const long long HW = (long long)H * W;
    const float* sp = sparse + (long long)n * HW;
    const float max_value = 10.0f * from_ordered_bits(*maxbits);
    const int ZW = PRE_TW + 2 * radius, ZH = PRE_TH + 2 * radius;
    // depth with zeros (validity <= 0) and the padding replaced by max_value
    for (int e = tid; e < ZH * ZW; e += 256) {
        const int r = e / ZW, c = e - r * ZW;
        const int Y = y0 - radius + r, X = x0 - radius + c;
        float v = max_value;
        if (Y >= 0 && Y < H && X >= 0 && X < W) {
            const float s = sp[(long long)Y * W + X];
            const float valid = s > 0.f ? 1.f : s;
            v = (valid <= 0.f) ? max_value : s;
        }
        tile[e] = v;
    }
    __syncthreads();
    for (int e = tid; e < ZH * PRE_TW; e += 256) {  // row pass of the k x k min filter
        const int r = e / PRE_TW, c = e - r * PRE_TW;
        const float* s = tile + r * ZW + c + radius;
        float a = s[0];
        for (int d = 1; d <= radius; ++d) a = fminf(a, fminf(s[-d], s[d]));
        hmin[e] = a;
    }
    __syncthreads();
    for (int e = tid; e < PRE_TH * PRE_TW; e += 256) {
        const int r = e / PRE_TW, c = e - r * PRE_TW;
        const int Y = y0 + r, X = x0 + c;
        if (Y >= H || X >= W) continue;
        const float* s = hmin + (r + radius) * PRE_TW + c;
        float mn = s[0];
        for (int d = 1; d <= radius; ++d) mn = fminf(mn, fminf(s[-d * PRE_TW], s[d * PRE_TW]));
        const long long o = (long long)Y * W + X;
        const float sd = sp[o];
        const float valid = sd > 0.f ? 1.f : sd;
        const float keep = (mn < sd - threshold) ? 0.f : 1.f;
        const float vclean = valid * keep;
        out_validity[(long long)n * HW + o] = vclean;
        if (out_sparse) out_sparse[(long long)n * HW + o] = sd * vclean;
        if (image) {
            for (int ch = 0; ch < C; ++ch) {
                const long long io = ((long long)n * C + ch) * HW + o;
                const float t = image[io] / 255.0f;
                // [-1, 1]: 2.0 * (images / 255.0) - 1.0 (reference src/transforms.py:205-208); 2 t is exact, one rounding either way
                out_image[io] = signed_range ? __fsub_rn(2.0f * t, 1.0f) : t;
            }
        }
    }
}

// out[n*5 + {0..4}] += {sum |1000o-1000g|, sum (1000g-1000o)^2, sum |1/(.001g)-1/(.001o)|, sum (..)^2, count}
__global__ __launch_bounds__(256) void eval_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                   const float* __restrict__ gtv, double* __restrict__ out, int HW,
                                                   float dmin, float dmax) {
    const int n = blockIdx.y;
    const float* p = pred + (long long)n * HW;
    const float* g = gt + (long long)n * HW;
    const float* v = gtv + (long long)n * HW;
    double acc[5] = {0, 0, 0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        const float gg = g[i];
        if (v[i] > 0.f && gg > dmin && gg < dmax) {
            const float o = p[i];
            const float e = 1000.0f * o - 1000.0f * gg;          // fp32 like the numpy reference
            const float ie = 1.0f / (0.001f * gg) - 1.0f / (0.001f * o);
            acc[0] += fabsf(e);
            acc[1] += (double)(e * e);
            acc[2] += fabsf(ie);
            acc[3] += (double)(ie * ie);
            acc[4] += 1.0;
        }
    }
    __shared__ double red[4][5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double a = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = a;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int k = threadIdx.x;
        atomicAdd(out + (long long)n * 5 + k, red[0][k] + red[1][k] + red[2][k] + red[3][k]);
    }
}

}  // namespace kbn

extern "C" {

int kbn_preprocess_forward(const float* image, const float* sparse_depth, float* out_image, float* out_validity,
                           float* out_sparse_depth, void* workspace, size_t workspace_bytes, int n,
                           int image_channels, int height, int width, int kernel_size, float threshold,
                           int image_range, kbn_stream_t stream) {
    using namespace kbn;
    if (image_range != KBN_IMAGE_RANGE_0_1 && image_range != KBN_IMAGE_RANGE_M1_1) return KBN_ERR_INVALID_ARGUMENT;
    if (!sparse_depth || !out_validity || n < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if ((image == nullptr) != (out_image == nullptr) || (image && image_channels < 1)) return KBN_ERR_INVALID_ARGUMENT;
    if (kernel_size < 1 || (kernel_size & 1) == 0) return KBN_ERR_INVALID_ARGUMENT;
    if (kernel_size / 2 > PRE_MAXR) return KBN_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < sizeof(unsigned)) return KBN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned* maxbits = static_cast<unsigned*>(workspace);
    if (hipMemsetAsync(maxbits, 0, sizeof(unsigned), st) != hipSuccess) return KBN_ERR_LAUNCH;
    const long long total = (long long)n * height * width;
    int blocks = (int)((total + 256 * 8 - 1) / (256 * 8));
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(max_reduce_kernel, dim3(blocks), dim3(256), 0, st, sparse_depth, total, maxbits);
    const int tilesX = ceil_div(width, PRE_TW), tilesY = ceil_div(height, PRE_TH);
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((long long)tilesX * tilesY * n)), dim3(256), 0, st, image,
                       sparse_depth, out_image, out_validity, out_sparse_depth, maxbits, image_channels, height, width,
                       tilesX, tilesY, kernel_size / 2, threshold, image_range == KBN_IMAGE_RANGE_M1_1 ? 1 : 0);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_eval_accumulate(const float* output_depth, const float* ground_truth, const float* ground_truth_validity,
                        double* sums, int n, int height, int width, float min_evaluate_depth,
                        float max_evaluate_depth, kbn_stream_t stream) {
    using namespace kbn;
    if (!output_depth || !ground_truth || !ground_truth_validity || !sums || n < 1 || height < 1 || width < 1)
        return KBN_ERR_INVALID_ARGUMENT;
    const int HW = height * width;
    int bx = (HW + 256 * 16 - 1) / (256 * 16);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(eval_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, output_depth, ground_truth,
                       ground_truth_validity, sums, HW, min_evaluate_depth, max_evaluate_depth);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

}  // extern "C"
