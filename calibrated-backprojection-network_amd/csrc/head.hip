// head.hip -- output head for gfx950: MultiScaleDecoder.output0 (3x3, linear; reference
// src/networks.py:1842-1851, 1985) fused with KBNetModel.forward's sigmoid and depth
// mapping d_min / (sigmoid + d_min/d_max) (reference src/kbnet_model.py:181-184).
// HBM-bound: `channels` planes in, one plane out.  16 x 32 output tile per workgroup,
// the input tile (+1 halo, zero padded) staged in LDS, weights read through the scalar cache.
#include <math.h>

#include "kbn_common.h"

namespace kbn {

constexpr int HD_TW = 32, HD_TH = 16, HD_FW = HD_TW + 2, HD_FH = HD_TH + 2, HD_MAXC = 16;

__global__ __launch_bounds__(256) void depth_head_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ depth, float* __restrict__ logits,
                                                         int C, int H, int W, int tilesX, int tilesY, float dmin,
                                                         float ratio) {
    __shared__ float tile[HD_MAXC * HD_FH * HD_FW];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % tilesX;
    bid /= tilesX;
    const int ty = bid % tilesY;
    const int n = bid / tilesY;
    const int oy0 = ty * HD_TH, ox0 = tx * HD_TW;
    const long long HW = (long long)H * W;
    const float* xn = x + (long long)n * C * HW;
    for (int e = tid; e < HD_FH * HD_FW; e += 256) {
        int r = e / HD_FW, c = e - r * HD_FW;
        int Y = oy0 - 1 + r, X = ox0 - 1 + c;
        bool inb = (Y >= 0 && Y < H && X >= 0 && X < W);
        long long off = (long long)Y * W + X;
        for (int ch = 0; ch < C; ++ch) tile[ch * (HD_FH * HD_FW) + e] = inb ? xn[ch * HW + off] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < (HD_TW * HD_TH) / 256; ++u) {
        int e = tid + u * 256;
        int oy = e / HD_TW, ox = e - oy * HD_TW;
        float acc = 0.f;
        for (int ch = 0; ch < C; ++ch) {
            const float* t = tile + ch * (HD_FH * HD_FW) + oy * HD_FW + ox;
            const float* wc = w + ch * 9;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc = fmaf(wc[ky * 3 + kx], t[ky * HD_FW + kx], acc);
        }
        int Y = oy0 + oy, X = ox0 + ox;
        if (Y < H && X < W) {
            long long o = (long long)n * HW + (long long)Y * W + X;
            if (logits) logits[o] = acc;
            float s = 1.0f / (1.0f + expf(-acc));
            depth[o] = dmin / (s + ratio);
        }
    }
}

}  // namespace kbn

extern "C" int kbn_depth_head_forward(const float* x, const float* weight, float* depth, float* logits, int n,
                                      int channels, int height, int width, float min_predict_depth,
                                      float max_predict_depth, kbn_stream_t stream) {
    using namespace kbn;
    if (!x || !weight || !depth || n < 1 || channels < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (channels > HD_MAXC) return KBN_ERR_UNSUPPORTED;
    int tilesX = ceil_div(width, HD_TW), tilesY = ceil_div(height, HD_TH);
    long long blocks = (long long)tilesX * tilesY * n;
    if (blocks > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    // the reference evaluates d_min / d_max in double and adds it as an fp32 scalar
    float ratio = (float)((double)min_predict_depth / (double)max_predict_depth);
    hipLaunchKernelGGL(depth_head_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, weight,
                       depth, logits, channels, height, width, tilesX, tilesY, min_predict_depth, ratio);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}
