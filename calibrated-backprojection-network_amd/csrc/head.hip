// head.hip -- output head for gfx950: MultiScaleDecoder.output0 (3x3, linear; reference
// src/networks.py:1842-1851, 1985) fused with KBNetModel.forward's sigmoid and depth
// mapping d_min / (sigmoid + d_min/d_max) (reference src/kbnet_model.py:181-184).
// HBM-bound: `channels` planes in, one plane out.
//
// Fast kernel (W % 4 == 0, 16-byte aligned planes): 16 x 64 output tile per workgroup.  The input tile
// (+1 halo, columns from x0-4 so that every 16-byte granule is 4 in-image pixels) reaches LDS by LDS-DMA
// into a pre-zeroed buffer (out-of-image granules are never written = zero padding); each thread then
// produces 4 consecutive pixels of a row from a 3 x 6 window per channel (two LDS reads per row instead of
// nine per pixel), weights through the scalar cache, one 16-byte store.  The general kernel (any shape)
// stages through registers and computes one pixel at a time.
#include <math.h>
#include <stdlib.h>

#include "conv_common.h"

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

// ---- fast kernel -------------------------------------------------------------------------------
constexpr int HQ_TW = 64, HQ_TH = 16;
constexpr int HQ_COLS = HQ_TW + 8, HQ_ROWS = HQ_TH + 2;            // staged columns x0-4 .. x0+67, rows y0-1 .. y0+16
constexpr int HQ_PLANE = HQ_ROWS * HQ_COLS;                        // 1296 floats per channel
constexpr int HQ_NF4 = HQ_PLANE / 4;                               // 324 granules per channel

__global__ __launch_bounds__(256) void depth_head_dma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ depth, float* __restrict__ logits,
                                                             int C, int H, int W, int tilesX, int tilesY, float dmin,
                                                             float ratio) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [C][HQ_ROWS][HQ_COLS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int tx = bid % tilesX;
    bid /= tilesX;
    const int ty = bid % tilesY;
    const int n = bid / tilesY;
    const int oy0 = ty * HQ_TH, ox0 = tx * HQ_TW;
    const long long HW = (long long)H * W;

    {   // zero padding: clear the tile, then let the DMAs overwrite the in-image granules
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int e = tid * 4; e < C * HQ_PLANE; e += 1024) *reinterpret_cast<f32x4*>(tile + e) = zero;
    }
    __syncthreads();
    // wave w stages channels w, w+4, ...: 324 granules = 6 DMA instructions per channel
    unsigned gv[6];
    unsigned long long gm[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int f = j * 64 + lane;
        int g = -1;
        if (f < HQ_NF4) {
            const int r = f / (HQ_COLS / 4), cv = f - r * (HQ_COLS / 4);
            const int Y = oy0 - 1 + r, X = ox0 - 4 + cv * 4;
            if (Y >= 0 && Y < H && X >= 0 && X < W) g = (Y * W + X) * 4;
        }
        gv[j] = g < 0 ? 0u : (unsigned)g;
        gm[j] = __ballot(g >= 0);
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(tile));
    for (int ch = wave; ch < C; ch += 4) {
        const float* src = uniform_ptr(x + ((long long)n * C + ch) * HW);
        const unsigned dst = lds0 + 4u * (unsigned)(ch * HQ_PLANE);
#pragma unroll
        for (int j = 0; j < 6; ++j) lds_dma16_sm(src, gv[j], dst + j * 1024, gm[j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // thread -> 4 consecutive pixels: row oy = tid / 16, columns 4 * (tid % 16) ..
    const int oy = tid >> 4, oxq = (tid & 15) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < C; ++ch) {
        const float* t = tile + ch * HQ_PLANE + oy * HQ_COLS + oxq;   // staged column oxq <-> X = ox0 + oxq - 4
        const float* wc = w + ch * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(t + ky * HQ_COLS);       // X-4 .. X-1
            const f32x4 b = *reinterpret_cast<const f32x4*>(t + ky * HQ_COLS + 4);   // X   .. X+3
            const float c = t[ky * HQ_COLS + 8];                                      // X+4
            const float v[6] = {a[3], b[0], b[1], b[2], b[3], c};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float wk = wc[ky * 3 + kx];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = fmaf(wk, v[i + kx], acc[i]);
            }
        }
    }
    const int Y = oy0 + oy, X = ox0 + oxq;
    if (Y < H && X < W) {   // W % 4 == 0: a quad is inside or outside as a whole
        const long long o = (long long)n * HW + (long long)Y * W + X;
        f32x4 d;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = 1.0f / (1.0f + expf(-acc[i]));
            d[i] = dmin / (s + ratio);
        }
        if (logits) *reinterpret_cast<f32x4*>(logits + o) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(depth + o) = d;
    }
}

}  // namespace kbn

extern "C" int kbn_depth_head_forward(const float* x, const float* weight, float* depth, float* logits, int n,
                                      int channels, int height, int width, float min_predict_depth,
                                      float max_predict_depth, kbn_stream_t stream) {
    using namespace kbn;
    if (!x || !weight || !depth || n < 1 || channels < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (channels > HD_MAXC) return KBN_ERR_UNSUPPORTED;
    // the reference evaluates d_min / d_max in double and adds it as an fp32 scalar
    float ratio = (float)((double)min_predict_depth / (double)max_predict_depth);
    const bool aligned = (width & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(depth) & 15) == 0 && (!logits || (reinterpret_cast<uintptr_t>(logits) & 15) == 0);
    if (aligned && !kbn::knob(kbn::KNOB_NO_HEAD_DMA)) {
        const int tilesX = ceil_div(width, HQ_TW), tilesY = ceil_div(height, HQ_TH);
        const long long blocks = (long long)tilesX * tilesY * n;
        if (blocks > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
        const size_t lds = sizeof(float) * (size_t)channels * HQ_PLANE;
        static kbn::DeviceOnce once;
        if (int rc = kbn::set_max_dynamic_lds(once, reinterpret_cast<const void*>(depth_head_dma_kernel), 160 * 1024))
            return rc;
        hipLaunchKernelGGL(depth_head_dma_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, weight,
                           depth, logits, channels, height, width, tilesX, tilesY, min_predict_depth, ratio);
        KBN_CHECK_LAUNCH();
        return KBN_OK;
    }
    int tilesX = ceil_div(width, HD_TW), tilesY = ceil_div(height, HD_TH);
    long long blocks = (long long)tilesX * tilesY * n;
    if (blocks > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(depth_head_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, weight,
                       depth, logits, channels, height, width, tilesX, tilesY, min_predict_depth, ratio);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}
