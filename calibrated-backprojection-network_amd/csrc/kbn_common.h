// Shared device/host helpers for the KBNet gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kbnet_hip.h"

#define KBN_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return KBN_ERR_LAUNCH;        \
    } while (0)

namespace kbn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

__device__ __forceinline__ float leaky_relu(float v, float slope) { return v > 0.f ? v : v * slope; }

// XCD-aware block remap (bijective for any grid size): the dispatcher places block b on
// XCD b % 8; give every XCD one contiguous range of logical tiles so that neighbouring
// tiles (which share input halos / weight panels) hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int NX = 8;
    int xcd = bid % NX, idx = bid / NX;
    int q = nblocks / NX, r = nblocks % NX;
    int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// PyTorch's nearest-neighbour source index (aten/src/ATen/native/UpSample.h
// nearest_idx): identity when sizes match, >>1 for an exact 2x, else
// min(floor(dst * (float)in/out), in-1).
__device__ __forceinline__ int nearest_src_index(int dst, int in_size, int out_size) {
    if (in_size == out_size) return dst;
    if (out_size == 2 * in_size) return dst >> 1;
    float scale = (float)in_size / (float)out_size;
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

// conv_igemm.hip
int conv2d_launch(const kbn_conv_src* srcs, int n_src, const float* packed_weight, float* out,
                  long long out_batch_stride, int n, int out_channels, int kernel_size, int stride,
                  int in_height, int in_width, int resize, int apply_activation, float negative_slope,
                  hipStream_t stream);

}  // namespace kbn
