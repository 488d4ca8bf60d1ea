// s2d.hip -- fused sparse-to-dense pool (S2D) for gfx950.
//
// Replaces networks.SparseToDensePool.forward (reference src/networks.py:2168-2196):
//   z = x[:, 0]; min-pools over the non-zero depths (999-sentinel semantics, :2175-2181),
//   max-pools (:2183-2186), cat, n_convolution x [conv1x1 + LeakyReLU], cat with x,
//   conv3x3 + LeakyReLU -- one launch, one read of x, one write of the n_filter maps.
//
// One workgroup (256 threads) produces a 16 x 32 output tile.  The sparse depth tile with
// a halo of R+1 (R = largest pool radius) is staged in LDS twice: `zmin` with zeros
// replaced by the 999 sentinel (+inf outside the image) and `zmax` (-inf outside).  Every
// pool is evaluated separably -- a row pass into `hbuf`, then a column pass straight into
// registers -- on the (16+2) x (32+2) "feature" region the 3x3 conv needs.  The 1x1 conv
// chain runs in registers, its outputs (plus the raw x channels) go to LDS as the 3x3
// conv's input tile, zero outside the image exactly like the reference's zero padding.
#include <math.h>

#include "kbn_common.h"

namespace kbn {

constexpr int S2D_TW = 32, S2D_TH = 16;
constexpr int S2D_FW = S2D_TW + 2, S2D_FH = S2D_TH + 2;
constexpr int S2D_NPOS = (S2D_FW * S2D_FH + 255) / 256;  // feature positions per thread (3)
constexpr int S2D_MAXPOOL = 8, S2D_MAXF = 8, S2D_MAXCONV = 4, S2D_MAXIN = 2;

struct S2DParams {
    const float* x;
    long long x_bstride;
    float* out;
    float* pyramid;
    const float* wpool[S2D_MAXCONV];
    const float* wconv;
    int N, H, W, inC;
    int nmin, npool;
    int ksize[S2D_MAXPOOL];
    int nconv, nf, R;
    int tilesX, tilesY;
    float slope;
};

__global__ __launch_bounds__(256) void s2d_kernel(const S2DParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int R = p.R;
    const int ZW = S2D_FW + 2 * R, ZH = S2D_FH + 2 * R;
    float* zmin = smem;
    float* zmax = zmin + ZH * ZW;
    float* hbuf = zmax + ZH * ZW;
    float* feat = smem;  // [(nf + inC)][FH][FW]; overlays the pool buffers, which are dead by then

    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int oy0 = ty * S2D_TH, ox0 = tx * S2D_TW;
    const float* xz = p.x + (long long)n * p.x_bstride;  // channel 0 = sparse depth

    // ---- stage the depth tile (+halo) -------------------------------------------------
    for (int e = tid; e < ZH * ZW; e += 256) {
        int r = e / ZW, c = e - r * ZW;
        int Y = oy0 - 1 - R + r, X = ox0 - 1 - R + c;
        float vmin = INFINITY, vmax = -INFINITY;
        if (Y >= 0 && Y < p.H && X >= 0 && X < p.W) {
            float v = xz[(long long)Y * p.W + X];
            vmax = v;
            vmin = (v == 0.f) ? 999.f : v;  // where(z == 0, -999, -z) in negated form
        }
        zmin[e] = vmin;
        zmax[e] = vmax;
    }
    __syncthreads();

    // ---- pools: separable min / max, results for this thread's feature positions ------
    float pooled[S2D_NPOS][S2D_MAXPOOL];
#pragma unroll
    for (int pi = 0; pi < S2D_MAXPOOL; ++pi) {
        if (pi >= p.npool) break;
        const bool is_min = pi < p.nmin;
        const int rad = p.ksize[pi] >> 1;
        const float* src = is_min ? zmin : zmax;
        // row pass over the rows the column pass will touch: [R-rad, R+FH+rad)
        const int rows = S2D_FH + 2 * rad;
        for (int e = tid; e < rows * S2D_FW; e += 256) {
            int rr = e / S2D_FW, c = e - rr * S2D_FW;
            int r = rr + R - rad;
            const float* s = src + r * ZW + c + R;
            float a = s[0];
            if (is_min) {
                for (int d = 1; d <= rad; ++d) a = fminf(a, fminf(s[-d], s[d]));
            } else {
                for (int d = 1; d <= rad; ++d) a = fmaxf(a, fmaxf(s[-d], s[d]));
            }
            hbuf[r * S2D_FW + c] = a;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < S2D_NPOS; ++u) {
            int e = tid + u * 256;
            float a = 0.f;
            if (e < S2D_FH * S2D_FW) {
                int fy = e / S2D_FW, fx = e - fy * S2D_FW;
                const float* s = hbuf + (fy + R) * S2D_FW + fx;
                a = s[0];
                if (is_min) {
                    for (int d = 1; d <= rad; ++d) a = fminf(a, fminf(s[-d * S2D_FW], s[d * S2D_FW]));
                    a = (a == 999.f) ? 0.f : a;  // where(pool == 999, 0, pool)
                } else {
                    for (int d = 1; d <= rad; ++d) a = fmaxf(a, fmaxf(s[-d * S2D_FW], s[d * S2D_FW]));
                }
            }
            pooled[u][pi] = a;
        }
        __syncthreads();
    }

    const int nch = p.nf + p.inC;
    // ---- 1x1 conv chain in registers; features + raw x channels to LDS ----------------
#pragma unroll
    for (int u = 0; u < S2D_NPOS; ++u) {
        int e = tid + u * 256;
        if (e >= S2D_FH * S2D_FW) continue;
        int fy = e / S2D_FW, fx = e - fy * S2D_FW;
        int Y = oy0 - 1 + fy, X = ox0 - 1 + fx;
        const bool inb = (Y >= 0 && Y < p.H && X >= 0 && X < p.W);
        if (p.pyramid) {
            if (inb && fy >= 1 && fy <= S2D_TH && fx >= 1 && fx <= S2D_TW) {
                float* py = p.pyramid + ((long long)n * p.npool) * p.H * p.W + (long long)Y * p.W + X;
#pragma unroll
                for (int pi = 0; pi < S2D_MAXPOOL; ++pi)
                    if (pi < p.npool) py[(long long)pi * p.H * p.W] = pooled[u][pi];
            }
            continue;
        }
        float h[S2D_MAXF], g[S2D_MAXF];
#pragma unroll
        for (int f = 0; f < S2D_MAXF; ++f) {
            float a = 0.f;
            if (f < p.nf) {
#pragma unroll
                for (int pi = 0; pi < S2D_MAXPOOL; ++pi)
                    if (pi < p.npool) a = fmaf(p.wpool[0][f * p.npool + pi], pooled[u][pi], a);
                a = leaky_relu(a, p.slope);
            }
            h[f] = a;
        }
#pragma unroll
        for (int i = 1; i < S2D_MAXCONV; ++i) {
            if (i >= p.nconv) break;
            const float* w = p.wpool[i];
#pragma unroll
            for (int f = 0; f < S2D_MAXF; ++f) {
                float a = 0.f;
                if (f < p.nf) {
#pragma unroll
                    for (int q = 0; q < S2D_MAXF; ++q)
                        if (q < p.nf) a = fmaf(w[f * p.nf + q], h[q], a);
                    a = leaky_relu(a, p.slope);
                }
                g[f] = a;
            }
#pragma unroll
            for (int f = 0; f < S2D_MAXF; ++f) h[f] = g[f];
        }
#pragma unroll
        for (int f = 0; f < S2D_MAXF; ++f)
            if (f < p.nf) feat[f * (S2D_FH * S2D_FW) + e] = inb ? h[f] : 0.f;
#pragma unroll
        for (int ci = 0; ci < S2D_MAXIN; ++ci)
            if (ci < p.inC)
                feat[(p.nf + ci) * (S2D_FH * S2D_FW) + e] =
                    inb ? xz[(long long)ci * p.H * p.W + (long long)Y * p.W + X] : 0.f;
    }
    if (p.pyramid) return;
    __syncthreads();

    // ---- 3x3 conv over [features | x] + LeakyReLU --------------------------------------
#pragma unroll
    for (int u = 0; u < (S2D_TW * S2D_TH) / 256; ++u) {
        int e = tid + u * 256;
        int oy = e / S2D_TW, ox = e - oy * S2D_TW;
        float acc[S2D_MAXF];
#pragma unroll
        for (int f = 0; f < S2D_MAXF; ++f) acc[f] = 0.f;
        for (int ch = 0; ch < nch; ++ch) {
            const float* fb = feat + ch * (S2D_FH * S2D_FW) + oy * S2D_FW + ox;
            float v[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) v[ky * 3 + kx] = fb[ky * S2D_FW + kx];
#pragma unroll
            for (int f = 0; f < S2D_MAXF; ++f) {
                if (f < p.nf) {
                    const float* w = p.wconv + ((long long)f * nch + ch) * 9;
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc[f] = fmaf(w[t], v[t], acc[f]);
                }
            }
        }
        int Y = oy0 + oy, X = ox0 + ox;
        if (Y < p.H && X < p.W) {
            float* o = p.out + ((long long)n * p.nf) * p.H * p.W + (long long)Y * p.W + X;
#pragma unroll
            for (int f = 0; f < S2D_MAXF; ++f)
                if (f < p.nf) o[(long long)f * p.H * p.W] = leaky_relu(acc[f], p.slope);
        }
    }
}

static int s2d_launch(S2DParams& p, const int* min_pool_sizes, int n_min, const int* max_pool_sizes,
                      int n_max, hipStream_t stream) {
    if (n_min < 0 || n_max < 0 || n_min + n_max < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (n_min + n_max > S2D_MAXPOOL) return KBN_ERR_UNSUPPORTED;
    if ((n_min && !min_pool_sizes) || (n_max && !max_pool_sizes)) return KBN_ERR_INVALID_ARGUMENT;
    int R = 0;
    for (int i = 0; i < n_min + n_max; ++i) {
        int k = i < n_min ? min_pool_sizes[i] : max_pool_sizes[i - n_min];
        if (k < 3 || (k & 1) == 0) return KBN_ERR_INVALID_ARGUMENT;  // the caller drops sizes <= 1
        if (k > 31) return KBN_ERR_UNSUPPORTED;
        p.ksize[i] = k;
        if (k / 2 > R) R = k / 2;
    }
    for (int i = n_min + n_max; i < S2D_MAXPOOL; ++i) p.ksize[i] = 1;
    p.nmin = n_min;
    p.npool = n_min + n_max;
    p.R = R;
    p.tilesX = ceil_div(p.W, S2D_TW);
    p.tilesY = ceil_div(p.H, S2D_TH);
    const int ZW = S2D_FW + 2 * R, ZH = S2D_FH + 2 * R;
    size_t lds_pool = sizeof(float) * ((size_t)2 * ZH * ZW + (size_t)ZH * S2D_FW);
    size_t lds_feat = sizeof(float) * (size_t)(S2D_MAXF + S2D_MAXIN) * S2D_FH * S2D_FW;
    size_t lds = lds_pool > lds_feat ? lds_pool : lds_feat;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(s2d_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return KBN_ERR_LAUNCH;
        attr_set = true;
    }
    long long blocks = (long long)p.tilesX * p.tilesY * p.N;
    if (blocks > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(s2d_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, p);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

}  // namespace kbn

extern "C" {

int kbn_s2d_forward(const float* x, const float* const* w_pool_convs, const float* w_conv, float* out,
                    int n, int height, int width, int input_channels, const int* min_pool_sizes,
                    int n_min, const int* max_pool_sizes, int n_max, int n_convolution, int n_filter,
                    float negative_slope, kbn_stream_t stream) {
    using namespace kbn;
    if (!x || !w_pool_convs || !w_conv || !out) return KBN_ERR_INVALID_ARGUMENT;
    if (n < 1 || height < 1 || width < 1 || input_channels < 1 || n_convolution < 1 || n_filter < 1)
        return KBN_ERR_INVALID_ARGUMENT;
    if (n_filter > S2D_MAXF || input_channels > S2D_MAXIN || n_convolution > S2D_MAXCONV)
        return KBN_ERR_UNSUPPORTED;
    S2DParams p{};
    p.x = x;
    p.x_bstride = (long long)input_channels * height * width;
    p.out = out;
    p.pyramid = nullptr;
    for (int i = 0; i < S2D_MAXCONV; ++i) {
        p.wpool[i] = i < n_convolution ? w_pool_convs[i] : w_pool_convs[0];
        if (i < n_convolution && !w_pool_convs[i]) return KBN_ERR_INVALID_ARGUMENT;
    }
    p.wconv = w_conv;
    p.N = n; p.H = height; p.W = width; p.inC = input_channels;
    p.nconv = n_convolution; p.nf = n_filter; p.slope = negative_slope;
    return s2d_launch(p, min_pool_sizes, n_min, max_pool_sizes, n_max, (hipStream_t)stream);
}

int kbn_s2d_pyramid(const float* x_depth, long long batch_stride, float* pyramid, int n, int height,
                    int width, const int* min_pool_sizes, int n_min, const int* max_pool_sizes, int n_max,
                    kbn_stream_t stream) {
    using namespace kbn;
    if (!x_depth || !pyramid || n < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    S2DParams p{};
    p.x = x_depth;
    p.x_bstride = batch_stride;
    p.out = nullptr;
    p.pyramid = pyramid;
    p.N = n; p.H = height; p.W = width; p.inC = 1;
    p.nconv = 1; p.nf = 1; p.slope = 0.f;
    return s2d_launch(p, min_pool_sizes, n_min, max_pool_sizes, n_max, (hipStream_t)stream);
}

}  // extern "C"
