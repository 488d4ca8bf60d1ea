// s2d.hip -- fused sparse-to-dense pool (S2D) for gfx950.
//
// Replaces networks.SparseToDensePool.forward (reference src/networks.py:2168-2196):
//   z = x[:, 0]; min-pools over the non-zero depths (999-sentinel semantics, :2175-2181),
//   max-pools (:2183-2186), cat, n_convolution x [conv1x1 + LeakyReLU], cat with x,
//   conv3x3 + LeakyReLU -- one launch, one read of x, one write of the n_filter maps.
//
// One workgroup (512 threads) produces a 32 x 32 output tile.  The sparse depth tile with
// a halo of R+1 (R = largest pool radius) is staged in LDS twice: `zmin` with zeros
// replaced by the 999 sentinel (+inf outside the image) and `zmax` (-inf outside).  Every
// pool is evaluated separably -- a row pass into `hbuf`, then a column pass straight into
// registers -- on the (32+2) x (32+2) "feature" region the 3x3 conv needs.  The 1x1 conv
// chain runs in registers, its outputs (plus the raw x channels) go to LDS as the 3x3
// conv's input tile, zero outside the image exactly like the reference's zero padding.
#include <math.h>
#include <stdlib.h>

#include <initializer_list>

#include "kbn_common.h"

namespace kbn {

constexpr int S2D_TW = 32, S2D_TH = 32, S2D_THREADS = 512;
constexpr int S2D_FW = S2D_TW + 2, S2D_FH = S2D_TH + 2;
constexpr int S2D_NF = S2D_FW * S2D_FH;                  // feature positions per tile (612)
constexpr int S2D_NPOS = (S2D_NF + S2D_THREADS - 1) / S2D_THREADS;           // feature positions per thread (3)
constexpr int S2D_MAXPOOL = 8, S2D_MAXF = 8, S2D_MAXCONV = 4, S2D_MAXIN = 2;
constexpr int S2D_MAXCH = S2D_MAXF + S2D_MAXIN;
// LDS weight block (floats): 3x3 conv as [ch][tap][8 filters], 1x1 convs as [input][8 filters]
constexpr int S2D_WC = S2D_MAXCH * 9 * 8, S2D_WP = 8 * 8;
constexpr int S2D_WFLOATS = S2D_WC + S2D_MAXCONV * S2D_WP;

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
    int hoff[S2D_MAXPOOL];  // LDS offset (floats) of each pool's row-pass buffer
    int nconv, nf, R, Rmin, Rmax;
    int tilesX, tilesY;
    int pool_floats;        // zmin + zmax + row-pass buffers
    int dbg;                // ablation (KBN_S2D_DEBUG): 1 no row pass, 2 no column pass, 4 no 1x1, 8 no 3x3, 16 no staging
    float slope;
};

// Pool configuration: compile-time lists for the reference's shipped presets (every pool loop
// unrolls to straight-line code: all LDS reads of a window issue back to back with immediate
// offsets), a run-time list for anything else.
template <int NMIN, int... KS>
struct StaticPools {
    static constexpr bool is_static = true;
    static constexpr int NP = sizeof...(KS);
    static constexpr int K[NP] = {KS...};
    static constexpr int cmax(int lo, int hi) {
        int m = 0;
        for (int i = lo; i < hi; ++i) m = (K[i] / 2 > m) ? K[i] / 2 : m;
        return m;
    }
    static constexpr int RMIN = cmax(0, NMIN), RMAX = cmax(NMIN, NP);
    __device__ static constexpr int nmin(const S2DParams&) { return NMIN; }
    __device__ static constexpr int npool(const S2DParams&) { return NP; }
    __device__ static constexpr int ksize(const S2DParams&, int pi) { return pi < NP ? K[pi] : 1; }
    __device__ static constexpr int rmin(const S2DParams&) { return RMIN; }
    __device__ static constexpr int rmax(const S2DParams&) { return RMAX; }
};
struct DynamicPools {
    static constexpr bool is_static = false;
    static constexpr int RMIN = 15, RMAX = 15;
    __device__ static int nmin(const S2DParams& p) { return p.nmin; }
    __device__ static int npool(const S2DParams& p) { return p.npool; }
    __device__ static int ksize(const S2DParams& p, int pi) { return p.ksize[pi]; }
    __device__ static int rmin(const S2DParams& p) { return p.Rmin; }
    __device__ static int rmax(const S2DParams& p) { return p.Rmax; }
};
using KittiPools = StaticPools<5, 5, 7, 9, 11, 13, 15, 17>;   // bash/kitti/run_kbnet_kitti_validation.sh:15-16
using VoidPools = StaticPools<2, 15, 17, 23, 27, 29>;         // bash/void/run_kbnet_void1500.sh:15-16
using VoidTrainPools = StaticPools<3, 15, 17, 19, 23, 27>;    // bash/void/train_kbnet_void1500.sh:21-22

template <typename CFG>
__global__ __launch_bounds__(S2D_THREADS) void s2d_kernel(const S2DParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int R = p.R;
    const int ZW = S2D_FW + 2 * R, ZH = S2D_FH + 2 * R;
    float* wl = smem;                      // weights, live for the whole kernel
    float* zmin = smem + S2D_WFLOATS;
    float* zmax = zmin + ZH * ZW;
    float* hb = zmax + ZH * ZW;            // row-pass buffers, one per pool
    float* feat = smem + S2D_WFLOATS;      // [(nf + inC)][FH][FW]; overlays the pool buffers (dead by then)

    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int oy0 = ty * S2D_TH, ox0 = tx * S2D_TW;
    const float* xz = p.x + (long long)n * p.x_bstride;  // channel 0 = sparse depth
    const int nch = p.nf + p.inC;

    // ---- weights -> LDS, transposed so that the 8 filters of one (input, tap) are contiguous --
    if (!p.pyramid) {
        for (int e = tid; e < S2D_WC; e += S2D_THREADS) {
            const int f = e & 7, t = (e >> 3) % 9, ch = e / 72;
            wl[e] = (f < p.nf && ch < nch) ? p.wconv[((long long)f * nch + ch) * 9 + t] : 0.f;
        }
        for (int e = tid; e < S2D_MAXCONV * S2D_WP; e += S2D_THREADS) {
            const int f = e & 7, q = (e >> 3) & 7, i = e >> 6;
            const int cin = (i == 0) ? CFG::npool(p) : p.nf;
            wl[S2D_WC + e] = (i < p.nconv && f < p.nf && q < cin) ? p.wpool[i][f * cin + q] : 0.f;
        }
    }
    // ---- stage the depth tile (+halo) -------------------------------------------------
    if (!(p.dbg & 16))
    for (int e = tid; e < ZH * ZW; e += S2D_THREADS) {
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

    // ---- row pass, nested: one sweep outwards gives every min pool, another every max pool ----
    const int nmin = CFG::nmin(p), npool = CFG::npool(p);
    if (!(p.dbg & 1))
    for (int e = tid; e < ZH * S2D_FW; e += S2D_THREADS) {
        const int r = e / S2D_FW, c = e - r * S2D_FW;
        {
            const float* s = zmin + r * ZW + c + R;
            float a = s[0];
#pragma unroll
            for (int d = 1; d <= CFG::RMIN; ++d) {
                if (d > CFG::rmin(p)) break;
                a = fminf(a, fminf(s[-d], s[d]));
#pragma unroll
                for (int pi = 0; pi < S2D_MAXPOOL; ++pi) {
                    if (pi < nmin && (CFG::ksize(p, pi) >> 1) == d) {
                        const int rr = r - (R - d);
                        if (rr >= 0 && rr < S2D_FH + 2 * d) hb[p.hoff[pi] + rr * S2D_FW + c] = a;
                    }
                }
            }
        }
        {
            const float* s = zmax + r * ZW + c + R;
            float a = s[0];
#pragma unroll
            for (int d = 1; d <= CFG::RMAX; ++d) {
                if (d > CFG::rmax(p)) break;
                a = fmaxf(a, fmaxf(s[-d], s[d]));
#pragma unroll
                for (int pi = 0; pi < S2D_MAXPOOL; ++pi) {
                    if (pi >= nmin && pi < npool && (CFG::ksize(p, pi) >> 1) == d) {
                        const int rr = r - (R - d);
                        if (rr >= 0 && rr < S2D_FH + 2 * d) hb[p.hoff[pi] + rr * S2D_FW + c] = a;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- column pass: pooled values of this thread's feature positions -> registers ------
    float pooled[S2D_NPOS][S2D_MAXPOOL];
#pragma unroll
    for (int u = 0; u < S2D_NPOS; ++u) {
        const int e = tid + u * S2D_THREADS;
#pragma unroll
        for (int pi = 0; pi < S2D_MAXPOOL; ++pi) {
            float a = 0.f;
            if (pi < npool && e < S2D_NF && !(p.dbg & 2)) {
                const int k = CFG::ksize(p, pi);
                const float* s = hb + p.hoff[pi] + e;  // rows fy .. fy+k-1 of this pool's buffer
                a = s[0];
                if (pi < nmin) {
                    if constexpr (CFG::is_static) {
#pragma unroll
                        for (int d = 1; d < 2 * CFG::RMIN + 1; ++d)
                            if (d < k) a = fminf(a, s[d * S2D_FW]);
                    } else {
#pragma unroll 4
                        for (int d = 1; d < k; ++d) a = fminf(a, s[d * S2D_FW]);
                    }
                    a = (a == 999.f) ? 0.f : a;  // where(pool == 999, 0, pool)
                } else {
                    if constexpr (CFG::is_static) {
#pragma unroll
                        for (int d = 1; d < 2 * CFG::RMAX + 1; ++d)
                            if (d < k) a = fmaxf(a, s[d * S2D_FW]);
                    } else {
#pragma unroll 4
                        for (int d = 1; d < k; ++d) a = fmaxf(a, s[d * S2D_FW]);
                    }
                }
            }
            pooled[u][pi] = a;
        }
    }
    __syncthreads();  // the pool buffers are dead from here on: `feat` overlays them

    // ---- 1x1 conv chain in registers; features + raw x channels to LDS ----------------
#pragma unroll
    for (int u = 0; u < S2D_NPOS; ++u) {
        const int e = tid + u * S2D_THREADS;
        if (e >= S2D_NF) continue;
        const int fy = e / S2D_FW, fx = e - fy * S2D_FW;
        const int Y = oy0 - 1 + fy, X = ox0 - 1 + fx;
        const bool inb = (Y >= 0 && Y < p.H && X >= 0 && X < p.W);
        if (p.pyramid) {
            if (inb && fy >= 1 && fy <= S2D_TH && fx >= 1 && fx <= S2D_TW) {
                float* py = p.pyramid + ((long long)n * npool) * p.H * p.W + (long long)Y * p.W + X;
#pragma unroll
                for (int pi = 0; pi < S2D_MAXPOOL; ++pi)
                    if (pi < npool) py[(long long)pi * p.H * p.W] = pooled[u][pi];
            }
            continue;
        }
        float h[S2D_MAXF], g[S2D_MAXF];
#pragma unroll
        for (int q = 0; q < S2D_MAXPOOL; ++q) h[q] = pooled[u][q];  // inputs of layer 0 (zero beyond npool)
#pragma unroll
        for (int i = 0; i < S2D_MAXCONV; ++i) {
            if (i >= p.nconv || (p.dbg & 4)) break;
            const f32x4* w4 = reinterpret_cast<const f32x4*>(wl + S2D_WC + i * S2D_WP);
#pragma unroll
            for (int f = 0; f < S2D_MAXF; ++f) g[f] = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {  // zero weights beyond the real fan-in
                const f32x4 wa = w4[q * 2], wb = w4[q * 2 + 1];
                g[0] = fmaf(wa[0], h[q], g[0]); g[1] = fmaf(wa[1], h[q], g[1]);
                g[2] = fmaf(wa[2], h[q], g[2]); g[3] = fmaf(wa[3], h[q], g[3]);
                g[4] = fmaf(wb[0], h[q], g[4]); g[5] = fmaf(wb[1], h[q], g[5]);
                g[6] = fmaf(wb[2], h[q], g[6]); g[7] = fmaf(wb[3], h[q], g[7]);
            }
#pragma unroll
            for (int f = 0; f < S2D_MAXF; ++f) h[f] = leaky_relu(g[f], p.slope);
        }
#pragma unroll
        for (int f = 0; f < S2D_MAXF; ++f)
            if (f < p.nf) feat[f * S2D_NF + e] = inb ? h[f] : 0.f;
#pragma unroll
        for (int ci = 0; ci < S2D_MAXIN; ++ci)
            if (ci < p.inC)
                feat[(p.nf + ci) * S2D_NF + e] = inb ? xz[(long long)ci * p.H * p.W + (long long)Y * p.W + X] : 0.f;
    }
    if (p.pyramid) return;
    __syncthreads();

    // ---- 3x3 conv over [features | x] + LeakyReLU.  The phase is LDS-issue bound (one value per FMA pair
    //      plus the broadcast weight reads), so half of the threads take 4 consecutive pixels of a row each:
    //      a row of the window is 6 values = three 8-byte reads (+ two for the odd-aligned pairs the packed
    //      FMAs of the middle tap need), and every weight read serves four pixels.  Packed fp32: one
    //      v_pk_fma_f32 per (channel, tap, filter, pixel pair), weight broadcast to both halves. ----------
    if (tid < S2D_TH * S2D_TW / 4) {
        const int oy = tid / (S2D_TW / 4), oxq = (tid - oy * (S2D_TW / 4)) * 4;
        f32x2 acc[2][S2D_MAXF];   // [pixel pair (0,1) / (2,3)][filter]
#pragma unroll
        for (int f = 0; f < S2D_MAXF; ++f) acc[0][f] = acc[1][f] = (f32x2){0.f, 0.f};
        for (int ch = 0; ch < ((p.dbg & 8) ? 0 : nch); ++ch) {
            const float* fr = feat + ch * S2D_NF + oy * S2D_FW + oxq;   // column oxq <-> x - 1 of the first pixel
            const f32x4* w4 = reinterpret_cast<const f32x4*>(wl + ch * 72);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* r = fr + ky * S2D_FW;
                const f32x2 e0 = *reinterpret_cast<const f32x2*>(r), e1 = *reinterpret_cast<const f32x2*>(r + 2),
                            e2 = *reinterpret_cast<const f32x2*>(r + 4);
                const f32x2 o0 = (f32x2){r[1], r[2]}, o1 = (f32x2){r[3], r[4]};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x2 va = kx == 0 ? e0 : (kx == 1 ? o0 : e1);   // pixels 0, 1 see columns kx, kx + 1
                    const f32x2 vb = kx == 0 ? e1 : (kx == 1 ? o1 : e2);   // pixels 2, 3 see columns kx + 2, kx + 3
                    const f32x4 wa = w4[(ky * 3 + kx) * 2], wb = w4[(ky * 3 + kx) * 2 + 1];
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        acc[0][f] = __builtin_elementwise_fma(va, (f32x2){wa[f], wa[f]}, acc[0][f]);
                        acc[1][f] = __builtin_elementwise_fma(vb, (f32x2){wa[f], wa[f]}, acc[1][f]);
                        acc[0][4 + f] = __builtin_elementwise_fma(va, (f32x2){wb[f], wb[f]}, acc[0][4 + f]);
                        acc[1][4 + f] = __builtin_elementwise_fma(vb, (f32x2){wb[f], wb[f]}, acc[1][4 + f]);
                    }
                }
            }
        }
        const int Y = oy0 + oy, X = ox0 + oxq;
        if (Y < p.H && X < p.W) {
            const long long HW = (long long)p.H * p.W;
            float* o = p.out + ((long long)n * p.nf) * HW + (long long)Y * p.W + X;
            const bool vec = (X + 3 < p.W) && ((p.W & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
#pragma unroll
            for (int f = 0; f < S2D_MAXF; ++f) {
                if (f >= p.nf) continue;
                const f32x4 v = (f32x4){leaky_relu(acc[0][f].x, p.slope), leaky_relu(acc[0][f].y, p.slope),
                                        leaky_relu(acc[1][f].x, p.slope), leaky_relu(acc[1][f].y, p.slope)};
                if (vec) {
                    *reinterpret_cast<f32x4*>(o + f * HW) = v;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (X + i < p.W) o[f * HW + i] = v[i];
                }
            }
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
    for (int i = n_min + n_max; i < S2D_MAXPOOL; ++i) { p.ksize[i] = 1; p.hoff[i] = 0; }
    p.nmin = n_min;
    p.npool = n_min + n_max;
    p.R = R;
    p.Rmin = 0; p.Rmax = 0;
    const int ZW = S2D_FW + 2 * R, ZH = S2D_FH + 2 * R;
    int off = 0;
    for (int i = 0; i < p.npool; ++i) {
        const int rad = p.ksize[i] / 2;
        if (i < n_min) { if (rad > p.Rmin) p.Rmin = rad; } else { if (rad > p.Rmax) p.Rmax = rad; }
        p.hoff[i] = off;
        off += (S2D_FH + 2 * rad) * S2D_FW;
    }
    p.tilesX = ceil_div(p.W, S2D_TW);
    p.tilesY = ceil_div(p.H, S2D_TH);
    size_t pool_floats = (size_t)2 * ZH * ZW + (size_t)off;
    size_t feat_floats = (size_t)S2D_MAXCH * S2D_NF;
    p.pool_floats = (int)pool_floats;
    p.dbg = knob(KNOB_S2D_DEBUG);
    size_t lds = sizeof(float) * (S2D_WFLOATS + (pool_floats > feat_floats ? pool_floats : feat_floats));
    if (lds > 160 * 1024) return KBN_ERR_UNSUPPORTED;
    long long blocks = (long long)p.tilesX * p.tilesY * p.N;
    if (blocks > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    auto matches = [&](int nm, std::initializer_list<int> ks) {
        if (p.nmin != nm || p.npool != (int)ks.size()) return false;
        int i = 0;
        for (int k : ks)
            if (p.ksize[i++] != k) return false;
        return true;
    };
    auto launch = [&](auto kern, DeviceOnce& once) -> int {
        if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(S2D_THREADS), lds, stream, p);
        return KBN_OK;
    };
    static DeviceOnce set_kitti, set_void, set_voidtrain, set_dyn;
    int rc;
    if (matches(5, {5, 7, 9, 11, 13, 15, 17})) rc = launch(s2d_kernel<KittiPools>, set_kitti);
    else if (matches(2, {15, 17, 23, 27, 29})) rc = launch(s2d_kernel<VoidPools>, set_void);
    else if (matches(3, {15, 17, 19, 23, 27})) rc = launch(s2d_kernel<VoidTrainPools>, set_voidtrain);
    else rc = launch(s2d_kernel<DynamicPools>, set_dyn);
    if (rc != KBN_OK) return rc;
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
