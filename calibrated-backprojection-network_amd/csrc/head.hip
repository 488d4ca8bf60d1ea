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
    if (aligned) {
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

// ================================================================================================
// Fused tail of the decoder: DecoderBlock `deconv0`'s second conv (C -> C, 3x3, LeakyReLU; reference
// src/net_utils.py:1485-1487) + output0 (C -> 1, 3x3, linear; src/networks.py:1985) + the sigmoid depth
// mapping (src/kbnet_model.py:181-184) in ONE launch: the C-channel full-resolution tensor between the two
// convs (2 x 20.5 MB per KITTI frame of HBM traffic) never leaves the CU.
//
// Workgroup = 512 threads = 64 x 16 output pixels (two workgroups per CU).  The C input planes of the tile (+2 halo; columns from
// x0-4: every 16-byte granule is 4 in-image pixels) reach LDS by LDS-DMA.  The first conv runs on
// v_mfma_f32_16x16x4_f32 over the 66 x 18 positions the head needs (M = 16 consecutive positions of the
// flattened region, N = 16 filters of which C are real, K = 9 taps x C channels in steps of 4 channels);
// the B operand (raw OIHW weights) sits in registers for the whole kernel.  Its activated output -- zero
// outside the image, like the reference's padding -- overlays the input tile in LDS, and each thread
// finishes 2 consecutive pixels of a row: 3 x 4 window per channel, sigmoid, depth mapping, 8-byte store.
namespace kbn {

constexpr int CH_TW = 64, CH_TH = 16;
constexpr int CH_FW = CH_TW + 2, CH_FH = CH_TH + 2;                // positions of the first conv (66 x 18)
constexpr int CH_NPOS = CH_FW * CH_FH;                             // 1188
constexpr int CH_NMB = (CH_NPOS + 15) / 16;                        // 75 m-blocks
constexpr int CH_THREADS = 512, CH_WAVES = CH_THREADS / 64;
constexpr int CH_MW = (CH_NMB + CH_WAVES - 1) / CH_WAVES;          // m-blocks per wave (10; the surplus ones are padding)
constexpr int CH_COLS = CH_TW + 8, CH_ROWS = CH_TH + 4;            // staged input: columns x0-4 .. x0+67, rows y0-2 .. y0+17
constexpr int CH_PLANE = ((CH_ROWS * CH_COLS + 15) / 32) * 32 + 16;   // 16 (mod 32): k / k+1 planes of a half-wave on disjoint banks
constexpr int CH_NF4 = CH_ROWS * (CH_COLS / 4);                    // granules per channel (360)
constexpr int CH_NJ = (CH_NF4 + 63) / 64;                          // DMA instructions per channel (6)
constexpr int CH_FP = CH_FW, CH_FPLANE = CH_FH * CH_FP;            // feature tile: pitch = row length, so a flattened position IS its offset

struct ConvHeadParams {
    const float* x;
    long long x_bstride;
    const float* wconv;     // C x C x 3 x 3
    const float* wout;      // 1 x C x 3 x 3
    float* depth;
    float* logits;
    int C, H, W, tilesX, tilesY, nblocks;
    int act;
    int dbg;   // ablation (KBN_DEBUG): 1 no input staging, 2 no MFMAs, 4 no feature exchange / head arithmetic
    float slope, dmin, ratio;
};

template <int NG>   // channel groups of 4: C = 4 NG
__global__ __launch_bounds__(CH_THREADS, 4) void conv_head_kernel(const ConvHeadParams p) {
    constexpr int C = 4 * NG;
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [C][CH_PLANE] input, later [C][CH_FPLANE] features
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int H = p.H, W = p.W;
    const long long HW = (long long)H * W;

    // ---- B operand of the first conv, fragment order in LDS behind the tile: bw[s][lane] = W[filter li][channel
    //      4g + lk][tap], k-step s = tap * NG + g (one read per k-step and wave; keeps 9 NG registers free) ----
    float* const bws = tile + C * CH_PLANE;
    for (int e = tid; e < 9 * NG * 64; e += CH_THREADS) {
        const int s = e >> 6, l = e & 63, tap = s / NG, g = s - tap * NG, f = l & 15, k = l >> 4;
        bws[e] = f < C ? p.wconv[((long long)f * C + 4 * g + k) * 9 + tap] : 0.f;
    }
    // Persistent workgroups, two per CU: the weight table above and the lane's granule offsets are set up once.
    // (Starting half of the workgroups late, so that one's staging / head would run under the other's MFMAs, changes
    // nothing -- measured: the phases are issue bound, not latency bound; see tools/head_bench.py.)
    unsigned grel[CH_NJ];             // byte offset of this lane's granules from the tile's first staged element
    unsigned long long gfull[CH_NJ];  // lanes that own a granule at all
#pragma unroll
    for (int j = 0; j < CH_NJ; ++j) {
        const int f = j * 64 + lane, r = f / (CH_COLS / 4), cv = f - r * (CH_COLS / 4);
        grel[j] = f < CH_NF4 ? (unsigned)((r * W + cv * 4) * 4) : 0u;
        gfull[j] = __ballot(f < CH_NF4);
    }

  for (int tile_id = blockIdx.x; tile_id < p.nblocks; tile_id += gridDim.x) {
    asm volatile("" ::: "memory");   // nothing read from memory is hoisted out of the tile loop (108 scalar weights would spill)
    int bid = xcd_remap(tile_id, p.nblocks);
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int oy0 = ty * CH_TH, ox0 = tx * CH_TW;
    const float* xn = p.x + (long long)n * p.x_bstride;

    // ---- stage the input tile: zero padding only where the tile leaves the image ----
    const bool border = oy0 - 2 < 0 || oy0 + CH_TH + 2 > H || ox0 - 4 < 0 || ox0 + CH_TW + 4 > W;   // block-uniform
    if (border) {
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int e = tid * 4; e < C * CH_PLANE; e += 4 * CH_THREADS) *reinterpret_cast<f32x4*>(tile + e) = zero;
        __syncthreads();
    }
    {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(tile));
        if (!border) {
            // interior tile: every granule is in the image; the lane's offset from the tile's first staged
            // element is tile invariant, the tile position goes into the (scalar) base pointer
            const float* origin = xn + (long long)(oy0 - 2) * W + (ox0 - 4);
            for (int ch = wave; ch < ((p.dbg & 1) ? 0 : C); ch += CH_WAVES) {   // wave w stages channels w, w+8
                const float* src = uniform_ptr(origin + (long long)ch * HW);
                const unsigned dst = lds0 + 4u * (unsigned)(ch * CH_PLANE);
#pragma unroll
                for (int j = 0; j < CH_NJ; ++j) lds_dma16_sm(src, grel[j], dst + j * 1024, gfull[j]);
            }
        } else {
            unsigned gv[CH_NJ];
            unsigned long long gm[CH_NJ];
#pragma unroll
            for (int j = 0; j < CH_NJ; ++j) {
                const int f = j * 64 + lane;
                int g = -1;
                if (f < CH_NF4) {
                    const int r = f / (CH_COLS / 4), cv = f - r * (CH_COLS / 4);
                    const int Y = oy0 - 2 + r, X = ox0 - 4 + cv * 4;
                    if (Y >= 0 && Y < H && X >= 0 && X < W) g = (Y * W + X) * 4;
                }
                gv[j] = g < 0 ? 0u : (unsigned)g;
                gm[j] = __ballot(g >= 0);
            }
            for (int ch = wave; ch < ((p.dbg & 1) ? 0 : C); ch += CH_WAVES) {
                const float* src = uniform_ptr(xn + (long long)ch * HW);
                const unsigned dst = lds0 + 4u * (unsigned)(ch * CH_PLANE);
#pragma unroll
                for (int j = 0; j < CH_NJ; ++j) lds_dma16_sm(src, gv[j], dst + j * 1024, gm[j]);
            }
        }
    }

    // ---- first conv on the matrix cores: m-block mb = wave + 8 mi covers flattened positions 16 mb .. 16 mb + 15 ----
    int abase[CH_MW];
#pragma unroll
    for (int mi = 0; mi < CH_MW; ++mi) {
        int pos = (wave + CH_WAVES * mi) * 16 + li;
        pos = pos < CH_NPOS ? pos : 0;                       // padding rows of the last m-block: any valid address
        const int fr = pos / CH_FW, fc = pos - fr * CH_FW;
        abase[mi] = lk * CH_PLANE + fr * CH_COLS + fc + 2;   // window origin of position (fr, fc), channel lk of a group
    }
    f32x4 acc[CH_MW];
#pragma unroll
    for (int mi = 0; mi < CH_MW; ++mi) acc[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // K loop, fully unrolled and software pipelined: the A fragments of k-step s+1 are read while the MFMAs of
    // step s issue (straight-line code: the padding m-block of wave 3 is computed like the others and dropped).
    {
        constexpr int NS = 9 * NG;
        auto koff = [](int s) { const int tap = s / NG, g = s - tap * NG; return g * 4 * CH_PLANE + (tap / 3) * CH_COLS + (tap % 3); };
        float a[2][CH_MW], b[2];
        b[0] = bws[lane];
#pragma unroll
        for (int mi = 0; mi < CH_MW; ++mi) a[0][mi] = tile[abase[mi] + koff(0)];
        if (!(p.dbg & 2))
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) {
                b[(s + 1) & 1] = bws[(s + 1) * 64 + lane];
#pragma unroll
                for (int mi = 0; mi < CH_MW; ++mi) a[(s + 1) & 1][mi] = tile[abase[mi] + koff(s + 1)];
            }
#pragma unroll
            for (int mi = 0; mi < CH_MW; ++mi)
                acc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][mi], b[s & 1], acc[mi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch distance at one k-step (registers)
        }
    }
    __syncthreads();   // every read of the input tile is done: the features overlay it

    // ---- activated features -> LDS [C][FH][FP]; acc[mi][r] = position 16 mb + 4 lk + r, filter li ----
    if (li < C && !(p.dbg & 4)) {
        // The matrix pipe and the vector ALU share issue slots (every vector instruction here is time the other
        // workgroup's MFMAs do not get): feature pitch = row length makes position 16 mb + 4 lk + r its own LDS offset,
        // so an m-block's four values are one 16-byte store at a compile-time offset from one base register, and
        // LeakyReLU is max(v, slope v) (slope <= 1; slope 1 = no activation).
        const float slope = p.act ? p.slope : 1.f;
        float* const fdst = tile + li * CH_FPLANE + wave * 16 + 4 * lk;
        const bool last_ok = wave * 16 + 4 * lk + (CH_MW - 1) * CH_WAVES * 16 + 3 < CH_NPOS;   // NPOS is a multiple of 4
        int bpos = wave * 16 + 4 * lk;
        if (border) asm volatile("" : "+v"(bpos));   // (keeps the border path's decodes inside the tile loop: registers)
#pragma unroll
        for (int mi = 0; mi < CH_MW; ++mi) {
            if (wave + CH_WAVES * mi >= CH_NMB) continue;   // wave-uniform
            f32x4 v = acc[mi];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = v[r] * slope, o;
                asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(v[r]), "v"(t));
                v[r] = o;
            }
            if (border) {   // zero padding of the feature map at the image border (block-uniform, rare)
                const int pos0 = bpos + mi * CH_WAVES * 16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int fr = (pos0 + r) / CH_FW, fc = (pos0 + r) - fr * CH_FW;
                    const int Y = oy0 - 1 + fr, X = ox0 - 1 + fc;
                    v[r] = (Y >= 0 && Y < H && X >= 0 && X < W) ? v[r] : 0.f;
                }
            }
            if (mi < CH_MW - 1 || last_ok) *reinterpret_cast<f32x4*>(fdst + mi * CH_WAVES * 16) = v;
        }
    }
    __syncthreads();

    // ---- output0 + sigmoid mapping: thread -> 2 consecutive pixels of row tid / 32 ----
    const int oy = tid >> 5, oxq = (tid & 31) * 2;
    float o[2] = {0.f, 0.f};
    if (!(p.dbg & 4))
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        const float* t = tile + ch * CH_FPLANE + oy * CH_FP + oxq;
        const float* wc = p.wout + ch * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const f32x2 a = *reinterpret_cast<const f32x2*>(t + ky * CH_FP);
            const f32x2 b = *reinterpret_cast<const f32x2*>(t + ky * CH_FP + 2);
            const float v[4] = {a[0], a[1], b[0], b[1]};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float wk = wc[ky * 3 + kx];
                o[0] = fmaf(wk, v[kx], o[0]);
                o[1] = fmaf(wk, v[kx + 1], o[1]);
            }
        }
    }
    const int Y = oy0 + oy, X = ox0 + oxq;
    if (Y < H && X < W) {   // W % 4 == 0: a pair is inside or outside as a whole
        const long long oo = (long long)n * HW + (long long)Y * W + X;
        f32x2 d;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float sg = 1.0f / (1.0f + expf(-o[i]));
            d[i] = p.dmin / (sg + p.ratio);
        }
        if (p.logits) *reinterpret_cast<f32x2*>(p.logits + oo) = (f32x2){o[0], o[1]};
        *reinterpret_cast<f32x2*>(p.depth + oo) = d;
    }
    __syncthreads();   // the next tile's staging overwrites the features
  }   // tile loop
}

}  // namespace kbn

extern "C" int kbn_conv_head_forward(const float* x, long long x_batch_stride, const float* w_conv, const float* w_out,
                                     float* depth, float* logits, int n, int channels, int height, int width,
                                     int apply_activation, float negative_slope, float min_predict_depth,
                                     float max_predict_depth, kbn_stream_t stream) {
    using namespace kbn;
    if (!x || !w_conv || !w_out || !depth || n < 1 || channels < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (channels > 16 || (channels & 3) || (width & 3) || (x_batch_stride & 3)) return KBN_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(depth) & 15) ||
        (logits && (reinterpret_cast<uintptr_t>(logits) & 15)))
        return KBN_ERR_UNSUPPORTED;
    if ((long long)height * width > 0x1fffffffLL) return KBN_ERR_UNSUPPORTED;   // 32-bit byte offsets inside a plane
    if (knob(KNOB_NO_HEAD_FUSION)) return KBN_ERR_UNSUPPORTED;
    if (apply_activation && !(negative_slope >= 0.f && negative_slope <= 1.f)) return KBN_ERR_UNSUPPORTED;   // max(v, slope v) form
    ConvHeadParams p;
    p.x = x; p.x_bstride = x_batch_stride; p.wconv = w_conv; p.wout = w_out; p.depth = depth; p.logits = logits;
    p.C = channels; p.H = height; p.W = width;
    p.tilesX = ceil_div(width, CH_TW); p.tilesY = ceil_div(height, CH_TH);
    const long long blocks = (long long)p.tilesX * p.tilesY * n;
    if (blocks > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.nblocks = (int)blocks;
    p.act = apply_activation ? 1 : 0; p.slope = negative_slope;
    p.dbg = knob(KNOB_DEBUG);
    p.dmin = min_predict_depth;
    p.ratio = (float)((double)min_predict_depth / (double)max_predict_depth);   // evaluated in double like the reference's scalar
    const size_t lds = sizeof(float) * ((size_t)channels * CH_PLANE + (size_t)9 * (channels / 4) * 64);
    auto launch = [&](auto kern, DeviceOnce& once) -> int {
        if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
        int cus = device_cu_count();
        if (cus < 1) cus = 256;
        const long long grid = blocks < 2LL * cus ? blocks : 2LL * cus;   // persistent: two workgroups per CU
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CH_THREADS), lds, (hipStream_t)stream, p);
        return KBN_OK;
    };
    static DeviceOnce o1, o2, o3, o4;
    int rc;
    switch (channels / 4) {
        case 1: rc = launch(conv_head_kernel<1>, o1); break;
        case 2: rc = launch(conv_head_kernel<2>, o2); break;
        case 3: rc = launch(conv_head_kernel<3>, o3); break;
        default: rc = launch(conv_head_kernel<4>, o4); break;
    }
    if (rc != KBN_OK) return rc;
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}
