// conv_bf16.hip -- THROUGHPUT-ONLY bf16 leg of the decoder (BASELINE.json configs[2]: "KITTI 352x1216 batch=32, bf16").
//
// The parity-gated path of this library is fp32 end to end (bf16 convs miss north_star's 1e-4 bar by two
// orders of magnitude).  This file is the measured alternative the brief asks to report NEXT to it, never instead of it:
// the decoder's 3x3 stride-1 convs -- nearest-2x up-conv and concat conv of every DecoderBlock, reference
// src/net_utils.py:484-499, 1483-1487; 81 % of the network's FLOPs -- with bf16 MFMA operands and fp32 accumulation.
// Activations stay fp32 NCHW in HBM (the drop-in layout; the rest of the network keeps running the fp32 kernels), inputs
// and weights are rounded to bf16 (round to nearest even, v_cvt_pk_bf16_f32) on their way into LDS, products accumulate
// in fp32 on v_mfma_f32_32x32x16_bf16.  bench.py reports its rate and its measured error under separate keys.
//
// Workgroup = 256 threads = 16 x 32 output pixels x 64 filters.  K loop over chunks of 16 input channels (all 9 taps per
// chunk), double buffered:
//   A  the (16+2) x (32+2) input tile of the chunk (up-conv: the (8+2) x (16+2) low-resolution pixels it maps to; the
//      fragment reads do the upsampling): a thread loads the 8 channels of one pixel (8 coalesced dword loads with a
//      scalar plane base), converts and writes ONE 16-byte LDS word; layout [k-group][pixel][8 channels] -> the MFMA A
//      fragment of a lane (pixel = lane % 32, k-group = lane / 32) is one ds_read_b128;
//   B  the chunk's weights, pre-packed [tap][k-group][filter][8 channels] bf16: a straight LDS-DMA copy; B fragment =
//      one ds_read_b128 (filter = lane % 32).
// Wave w owns output rows 4w .. 4w+3 (four 32-pixel m-blocks) x two 32-filter n-blocks: per tap 4 A + 2 B reads, 8 MFMAs.
// Stride-2 variant (the image convs of the KB blocks and conv5, reference src/net_utils.py:1348, src/networks.py:521-525):
// 4 x 32 output pixels per workgroup, (8+1) x (64+1) staged pixels, fragment pixel (2 y + ky, 2 x + kx).
// With 16x the fp32 matrix rate the kernel is bound by its operand traffic (L2 -> LDS weights, HBM inputs), not by MFMAs.
// (Tried: x-aligned 40-column tiles with 16-byte staging loads -- slower, 465 -> 551 us on deconv1's conv at batch 16:
// fewer staging items than threads, 18 % more bytes, 64 load registers.)
#include "conv_common.h"

namespace kbn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BF_TH = 16, BF_TW = 32, BF_NT = 64, BF_CK = 16, BF_MB = 4;   // tile, filters per workgroup, chunk, m-blocks (rows) per wave
// Staged input tile of a chunk.  Plain conv: the (16+2) x (32+2) pixels around the output tile.  Nearest-2x up-conv: the
// (8+2) x (16+2) LOW-resolution pixels those map to -- the fragment reads do the upsampling ((y + ky + 1) >> 1,
// (x + kx + 1) >> 1 in staged coordinates), so every source pixel is fetched once, not four times.
template <int MODE>   // 0 plain 3x3, 1 nearest-2x up-conv, 2 stride-2 conv
struct BfGeom {
    static constexpr bool UP = MODE == 1, S2 = MODE == 2;
    static constexpr int MB = S2 ? 1 : BF_MB;                     // 32-pixel rows per wave
    static constexpr int TH = 4 * MB;                             // output rows per workgroup (16, stride 2: 4)
    static constexpr int ROWS = UP ? TH / 2 + 2 : (S2 ? 2 * TH + 1 : TH + 2);
    static constexpr int COLS = UP ? BF_TW / 2 + 2 : (S2 ? 2 * BF_TW + 1 : BF_TW + 2), NPIX = ROWS * COLS;
    static constexpr int A_BYTES = 2 * NPIX * 16;                 // [k-group][pixel][8 bf16]
    static constexpr int PR = (NPIX + 255) / 256, ROUNDS = 2 * PR; // staging round u = k-group * PR + pixel round
};
template <int MODE, int NB>
struct BfStage {
    static constexpr int B_BYTES = 9 * 2 * NB * 32 * 16;          // [tap][k-group][filter][8 bf16]
    static constexpr int BYTES = BfGeom<MODE>::A_BYTES + B_BYTES;
};

struct Bf16ConvParams {
    const float* src[2];
    long long src_bstride[2];
    int srcC[2];
    int nsrc;
    const unsigned short* wp;   // packed bf16 weights: [n-tile][chunk][tap][k-group][64 filters][8 channels]
    float* out;
    long long out_bstride;
    int N, OC, Cin, H, W;       // output size
    int sH, sW;                 // source planes: H x W, (H/2) x (W/2) for the up-conv, the input size of a stride-2 conv
    int up2x;
    int tilesX, tilesY, nTilesN, nblocks;
    int act;
    float slope;
};

__device__ __forceinline__ unsigned short bf16_bits(float v) {   // round to nearest even
    const bf16x2 c = __builtin_convertvector((f32x2){v, 0.f}, bf16x2);
    return __builtin_bit_cast(unsigned short, c[0]);
}

// OIHW fp32 -> [n-tile][chunk][tap][k-group][n][8 k] bf16, zero padded in both channel directions
__global__ void pack_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed, int OC, int Cin,
                                 int nchunks, int NT, long long total) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    int r = (int)(e % (9 * 2 * NT * 8));
    const long long q = e / (9 * 2 * NT * 8);
    const int chunk = (int)(q % nchunks), nt = (int)(q / nchunks);
    const int tap = r / (2 * NT * 8); r -= tap * 2 * NT * 8;
    const int g = r / (NT * 8); r -= g * NT * 8;
    const int n = r >> 3, k = r & 7;
    const int c = chunk * BF_CK + g * 8 + k, oc = nt * NT + n;
    const float v = (c < Cin && oc < OC) ? w[((long long)oc * Cin + c) * 9 + tap] : 0.f;
    packed[e] = bf16_bits(v);
}

template <int NB, int MODE>   // NB: 32-filter n-blocks per workgroup (2, or 1 for layers with <= 32 filters); MODE: see BfGeom
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_kernel(const Bf16ConvParams p) {
    using G = BfGeom<MODE>;
    using ST = BfStage<MODE, NB>;
    constexpr bool UP = G::UP, S2 = G::S2;
    constexpr int MB = G::MB;
    constexpr int NT = NB * 32, B_BYTES = ST::B_BYTES, STAGE = ST::BYTES, NPIX = G::NPIX, PR = G::PR, ROUNDS = G::ROUNDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 31, g = lane >> 5;
    int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int nt = bid % p.nTilesN;
    bid /= p.nTilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int oy0 = ty * G::TH, ox0 = tx * BF_TW;
    const int H = p.H, W = p.W;
    const int sH = p.sH, sW = p.sW;                              // source planes
    const long long plane = (long long)sH * sW;
    const int nchunks = p.Cin / BF_CK;

    // ---- this thread's staging pixels (the same for both k-groups): byte offset inside a source plane, or -1 ----
    int goff[PR];
#pragma unroll
    for (int u = 0; u < PR; ++u) {
        const int pix = u * 256 + tid;
        const int r = pix / G::COLS, c = pix - r * G::COLS;
        const int Y = (UP ? (oy0 >> 1) : (S2 ? 2 * oy0 : oy0)) - 1 + r, X = (UP ? (ox0 >> 1) : (S2 ? 2 * ox0 : ox0)) - 1 + c;
        goff[u] = (pix < NPIX && Y >= 0 && Y < sH && X >= 0 && X < sW) ? (Y * sW + X) * 4 : -1;
    }

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(reinterpret_cast<const float*>(smem)));
    const unsigned short* wp_nt = p.wp + (long long)nt * nchunks * (B_BYTES / 2);

    float va[ROUNDS][8];
    // chunk -> (source, first channel inside it); every source holds a multiple of 16 channels (launcher).  The channel
    // plane goes into the scalar base of the load, the lane contributes its tile-invariant pixel offset: no vector
    // address arithmetic per load (vector instructions are paid for in matrix-pipe time).
    auto load_chunk = [&](int chunk) {
        int c = chunk * BF_CK, s = 0;
        if (p.nsrc > 1 && c >= p.srcC[0]) { c -= p.srcC[0]; s = 1; }
        const float* base = p.src[s] + (long long)n * p.src_bstride[s] + (long long)c * plane;
#pragma unroll
        for (int u = 0; u < ROUNDS; ++u) {
            const int kg = u / PR, pr = u - kg * PR;
            const unsigned voff = goff[pr] < 0 ? 0u : (unsigned)goff[pr];   // masked lanes read element 0 (zeroed in store_chunk)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // scalar base + 32-bit lane offset, issued from asm (hipcc turns the C form into flat loads with 64-bit
                // vector address arithmetic per load); completion is awaited by the explicit vmcnt(0) before store_chunk
                const float* sb = base + (long long)(kg * 8 + k) * plane;   // wave-uniform: scalar registers
                asm volatile("global_load_dword %0, %1, %2" : "=v"(va[u][k]) : "v"(voff), "s"(sb) : "memory");
            }
        }
    };
    auto store_chunk = [&](int buf) {
        unsigned char* A = smem + buf * STAGE;
#pragma unroll
        for (int u = 0; u < ROUNDS; ++u) {
            const int kg = u / PR, pr = u - kg * PR, pix = pr * 256 + tid;
            if (pix >= NPIX) continue;
            const bool ok = goff[pr] >= 0;       // zero padding outside the image
            bf16x8 v;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const bf16x2 c = __builtin_convertvector((f32x2){ok ? va[u][k] : 0.f, ok ? va[u][k + 1] : 0.f}, bf16x2);
                v[k] = c[0]; v[k + 1] = c[1];
            }
            *reinterpret_cast<bf16x8*>(A + (kg * NPIX + pix) * 16) = v;
        }
    };
    auto stage_b = [&](int buf, int chunk) {
        const float* src = reinterpret_cast<const float*>(wp_nt + (long long)chunk * (B_BYTES / 2));
        const unsigned dst = lds0 + (unsigned)(buf * STAGE + G::A_BYTES);
        constexpr int n4 = B_BYTES / 16;                   // 1152 / 576 granules
#pragma unroll
        for (int e0 = 0; e0 < n4; e0 += 256) {
            const int eb = e0 + wave * 64;
            if (eb + lane < n4) lds_dma16_s(src + eb * 4, (unsigned)(lane * 16), dst + eb * 16);
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

    // A fragment of (row MB wave + mb, pixel lm, tap (ky, kx)): staged pixel (MB w + mb + ky, lm + kx); up-conv: the
    // low-resolution pixel ((MB w + mb + ky + 1) >> 1, (lm + kx + 1) >> 1); stride 2: (2 (MB w + mb) + ky, 2 lm + kx)
    int acol[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) acol[kx] = (g * NPIX + (UP ? ((lm + kx + 1) >> 1) : (S2 ? 2 * lm + kx : lm + kx))) * 16;
    const int arow = (UP ? MB / 2 : (S2 ? 2 * MB : MB)) * wave * G::COLS * 16;
    const int bbase = (g * NT + lm) * 16;
    auto compute = [&](int buf) {
        const unsigned char* A = smem + buf * STAGE + arow;
        const unsigned char* B = smem + buf * STAGE + G::A_BYTES + bbase;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            bf16x8 a[MB], b[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int r = UP ? ((mb + ky + 1) >> 1) : (S2 ? 2 * mb + ky : mb + ky);
                a[mb] = *reinterpret_cast<const bf16x8*>(A + acol[kx] + r * G::COLS * 16);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[nb] = *reinterpret_cast<const bf16x8*>(B + (tap * 2 * NT + nb * 32) * 16);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
        }
    };

    load_chunk(0);
    stage_b(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < nchunks;
        if (more) {
            load_chunk(c + 1);            // global loads in flight under the MFMAs
            stage_b(cur ^ 1, c + 1);
        }
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the asm loads of load_chunk and the weight DMA have landed
        if (more) store_chunk(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: acc[mb][nb][i]: pixel x = 8 (i / 4) + 4 (lane / 32) + (i % 4) of row 4 wave + mb, filter nb * 32 + lane % 32
    const long long HW = (long long)H * W;
    float* outn = p.out + (long long)n * p.out_bstride;
    const float slope = p.act ? p.slope : 1.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int Y = oy0 + MB * wave + mb;
        if (Y >= H) continue;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int oc = nt * NT + nb * 32 + lm;
            if (oc >= p.OC) continue;
            float* o = outn + (long long)oc * HW + (long long)Y * W + ox0 + 4 * g;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int X = ox0 + 8 * q4 + 4 * g;
                if (X >= W) continue;                                   // W % 4 == 0: a quad is in or out as a whole
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = acc[mb][nb][q4 * 4 + j];
                    v[j] = t > 0.f ? t : t * slope;
                }
                *reinterpret_cast<f32x4*>(o + 8 * q4) = v;
            }
        }
    }
}

}  // namespace kbn

extern "C" {

static int bf16_nt(int out_channels) { return out_channels <= 32 ? 32 : kbn::BF_NT; }   // filters per workgroup

size_t kbn_conv3x3_bf16_packed_weight_bytes(int out_channels, int in_channels) {
    using namespace kbn;
    if (out_channels < 1 || in_channels < 1 || (in_channels % BF_CK) != 0) return 0;
    const int nt = bf16_nt(out_channels);
    return (size_t)ceil_div(out_channels, nt) * (in_channels / BF_CK) * (9 * 2 * nt * 16);
}

int kbn_conv3x3_bf16_pack_weight(const float* weight, void* packed, int out_channels, int in_channels, kbn_stream_t stream) {
    using namespace kbn;
    const size_t bytes = kbn_conv3x3_bf16_packed_weight_bytes(out_channels, in_channels);
    if (!weight || !packed || bytes == 0) return KBN_ERR_INVALID_ARGUMENT;
    const long long total = (long long)(bytes / 2);
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight,
                       static_cast<unsigned short*>(packed), out_channels, in_channels, in_channels / BF_CK,
                       bf16_nt(out_channels), total);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_conv3x3_bf16_forward(const kbn_conv_src* srcs, int n_src, const void* packed_weight, float* out,
                             long long out_batch_stride, int n, int out_channels, int height, int width, int mode,
                             int apply_activation, float negative_slope, kbn_stream_t stream) {
    using namespace kbn;
    if (!srcs || n_src < 1 || n_src > 2 || !packed_weight || !out || n < 1 || out_channels < 1 || height < 1 || width < 1)
        return KBN_ERR_INVALID_ARGUMENT;
    if (mode < 0 || mode > 2) return KBN_ERR_INVALID_ARGUMENT;
    if ((width & 3) || (reinterpret_cast<uintptr_t>(out) & 15) || (out_batch_stride & 3)) return KBN_ERR_UNSUPPORTED;
    if (mode == 1 && (n_src != 1 || (height & 1) || (width & 1))) return KBN_ERR_UNSUPPORTED;
    Bf16ConvParams p{};
    int cin = 0;
    for (int s = 0; s < n_src; ++s) {
        const kbn_conv_src& a = srcs[s];
        if (a.kind != KBN_SRC_TENSOR || !a.data || a.channels < 1 || (a.channels % BF_CK) != 0) return KBN_ERR_UNSUPPORTED;
        if (s == 0) { p.sH = a.src_height; p.sW = a.src_width; }
        if (a.src_height != p.sH || a.src_width != p.sW) return KBN_ERR_INVALID_ARGUMENT;
        p.src[s] = a.data; p.src_bstride[s] = a.batch_stride; p.srcC[s] = a.channels;
        cin += a.channels;
    }
    // source planes vs output size: plain = same, up-conv = half, stride 2 = ceil(in / 2) = out
    const bool dims_ok = mode == 0 ? (p.sH == height && p.sW == width)
                       : mode == 1 ? (2 * p.sH == height && 2 * p.sW == width)
                                   : (ceil_div(p.sH, 2) == height && ceil_div(p.sW, 2) == width);
    if (!dims_ok) return KBN_ERR_INVALID_ARGUMENT;
    if ((long long)p.sH * p.sW > 0x1fffffffLL || (long long)height * width > 0x1fffffffLL) return KBN_ERR_UNSUPPORTED;
    if (n_src == 1) { p.src[1] = p.src[0]; p.src_bstride[1] = p.src_bstride[0]; p.srcC[1] = 0; }
    p.nsrc = n_src;
    p.wp = static_cast<const unsigned short*>(packed_weight);
    p.out = out; p.out_bstride = out_batch_stride;
    p.N = n; p.OC = out_channels; p.Cin = cin; p.H = height; p.W = width; p.up2x = mode == 1;
    const int ntf = bf16_nt(out_channels);
    const int th = mode == 2 ? BfGeom<2>::TH : BfGeom<0>::TH;
    p.tilesX = ceil_div(width, BF_TW); p.tilesY = ceil_div(height, th); p.nTilesN = ceil_div(out_channels, ntf);
    const long long blocks = (long long)p.tilesX * p.tilesY * n * p.nTilesN;
    if (blocks > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.nblocks = (int)blocks;
    p.act = apply_activation ? 1 : 0; p.slope = negative_slope;
    auto launch = [&](auto kern, size_t lds, DeviceOnce& once) -> int {
        if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
        hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), lds, (hipStream_t)stream, p);
        return KBN_OK;
    };
    static DeviceOnce o[6];
    int rc;
    switch (mode * 2 + (ntf == 32 ? 0 : 1)) {
        case 0: rc = launch(conv3x3_bf16_kernel<1, 0>, 2 * BfStage<0, 1>::BYTES, o[0]); break;
        case 1: rc = launch(conv3x3_bf16_kernel<2, 0>, 2 * BfStage<0, 2>::BYTES, o[1]); break;
        case 2: rc = launch(conv3x3_bf16_kernel<1, 1>, 2 * BfStage<1, 1>::BYTES, o[2]); break;
        case 3: rc = launch(conv3x3_bf16_kernel<2, 1>, 2 * BfStage<1, 2>::BYTES, o[3]); break;
        case 4: rc = launch(conv3x3_bf16_kernel<1, 2>, 2 * BfStage<2, 1>::BYTES, o[4]); break;
        default: rc = launch(conv3x3_bf16_kernel<2, 2>, 2 * BfStage<2, 2>::BYTES, o[5]); break;
    }
    if (rc != KBN_OK) return rc;
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

}  // extern "C"
