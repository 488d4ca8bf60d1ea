// tail.hip -- the decoder's tail in one launch, its conv on the 16-bit matrix core:
//     features = act(conv3x3(x))        DecoderBlock deconv0's second conv, no skip     reference src/net_utils.py:1485-1487
//     logits   = conv3x3(features)      MultiScaleDecoder.output0 (one filter, linear)  reference src/networks.py:1985
//     depth    = d_min / (sigmoid(logits) + d_min / d_max)                              reference src/kbnet_model.py:181-184
// Same fusion as conv_head_kernel (csrc/head.hip; the full-resolution feature tensor between the two convs never leaves the
// CU), but the C -> C conv (C <= 16; KBNet: 12) takes its fp32 products as three fp16 MFMAs over two-term splits of both
// operands like csrc/conv_split.hip / csrc/front.hip instead of running on the fp32 MFMAs: at 12 x 12 x 9 multiply-adds per
// pixel of a full-resolution map the fp32 pipe was what bound the launch (conv_head_kernel: 0.60 busy, 737 us per 32 KITTI
// frames for 35 GFLOP).  The fp16 window of the input follows the data tile by tile (max |x| over the pixels the workgroup
// loads); the features stay fp32 in LDS and the one-filter head + sigmoid mapping run on the vector ALU as before.
//
// Tile = 16 x 32 output pixels per workgroup of 8 waves; v_mfma_f32_16x16x32_f16, D[filter][pixel]: A operand = weights
// (16 filter rows, C used), B operand = pixels, K = (tap, 8-channel group): 18 groups in five K steps.
//   A  the 20 x 36 input pixels: C loads per pixel, split, two 16-byte granules [8 channels] per pixel and split term
//   B  features on the 18 x 34 pixels the head reads (39 blocks of 16): LeakyReLU, zero outside the image (the head's padding),
//      fp32 planes in LDS
//   C  one output pixel per thread: 9 C multiply-adds from LDS, sigmoid mapping, store
#include <math.h>

#include "conv_common.h"

namespace kbn {

typedef _Float16 th8 __attribute__((ext_vector_type(8)));
typedef _Float16 th2 __attribute__((ext_vector_type(2)));
typedef float tf4 __attribute__((ext_vector_type(4)));

constexpr int TL_TH = 16, TL_TW = 32, TL_THREADS = 512;
constexpr int TL_R1H = TL_TH + 2, TL_R1W = TL_TW + 2, TL_NP1 = TL_R1H * TL_R1W;      // features the head reads: 18 x 34
constexpr int TL_R0H = TL_R1H + 2, TL_R0W = TL_R1W + 2, TL_NP0 = TL_R0H * TL_R0W;    // inputs they read: 20 x 36
// feature rows padded to 36 floats (16-byte aligned quads); the plane pitch 18 x 36 = 648 is padded to 656 floats = 164 granules, 4 (mod 16):
// the head's ds_read_b128 lane groups mix the four channel lanes of four quads -- planes 4 granules apart make the 16 of a group distinct --
// and the conv's ds_write_b32 (channels 4 kq + r, 32 lanes per group) lands 16 banks apart (648: SQ_LDS_BANK_CONFLICT / IDX_ACTIVE 0.32)
#ifdef KBN_LDS_PITCH_OLD
constexpr int TL_FW = 36, TL_FP = TL_R1H * TL_FW;
#else
constexpr int TL_FW = 36, TL_FP = TL_R1H * TL_FW + 8;
#endif
constexpr int TL_NB = 39, TL_NBLK = 5;                                                 // 16-pixel blocks of the 18 x 34 region; per wave
constexpr int TL_WEXP = 13;
constexpr int TL_TAB = 16;                                                             // floats: 2^-e per filter

struct TailParams {
    const float* x;
    long long x_bstride;
    const _Float16* xp;      // the input as a PAIR tensor of 16 channels (include/kbnet_hip.h), or null (then `x`)
    long long xp_bstride;    // fp16 elements per frame
    const float* xscale;     // its per-frame 2^k
    const float* tab;        // TL_TAB floats
    const _Float16* wp;      // [5 k-steps][term][4 k-groups][16 filters][8]: k-group g = 4 ks + kq = (tap g >> 1, channels 8 (g & 1) + j)
    const float* wout;       // 1 x C x 3 x 3 (raw)
    float* depth;
    float* logits;           // or null
    int N, C, H, W, tilesX, tilesY, ntiles;
    float slope, dmin, ratio;
};

__device__ __forceinline__ void tl_split8(const float (&v)[8], float pre, th8& h1, th8& h2) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const f32x2 a = (f32x2){v[k], v[k + 1]} * pre;
        const th2 c1 = __builtin_convertvector(a, th2);
        const f32x2 f = {(float)c1[0], (float)c1[1]};
        const f32x2 hi = a * 2048.f;
        const f32x2 r = {__builtin_fmaf(f[0], -2048.f, hi[0]), __builtin_fmaf(f[1], -2048.f, hi[1])};
        const th2 c2 = __builtin_convertvector(r, th2);
        h1[k] = c1[0]; h1[k + 1] = c1[1];
        h2[k] = c2[0]; h2[k + 1] = c2[1];
    }
}

// PIN: the input is the up-conv's PAIR tensor (16 channels: two k-groups): stage A is 6 LDS-DMAs per wave -- no loads into
// registers, no tile maximum, no splitting; the window is the producer's
// ONE: the THROUGHPUT-ONLY one-term mode (KBN_FP16_ONE_TERM=1, BASELINE configs[2]'s 16-bit leg): h1 w1 alone -- plain fp16 operands,
// fp32 accumulation, a third of the MFMAs; the h2 planes of a pair input are neither fetched nor read
template <bool PIN, bool ONE = false>
__global__ __launch_bounds__(TL_THREADS, 2) void conv_tail_kernel(const TailParams p) {
    constexpr int IN_KG = TL_NP0 * 16, IN_PART = 2 * IN_KG, IN_BYTES = 2 * IN_PART;   // [term][k-group][pixel][8 ch] fp16
    constexpr int OFF_F = IN_BYTES;                                                    // [C][TL_FP] fp32
    static_assert(OFF_F + 12 * TL_FP * 4 + 12 * 9 * 4 <= 80 * 1024, "two workgroups per CU at KBNet's 12 channels");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0");   // fp16 results flush subnormals (see conv3x3_split_kernel)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    int bid = xcd_remap(blockIdx.x, p.ntiles);
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int oy0 = ty * TL_TH, ox0 = tx * TL_TW;
    const int H = p.H, W = p.W, C = p.C;
    const long long plane = (long long)H * W;
    float* const F = reinterpret_cast<float*>(smem + OFF_F);
    // output0's weights (C x 9) behind the feature planes: the head reads them per lane (channel = lane & 3 + 4 i) -- from global
    // memory that was a latency chain of 27 vector loads inside the channel loop (hoisting them cost the registers that keep two
    // workgroups per CU: round 5); from LDS they cost a broadcast read each (489 -> 467 us per 32 KITTI frames, same box: profiles/r06/v83_ab_tail_wout_lds.txt)
    float* const WO = F + C * TL_FP;
    if (tid < C * 9) WO[tid] = p.wout[tid];   // published by the barrier that ends stage A
    // the conv's A fragments (weights, 10 KB shared by every tile): requested FIRST, so that their L2 / L1 latency runs under the
    // tile's input DMA instead of behind the barrier that ends it (hipcc otherwise sinks the loads to their first use)
    th8 a1[5], a2[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        a1[ks] = *reinterpret_cast<const th8*>(p.wp + (ks * 2 + 0) * 512 + lane * 8);
        a2[ks] = *reinterpret_cast<const th8*>(p.wp + (ks * 2 + 1) * 512 + lane * 8);
    }
    asm volatile("" ::: "memory");   // keeps the requests above the DMA issue below

    // ---- A: input tile -> split granules; the fp16 window is the tile's own (max |x| over the pixels loaded here)
    float un_in;
    if constexpr (PIN) {
        constexpr int NR = (TL_NP0 + 63) / 64, NDMA = (ONE ? 2 : 4) * NR, DPW = NDMA / 8;   // 4 planes (term, k-group; ONE: the two h1 planes) x 12 rounds of 64 pixels
        static_assert(NDMA % 8 == 0, "whole rounds of the eight waves");
        un_in = 1.f / p.xscale[n];
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(reinterpret_cast<const float*>(smem)));
        const long long pplane = pair_plane_halves(H, W);
        const _Float16* pn = p.xp + (long long)n * p.xp_bstride;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int id = wave + 8 * i, pl = id / NR, j = id - pl * NR;
            const int t = pl >> 1, kg = pl & 1;
            const int pix = j * 64 + lane;
            const int r = pix / TL_R0W, c = pix - r * TL_R0W;
            const int Y = oy0 - 2 + r, X = ox0 - 2 + c;
            const unsigned voff = (pix < TL_NP0 && Y >= 0 && Y < H && X >= 0 && X < W) ? (unsigned)(Y * W + X) * 16u : (unsigned)(H * W) * 16u;
            const unsigned long long mask = (j == NR - 1 && (TL_NP0 & 63)) ? ((1ull << (TL_NP0 & 63)) - 1) : ~0ull;
            lds_dma16_sm(reinterpret_cast<const float*>(pn + (long long)(kg * 2 + t) * pplane), voff,
                         lds0 + (unsigned)(t * IN_PART + kg * IN_KG + j * 64 * 16), mask);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        const float* xn = p.x + (long long)n * p.x_bstride;
        float raw[2][16];
        float tm = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pix = u * TL_THREADS + tid;
            const int r = pix / TL_R0W, c = pix - r * TL_R0W;
            const int Y = oy0 - 2 + r, X = ox0 - 2 + c;
            const bool ok = pix < TL_NP0 && Y >= 0 && Y < H && X >= 0 && X < W;
            const float* src = xn + (long long)(ok ? Y : 0) * W + (ok ? X : 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                raw[u][j] = (ok && j < C) ? src[(long long)(j < C ? j : 0) * plane] : 0.f;
                tm = fmaxf(tm, fabsf(raw[u][j]));
            }
        }
        tm = __uint_as_float(wave_max_bits(tm));
        if (lane == 0) F[wave] = tm;   // the feature planes are idle until the conv writes them (two barriers on)
        __syncthreads();
        tm = fmaxf(fmaxf(fmaxf(F[0], F[1]), fmaxf(F[2], F[3])), fmaxf(fmaxf(F[4], F[5]), fmaxf(F[6], F[7])));
        const unsigned abits = __builtin_amdgcn_readfirstlane(__float_as_uint(tm));
        int k = 14 + 127 - (int)(abits >> 23);
        k = k > 100 ? 100 : (k < -100 ? -100 : k);
        const float pre_in = __uint_as_float((unsigned)(127 + k) << 23);
        un_in = __uint_as_float((unsigned)(127 - k) << 23);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pix = u * TL_THREADS + tid;
            if (pix < TL_NP0) {
#pragma unroll
                for (int kg = 0; kg < 2; ++kg) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = raw[u][8 * kg + j];
                    th8 h1, h2;
                    tl_split8(v, pre_in, h1, h2);
                    *reinterpret_cast<th8*>(smem + kg * IN_KG + pix * 16) = h1;
                    *reinterpret_cast<th8*>(smem + IN_PART + kg * IN_KG + pix * 16) = h2;
                }
            }
        }
    }

    // ---- per-lane offsets
    int tapoff[5];   // k-group g = 4 ks + kq: tap g >> 1, channel group g & 1; groups 18, 19 carry zero weights (any valid address)
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int g = min(4 * ks + kq, 17), tap = g >> 1;
        tapoff[ks] = (g & 1) * IN_KG + ((tap / 3) * TL_R0W + tap % 3) * 16;
    }
    // block j of the 18 x 34 region: j < 36: row j >> 1, columns 16 (j & 1) + l15; 36 / 37: column 32 / 33, rows l15; 38: the
    // four pixels of rows 16, 17 x columns 32, 33.  Wave w takes blocks w, w + 8, ..
    int inoff[TL_NBLK], foff[TL_NBLK];
    unsigned inside = 0, valid = 0;
#pragma unroll
    for (int i = 0; i < TL_NBLK; ++i) {
        const int j = wave + 8 * i;
        int r1, c1;
        bool ok = true;
        if (j < 36) { r1 = j >> 1; c1 = 16 * (j & 1) + l15; }
        else if (j < 38) { r1 = l15; c1 = 32 + (j - 36); }
        else { r1 = 16 + (l15 >> 1); c1 = 32 + (l15 & 1); ok = l15 < 4; }
        if (!ok) { r1 = 0; c1 = 0; }
        const int Y = oy0 - 1 + r1, X = ox0 - 1 + c1;
        inoff[i] = (r1 * TL_R0W + c1) * 16;
        foff[i] = r1 * TL_FW + c1;
        if (ok && j < TL_NB) valid |= 1u << i;
        if (ok && j < TL_NB && Y >= 0 && Y < H && X >= 0 && X < W) inside |= 1u << i;
    }
    const int nblk = wave + 8 * (TL_NBLK - 1) < TL_NB ? TL_NBLK : TL_NBLK - 1;   // wave-uniform
    __syncthreads();   // IN complete (and every wave has read the reduction scratch)

    // ---- B: the C -> C conv on the 612 pixels the head reads
    {
        tf4 sc = *reinterpret_cast<const tf4*>(p.tab + 4 * kq);   // 2^-e of this lane's four filters
        sc *= un_in;
        const f32x2 sc01 = {sc[0], sc[1]}, sc23 = {sc[2], sc[3]};
#pragma unroll
        for (int i = 0; i < TL_NBLK; ++i) {
            if (i < nblk) {
                const unsigned char* inb = smem + inoff[i];
                tf4 m = (tf4){0.f, 0.f, 0.f, 0.f}, s = m;
                // the block's ten fragment reads go out together: one LDS latency per block instead of one per K step
                // the block's ten fragment reads go out together (one LDS latency per block instead of one per K step): with the weights
                // requested first, 486 -> 474 us per 32 KITTI frames inside the forward (round 5; output0's 27 weights per lane, fetched in
                // front of the second barrier instead of inside the channel loop: 604 us; at the top of the kernel: 27 more live registers
                // cross the 128 that keep two workgroups per CU, 884 us -- both measured, both dropped)
                th8 b1[5], b2[5];
#pragma unroll
                for (int ks = 0; ks < 5; ++ks) {
                    b1[ks] = *reinterpret_cast<const th8*>(inb + tapoff[ks]);
                    if constexpr (!ONE) b2[ks] = *reinterpret_cast<const th8*>(inb + IN_PART + tapoff[ks]);
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int ks = 0; ks < 5; ++ks) {
                    m = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ks], b1[ks], m, 0, 0, 0);
                    if constexpr (!ONE) {
                        s = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[ks], b1[ks], s, 0, 0, 0);
                        s = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ks], b2[ks], s, 0, 0, 0);
                    }
                }
                f32x2 t01 = (f32x2){s[0], s[1]} * 0.00048828125f + (f32x2){m[0], m[1]};
                f32x2 t23 = (f32x2){s[2], s[3]} * 0.00048828125f + (f32x2){m[2], m[3]};
                t01 *= sc01; t23 *= sc23;
                const f32x2 u01 = t01 * p.slope, u23 = t23 * p.slope;   // LeakyReLU as max(t, slope t), 0 <= slope <= 1 (1: linear)
                tf4 v = {fmaxf(t01[0], u01[0]), fmaxf(t01[1], u01[1]), fmaxf(t23[0], u23[0]), fmaxf(t23[1], u23[1])};
                if (!((inside >> i) & 1)) v = (tf4){0.f, 0.f, 0.f, 0.f};   // outside the image: the zero padding of output0
                if ((valid >> i) & 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * kq + r < C) F[(4 * kq + r) * TL_FP + foff[i]] = v[r];
                }
            }
        }
    }
    // output0's weights of this lane's channels (t, t + 4, t + 8 of at most 12): requested before the barrier
    __syncthreads();

    // ---- C: output0 + sigmoid mapping.  Four lanes share a quad of 4 consecutive output pixels, each summing a third /
    // quarter of the channels (a 3 x 6 window per channel: one 16-byte + one 8-byte LDS read per row), then two DPP moves.
    {
        const int t = lane & 3, quad = tid >> 2;           // 128 quads: row quad >> 3, columns 4 (quad & 7) ..
        const int y = quad >> 3, x0 = 4 * (quad & 7);
        tf4 acc = (tf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const int c = t + 4 * ci;
            if (c < C) {
                const float* fc = F + c * TL_FP + y * TL_FW + x0;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const tf4 a = *reinterpret_cast<const tf4*>(fc + ky * TL_FW);
                    const f32x2 b = *reinterpret_cast<const f32x2*>(fc + ky * TL_FW + 4);
                    const float win[6] = {a[0], a[1], a[2], a[3], b[0], b[1]};
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float w = WO[c * 9 + ky * 3 + kx];
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(w, win[i + kx], acc[i]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = quad_sum(acc[i]);
        const int Y = oy0 + y, X = ox0 + x0 + t;           // lane t of the quad stores pixel t
        const float a = t == 0 ? acc[0] : (t == 1 ? acc[1] : (t == 2 ? acc[2] : acc[3]));
        if (Y < H && X < W) {
            const long long o = (long long)n * plane + (long long)Y * W + X;
            if (p.logits) p.logits[o] = a;
            const float sg = 1.0f / (1.0f + expf(-a));
            p.depth[o] = p.dmin / (sg + p.ratio);
        }
    }
}

__global__ void tail_pack_kernel(const float* __restrict__ w, float* __restrict__ tab, _Float16* __restrict__ out, int C) {
    __shared__ float inv[16];
    if (threadIdx.x < 16) {
        const int f = threadIdx.x;
        float m = 0.f;
        if (f < C)
            for (int i = 0; i < C * 9; ++i) m = fmaxf(m, fabsf(w[(long long)f * C * 9 + i]));
        int ex = TL_WEXP;
        if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &ex);
        int e = TL_WEXP - ex;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        inv[f] = ldexpf(1.f, -e);
        tab[f] = inv[f];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 5 * 2 * 64 * 8; e += blockDim.x) {   // [ks][term][kq][m 16][j 8]
        int r = e;
        const int j = r & 7; r >>= 3;
        const int m = r & 15; r >>= 4;
        const int kq = r & 3; r >>= 2;
        const int term = r & 1; r >>= 1;
        const int g = 4 * r + kq, tap = g >> 1, ch = 8 * (g & 1) + j;
        _Float16 h = (_Float16)0.f;
        if (g < 18 && ch < C && m < C) {
            const float ws = w[((long long)m * C + ch) * 9 + tap] / inv[m];
            const _Float16 w1 = (_Float16)ws;
            h = term == 0 ? w1 : (_Float16)((ws - (float)w1) * 2048.f);
        }
        out[e] = h;
    }
}

}  // namespace kbn

extern "C" {

size_t kbn_conv_tail_packed_weight_bytes(int channels) {
    if (channels < 1 || channels > 16) return 0;
    return (size_t)kbn::TL_TAB * 4 + 2 * (5 * 2 * 64 * 8);
}

int kbn_conv_tail_pack_weight(const float* w_conv, void* packed, int channels, kbn_stream_t stream) {
    using namespace kbn;
    if (!w_conv || !packed) return KBN_ERR_INVALID_ARGUMENT;
    if (channels < 1 || channels > 16) return KBN_ERR_UNSUPPORTED;
    float* tab = static_cast<float*>(packed);
    hipLaunchKernelGGL(tail_pack_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w_conv, tab,
                       reinterpret_cast<_Float16*>(tab + TL_TAB), channels);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

static int conv_tail_launch(const float* x, long long x_batch_stride, const void* x_pair, long long x_pair_batch_stride,
                            const float* x_pair_scale, const void* packed_w_conv, const float* w_out, float* depth,
                            float* logits, int n, int channels, int height, int width, int apply_activation, float negative_slope,
                            float min_predict_depth, float max_predict_depth, kbn_stream_t stream) {
    using namespace kbn;
    if ((!x && !x_pair) || !packed_w_conv || !w_out || !depth || n < 1 || channels < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (x_pair && (!x_pair_scale || (reinterpret_cast<uintptr_t>(x_pair) & 15) || (x_pair_batch_stride & 7) ||
                   x_pair_batch_stride < 2 * 2 * pair_plane_halves(height, width)))
        return KBN_ERR_INVALID_ARGUMENT;
    if (channels > 12 || knob(KNOB_NO_SPLIT) || knob(KNOB_NO_HEAD_FUSION)) return KBN_ERR_UNSUPPORTED;   // LDS: 12 feature planes beside the input tile
    if ((long long)height * width > 0x1fffffffLL) return KBN_ERR_UNSUPPORTED;
    if (x_pair && (long long)height * width >= 0x0fffffffLL) return KBN_ERR_UNSUPPORTED;   // 32-bit DMA offsets of 16-byte granules (as kbn_conv3x3_split_forward)
    if (apply_activation && !(negative_slope >= 0.f && negative_slope <= 1.f)) return KBN_ERR_UNSUPPORTED;   // max(v, slope v) form
    TailParams p{};
    p.x = x; p.x_bstride = x_batch_stride;
    p.xp = static_cast<const _Float16*>(x_pair); p.xp_bstride = x_pair_batch_stride; p.xscale = x_pair_scale;
    p.tab = static_cast<const float*>(packed_w_conv);
    p.wp = reinterpret_cast<const _Float16*>(p.tab + TL_TAB);
    p.wout = w_out; p.depth = depth; p.logits = logits;
    p.N = n; p.C = channels; p.H = height; p.W = width;
    p.tilesX = ceil_div(width, TL_TW); p.tilesY = ceil_div(height, TL_TH);
    const long long tiles = (long long)p.tilesX * p.tilesY * n;
    if (tiles > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.ntiles = (int)tiles;
    p.slope = apply_activation ? negative_slope : 1.f;
    p.dmin = min_predict_depth;
    p.ratio = (float)((double)min_predict_depth / (double)max_predict_depth);   // evaluated in double like the reference's scalar
    const size_t lds = (size_t)2 * 2 * TL_NP0 * 16 + (size_t)channels * TL_FP * 4 + (size_t)channels * 9 * 4;
    static DeviceOnce once, oncep, once1, oncep1;
    const bool one_term = knob(KNOB_FP16_ONE_TERM) != 0;   // THROUGHPUT-ONLY: h1 w1 alone
    auto go = [&](auto kern, DeviceOnce& o) -> int {
        if (int rc = set_max_dynamic_lds(o, reinterpret_cast<const void*>(kern), 80 * 1024)) return rc;
        hipLaunchKernelGGL(kern, dim3(p.ntiles), dim3(TL_THREADS), lds, (hipStream_t)stream, p);
        return KBN_OK;
    };
    if (int rc = x_pair ? (one_term ? go(conv_tail_kernel<true, true>, oncep1) : go(conv_tail_kernel<true, false>, oncep))
                        : (one_term ? go(conv_tail_kernel<false, true>, once1) : go(conv_tail_kernel<false, false>, once)))
        return rc;
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_conv_tail_forward(const float* x, long long x_batch_stride, const void* packed_w_conv, const float* w_out, float* depth,
                          float* logits, int n, int channels, int height, int width, int apply_activation, float negative_slope,
                          float min_predict_depth, float max_predict_depth, kbn_stream_t stream) {
    if (!x) return KBN_ERR_INVALID_ARGUMENT;
    return conv_tail_launch(x, x_batch_stride, nullptr, 0, nullptr, packed_w_conv, w_out, depth, logits, n, channels, height, width,
                            apply_activation, negative_slope, min_predict_depth, max_predict_depth, stream);
}

int kbn_conv_tail_forward_pair(const void* x_pair, long long x_pair_batch_stride, const float* x_pair_scale, const void* packed_w_conv,
                               const float* w_out, float* depth, float* logits, int n, int channels, int height, int width,
                               int apply_activation, float negative_slope, float min_predict_depth, float max_predict_depth,
                               kbn_stream_t stream) {
    if (!x_pair) return KBN_ERR_INVALID_ARGUMENT;
    return conv_tail_launch(nullptr, 0, x_pair, x_pair_batch_stride, x_pair_scale, packed_w_conv, w_out, depth, logits, n, channels,
                            height, width, apply_activation, negative_slope, min_predict_depth, max_predict_depth, stream);
}

}  // extern "C"
