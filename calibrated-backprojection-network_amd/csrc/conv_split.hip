// conv_split.hip -- 3x3 convs with fp32-grade results on the 16-bit matrix core (operands split in two fp16 terms).
//
// gfx950 runs v_mfma_f32_*_f32 on the fp32 VECTOR datapath (157 TFLOP/s, issue shared with every other vector
// instruction: tools/probe/valu_probe.hip); the matrix core proper takes 16-bit and narrower operands (2.5 PFLOP/s dense)
// and runs beside the vector ALU (tools/probe/bf16x_probe.hip: an MFMA wave keeps 32 clk per MFMA with a v_fma wave on
// the same SIMD).  This kernel feeds it fp32 operands as pairs of fp16 values:
//     activation  a 2^k    = h1 + 2^-11 h2,   h1 = fp16(a 2^k),   h2 = fp16((a 2^k - h1) 2^11)       (22+ bits of a)
//     weight      w 2^e    = w1 + w2,         w1 = fp16(w 2^e),   w2 = fp16(w 2^e - w1)              (22+ bits of w)
//     a w 2^(e+k) = h1 w1 + h1 w2 + h2 (w1 2^-11)   (+ h2 w2 2^-11, below 2^-22 |a w|, dropped)
//                 = h1 w1 + 2^-11 (h1 (w2 2^11) + h2 w1): the 3x3 kernels keep w2 scaled by 2^11 as well (normal in fp16
//                   down to |w2| = 2^-25) and sum the two small terms, both 2^11 times their share, in an accumulator of
//                   their own that enters the result once, in the epilogue -- no per-tap scaling of w1 in the K loop
// THREE fp16 MFMAs with fp32 accumulation per product, 3/16 of the fp32 MFMA's time.  The residual h2 is kept SCALED by
// 2^11 so that it sits in fp16's normal range whenever h1 does (the matrix core flushes fp16 subnormals); e is chosen
// per filter at pack time (largest |w 2^e| in [2^12, 2^13)); k (`act_exponent`) is the caller's: it places the fp16
// window on the layer's activations -- |a| 2^k up to 65504 is finite, |a| 2^k >= 2^-14 has the full 22 bits, smaller
// activations (h1 subnormal -> flushed, the mode register is set so) are carried by h2 alone with 11 bits.  There is
// no calibration and no state: every producer folds max |out| per frame into a device-side slot (kbn_conv_src.absmax,
// ops.ActStats) and the consumer derives k = 14 - floor(log2 max) per frame inside the forward (sp_act_scale below: the
// frame's maximum lands in [2^14, 2^15) of the window); a source without a slot takes the ABI default -6, which covers
// 0.0039 .. 4.2e6.  Measured against an fp64 evaluation (profiles/r02/bf16x_probe.txt, K = 576 .. 6912): rms error 0.28e-6 .. 0.9e-6
// of the output's rms, the fp32 MFMA chain (== fmaf chain) 0.44e-6 .. 1.7e-6 -- the accuracy class of the fp32 path, which
// is why this kernel sits on the parity-gated path (tests/test_hip_parity.py holds it to the same 1e-4 bar).
//
// Direct 3x3 conv (+ LeakyReLU) over one or two NCHW fp32 sources -- MODE 0: the concat convs of the decoder, reference
// src/net_utils.py:1483-1487; MODE 1: nearest-2x up-conv with nine taps, :484-499 (no longer instantiated since round 6: superseded by the folded form at the
// end of this file); MODE 2: stride 2, the image convs of the KB blocks, :1348 -- as an implicit GEMM with M = 32 output
// pixels of a row, N = 32 filters, K = 16 channels per v_mfma_f32_32x32x16_f16.  Workgroup = 512 threads = 8 waves = RG
// row groups x 8/RG filter groups; a wave owns MB rows (m-blocks) x two 32-filter n-blocks and keeps TWO accumulators
// per block: the main term h1 w1, and the two small terms (2^-11 of it) apart, so that the main accumulator is rounded
// once per 16-channel step.  Stride 1: 8 x 1 waves, tile 16 rows x 32 pixels x 64 filters; stride 2: 4 x 2 waves, tile
// 8 x 32 x 128.  K loop over chunks of 16 channels, 9 taps each:
//   A  the (16+2) x (32+2) input pixels of the chunk (stride 2: 17 x 65, columns de-interleaved; up-conv: the
//      (8+2) x (16+2) low-resolution pixels, the fragment reads do the upsampling), split on the way into LDS: a thread
//      loads 8 channels of a pixel (scalar plane base + lane offset), splits them (v_cvt_pk_f16_f32, v_cvt_f32_f16,
//      v_pk_fma_f32) and writes two 16-byte words, layout [part][k-group][pixel][8 channels]: an MFMA A fragment is one
//      ds_read_b128, fetched one m-block ahead of its MFMAs.  Double buffered: the global loads of chunk c+1 are in
//      flight under the MFMAs of chunk c, their split + LDS writes are spread over the MFMA groups of the later taps.
//   B  weights pre-split at pack time, [chunk][tap][part][k-group][filter][8 channels] fp16.  MODE 0: the nine taps of
//      a chunk are copied into LDS by LDS-DMA (double buffered): every global access of chunk c+1 is issued at the
//      start of chunk c and awaited once, late in it -- no vmcnt wait between MFMAs.  MODE 2 (its input tile leaves no
//      LDS for that): every wave loads its B fragments straight from global memory, one tap ahead (the row-group waves
//      of a filter group hit the same lines in L1).
// ONE barrier per chunk.  What bounds it: the chip's power limit (the MFMAs alone: 77 % of the launch at 1.7-2.0 GHz;
// zero-filled operands run 25 % faster through the same instruction stream), then the part of the skeleton that does
// not hide under them (DESIGN.md section 4 has the ablation).
#include "conv_common.h"

// 1: tiles without padding (no output rows below the map, no 32-filter blocks past the last filter) run a K loop whose
// MFMAs carry no tests at all; only the other tiles take the loop with a wave-uniform test in front of every MFMA (which
// puts each MFMA in a basic block of its own).  0: every tile takes the tested loop (A/B builds).
#ifndef KBN_SPLIT_STRAIGHT
#define KBN_SPLIT_STRAIGHT 1
#endif
// A/B builds (KBN_HIPCC_FLAGS=-DKBN_SPLIT_PRIO=n): 1 s_setprio(1) around every MFMA group of the concat kernel, 2 once for
// waves 4-7.  Measured inside the forward (four concat convs, 32 KITTI frames): 4168-4312 us without, 4334-4360 with 1, 4165 with 2:
// the waves of this kernel move in lockstep, there is nothing for the arbiter to prefer.  Off.
#ifndef KBN_SPLIT_PRIO
#define KBN_SPLIT_PRIO 0
#endif

namespace kbn {

typedef _Float16 sph8 __attribute__((ext_vector_type(8)));
typedef _Float16 sph2 __attribute__((ext_vector_type(2)));
typedef float spf16 __attribute__((ext_vector_type(16)));
typedef float spf4 __attribute__((ext_vector_type(4)));
typedef _Float16 sph4 __attribute__((ext_vector_type(4)));

constexpr int SP_TW = 32, SP_CK = 16, SP_MB = 4, SP_TH = 16, SP_THREADS = 512;
constexpr int SP_WEXP = 13;                // largest |w 2^e| of a filter in [2^12, 2^13)

template <int MODE>   // 0 plain 3x3, 1 nearest-2x up-conv, 2 stride-2 conv
struct SpGeom {
    static constexpr bool UP = MODE == 1, S2 = MODE == 2;
    static constexpr int TH = S2 ? 8 : SP_TH;                            // output rows per workgroup
    static constexpr int ROWS = UP ? TH / 2 + 2 : (S2 ? 2 * TH + 1 : TH + 2);
    static constexpr int COLS = UP ? SP_TW / 2 + 2 : (S2 ? 2 * SP_TW + 1 : SP_TW + 2), NPIX = ROWS * COLS;
    static constexpr int A_PART = 2 * NPIX * 16, A_BYTES = 2 * A_PART;   // [part][k-group][pixel][8 fp16]
    static constexpr int LDS = 2 * A_BYTES;
    static constexpr int PR = (NPIX + 255) / 256;                        // staging rounds of a 256-thread half (one k-group each)
    static constexpr int NLOADA = PR * 8;                                // vector-memory loads per chunk (inputs)
};

struct SplitConvParams {
    const float* src[2];
    long long src_bstride[2];
    int srcC[2];
    int nsrc;
    const float* inv_scale;     // per filter: 2^-e
    const _Float16* wp;         // [n-tile][chunk][tap][part][k-group][NT filters][8 channels] fp16
    float* out;
    long long out_bstride;
    int N, OC, Cin, H, W;       // output size
    int sH, sW;                 // source planes: H x W, (H/2) x (W/2) for the up-conv, the input size of a stride-2 conv
    int tilesX, tilesY, nTilesN, nblocks;
    int act;
    float slope;
    int vec4;                   // output rows are 16-byte aligned quads (width % 4 == 0, aligned base and strides)
    float prescale;             // 2^k on the activations (k = act_exponent of the launch): used when amax[0] is null
    float unscale;              // 2^-k
    const unsigned* amax[2];    // per-frame max |a| slots of the sources (kbn_common.h): k follows the data, frame by frame
    unsigned* out_amax;         // per-frame max |out| slot of the output, or null
    // producer-written split format ("pair" tensors, see below): source 0 and / or the output as fp16 pairs
    const _Float16* pair_src;   // source 0 in pair format, or null (then src[0] is an fp32 NCHW tensor)
    long long pair_src_bstride; // halves per frame
    const float* pair_src_scale;// per frame: the 2^k its producer applied
    _Float16* pair_out;         // the output in pair format, or null (then `out`)
    long long pair_out_bstride;
    float* pair_out_scale;      // per frame: the 2^k applied here (every workgroup of a frame writes the same value)
    const float* l1;            // per 16-channel chunk: max over filters of sum |w| (the table behind the packed weights)
    int tp_x0, tp_tilesY, tp_nblocks;   // conv3x3_split_mixed_kernel: first column, tile rows and workgroups of the transposed tiles
    int sub0;                   // conv1x1s2_split_kernel: source 0 holds only the pixels the conv reads (H x W planes, stride 1)
    // conv1x1s2_split_kernel: three more input channels taken in fp32 in the epilogue (the KB block's backprojection)
    const float* xyz;           // N x 3 x H x W (output size), or null
    long long xyz_bstride;
    const float* wxyz;          // out_channels x 3 fp32
    // split-K (KSPLIT kernels): workgroups per tile, elements between the partial sums' plane sets (p.out is the workspace then)
    int ksplit;
    long long ks_stride;
};

// two-term split of 8 floats: h1 = fp16(a 2^k), h2 = fp16((a 2^k - h1) 2^11)
__device__ __forceinline__ void sp_split8(const float (&v)[8], float prescale, sph8& h1, sph8& h2) {
    const float prescale_hi = prescale * 2048.f;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const f32x2 a = {v[k], v[k + 1]};
        const sph2 c1 = __builtin_convertvector(a * prescale, sph2);
        const f32x2 f = {(float)c1[0], (float)c1[1]};
        const f32x2 hi = a * prescale_hi;
        const f32x2 r = {__builtin_fmaf(f[0], -2048.f, hi[0]), __builtin_fmaf(f[1], -2048.f, hi[1])};
        const sph2 c2 = __builtin_convertvector(r, sph2);
        h1[k] = c1[0]; h1[k + 1] = c1[1];
        h2[k] = c2[0]; h2[k + 1] = c2[1];
    }
}

// The activation exponent of frame n: with slots on the sources, k = 14 - floor(log2(max |a|)) puts the frame's largest
// activation in [2^14, 2^15) of the fp16 window (65504 is the overflow: a factor 2 to spare for the rounding of h1), so
// |a| 2^k >= 2^-14 -- 29 binades below the maximum -- keeps the full 22 bits and anything smaller is off by less than
// 2^-40 of the maximum.  An all-zero frame (or a denormal maximum) takes k = 100, Inf / NaN maxima k = -100: finite
// scales either way.  Wave-uniform: n comes from blockIdx, the loads are scalar.
__device__ __forceinline__ void sp_act_scale(const SplitConvParams& p, int n, float& prescale, float& unscale) {
    prescale = p.prescale;
    unscale = p.unscale;
    if (p.amax[0]) {   // launch-uniform
        unsigned b = p.amax[0][n];
        if (p.amax[1]) b = max(b, p.amax[1][n]);
        int k = 14 + 127 - (int)(b >> 23);
        k = k > 100 ? 100 : (k < -100 ? -100 : k);
        prescale = __uint_as_float((unsigned)(127 + k) << 23);
        unscale = __uint_as_float((unsigned)(127 - k) << 23);
    }
}
__device__ __forceinline__ float sp_amax4(float m, const f32x4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// ------------------------------------------------------------------------------------------------------------------
// PAIR tensors: the producer-written split format.  A consumer that splits its fp32 inputs itself does so once per
// (input value x halo x filter tile of the consumer) -- 2.4 to 4.8 times per value in the decoder -- on the vector ALU,
// beside the MFMAs it feeds (staging ablation in DESIGN.md: 10-13 % of the decoder kernels).  A producer that knows its
// output is only read by split-operand kernels writes the two fp16 terms itself, once, in the layout the consumers
// stage: per frame [k-group = channel / 8][term h1 | h2][H * W + 1 pixels][8 channels] fp16 -- the same 4 bytes per
// value as fp32 -- and a consumer's staging is then one 16-byte LDS-DMA per pixel, k-group and term, no vector ALU
// work and no staging registers.  The extra granule at the end of every plane is ZERO (written by the producer): the
// per-lane DMA offset of a halo pixel outside the map points there.
//   The 2^k of a pair tensor is fixed by its producer BEFORE it has seen its output: from the bound
// |out| <= sum over sources (max |a_s| of the frame, from the source's slot) x (sum over the source's 16-channel chunks
// of max over filters of sum |w|), the table `l1` behind the packed weights), placed in [2^14, 2^15) like the measured
// maxima of sp_act_scale.  The bound overshoots the true maximum by a few binades (never accumulating over layers: every
// layer starts from the MEASURED maxima of its inputs), and a window up to 2^16 too high costs nothing (the terms keep
// 22 bits down to 2^-29 of the window, tests/test_split_math_cpu.py).  Every workgroup of a frame computes the same k
// and writes it to the tensor's per-frame scale slot; consumers read it there.  With two sources in different formats
// the accumulators are rescaled by the exact power of two between the two windows when the K loop changes source.
// (pair_plane_halves: conv_common.h)

__device__ __forceinline__ float sp_scale_of_bound(float bound) {   // 2^k with bound 2^k in [2^14, 2^15); finite for 0 / Inf / NaN
    int k = 14 + 127 - (int)(__float_as_uint(bound) >> 23 & 255u);
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return __uint_as_float((unsigned)(127 + k) << 23);
}
// the 2^k of this launch's pair output for frame n (wave-uniform: scalar loads)
__device__ __forceinline__ float sp_pair_out_scale(const SplitConvParams& p, int n) {
    const int nchunks = p.Cin / SP_CK, n0 = p.nsrc > 1 ? p.srcC[0] / SP_CK : nchunks;
    float w0 = 0.f, w1 = 0.f;
    for (int c = 0; c < n0; ++c) w0 += p.l1[c];
    for (int c = n0; c < nchunks; ++c) w1 += p.l1[c];
    float bound = __uint_as_float(p.amax[0][n]) * w0;
    if (p.nsrc > 1) bound += __uint_as_float(p.amax[1][n]) * w1;
    return sp_scale_of_bound(bound);
}
// window of ONE source from its slot (the other source of the launch is a pair tensor with a scale of its own)
__device__ __forceinline__ void sp_act_scale_of(const SplitConvParams& p, int s, int n, float& prescale, float& unscale) {
    prescale = p.prescale;
    unscale = p.unscale;
    if (p.amax[s]) {
        int k = 14 + 127 - (int)(p.amax[s][n] >> 23);
        k = k > 100 ? 100 : (k < -100 ? -100 : k);
        prescale = __uint_as_float((unsigned)(127 + k) << 23);
        unscale = __uint_as_float((unsigned)(127 - k) << 23);
    }
}
// Halves of two granules -> one whole granule per lane.  In the pair epilogues lane (pixel, g = lane >> 5) holds channels
// 4 g .. 4 g + 3 of every k-group of its 32 filters; `a` is its piece of k-group q, `b` of k-group q + 1.  One
// v_permlane32_swap per dword hands lanes 0-31 the whole granule q and lanes 32-63 the whole granule q + 1: 16-byte stores.
typedef unsigned spu2 __attribute__((ext_vector_type(2)));
typedef unsigned spu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ spu4 sp_pair_exchange(const sph4& a, const sph4& b) {
    const spu2 A = __builtin_bit_cast(spu2, a), B = __builtin_bit_cast(spu2, b);
    const auto r0 = __builtin_amdgcn_permlane32_swap(A[0], B[0], false, false);   // {A.lo | B.lo , A.hi | B.hi} by lane half
    const auto r1 = __builtin_amdgcn_permlane32_swap(A[1], B[1], false, false);
    return (spu4){r0[0], r1[0], r0[1], r1[1]};
}
// The same between 16-lane rows r and r + 1 (the 16x16x32 epilogue: lane (pixel lp, kq) holds channels 4 (kq & 1) .. of k-group
// kq >> 1): `a` is the lane's half granule of output pixel px = 0, `b` of px = 1; rows with even kq end up with the whole
// granule of px = 0, rows with odd kq with that of px = 1 (v_permlane16_swap: odd rows of the first operand <-> even rows of
// the second).
__device__ __forceinline__ spu4 sp_pair_exchange16(const sph4& a, const sph4& b) {
    const spu2 A = __builtin_bit_cast(spu2, a), B = __builtin_bit_cast(spu2, b);
    const auto r0 = __builtin_amdgcn_permlane16_swap(A[0], B[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(A[1], B[1], false, false);
    return (spu4){r0[0], r1[0], r0[1], r1[1]};
}
// two-term split of 4 floats already in window units (t = a 2^k)
__device__ __forceinline__ void sp_split4(const f32x4& t, sph4& h1, sph4& h2) {
#pragma unroll
    for (int k = 0; k < 4; k += 2) {   // two at a time: packed conversions and packed fp32 arithmetic
        const f32x2 a = {t[k], t[k + 1]};
        const sph2 c1 = __builtin_convertvector(a, sph2);
        const f32x2 f = {(float)c1[0], (float)c1[1]};
        const sph2 c2 = __builtin_convertvector((a - f) * 2048.f, sph2);   // a - f and the scaling are exact
        h1[k] = c1[0]; h1[k + 1] = c1[1];
        h2[k] = c2[0]; h2[k + 1] = c2[1];
    }
}

// pass 1 of the pack: per-filter exponent; inv_scale[oc] = 2^-e
__global__ void split_scale_kernel(const float* __restrict__ w, float* __restrict__ inv_scale, int OC, int per_filter) {
    const int oc = blockIdx.x;
    __shared__ float red[256];
    float m = 0.f;
    if (oc < OC)
        for (int i = threadIdx.x; i < per_filter; i += 256) m = fmaxf(m, fabsf(w[(long long)oc * per_filter + i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int ex = SP_WEXP;
        if (red[0] > 0.f && red[0] < 3.0e38f) (void)frexpf(red[0], &ex);   // red[0] = m 2^ex, m in [0.5, 1)
        int e = SP_WEXP - ex;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        inv_scale[oc] = ldexpf(1.f, -e);
    }
}

// pass 2: OIHW fp32 -> [n-tile][chunk][tap][part][k-group][n][8 k] fp16, zero padded
// `taps` = 9 (3x3) or 1 (1x1); channel c of the packed panel is channel c (c < skip_at) or c + skip of the weight: the
// 1x1 conv of the KB block leaves its three backprojection channels out of the panel (they are taken in fp32)
__global__ void pack_split_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale, _Float16* __restrict__ packed,
                                  int OC, int Cin, int nchunks, int NT, long long total, int taps, int skip_at, int skip) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int per_chunk = taps * 2 * 2 * NT * 8;
    int r = (int)(e % per_chunk);
    const long long q = e / per_chunk;
    const int chunk = (int)(q % nchunks), nt = (int)(q / nchunks);
    const int tap = r / (2 * 2 * NT * 8); r -= tap * 2 * 2 * NT * 8;
    const int part = r / (2 * NT * 8); r -= part * 2 * NT * 8;
    const int g = r / (NT * 8); r -= g * NT * 8;
    const int n = r >> 3, k = r & 7;
    const int c = chunk * SP_CK + g * 8 + k, oc = nt * NT + n;
    _Float16 h = (_Float16)0.f;
    if (c < Cin && oc < OC) {
        const int cw = c < skip_at ? c : c + skip;
        const float ws = w[((long long)oc * (Cin + skip) + cw) * taps + tap] * (1.f / inv_scale[oc]);   // w 2^e, exact
        const _Float16 w1 = (_Float16)ws;
        h = part == 0 ? w1 : (_Float16)((ws - (float)w1) * 2048.f);   // the residual scaled by 2^11 (|.| <= 4096), like the activations'

    }
    packed[e] = h;
}

// max |x| of each of n frames of `per_frame` contiguous floats (frames batch_stride apart) into slots[frame] (integer
// atomic max of the bit patterns).  HBM bound: 16-byte loads when the frames are 16-byte aligned, four in flight per thread.
// the bound table of the pair format: l1[c] = max over filters of sum |w| over the 16 input channels of chunk c (all taps);
// one block per chunk.  For the folded up-convs the unfolded 3 x 3 weights bound the folded ones (triangle inequality).
__global__ __launch_bounds__(256) void split_l1_kernel(const float* __restrict__ w, float* __restrict__ l1, int OC, int Cin, int taps) {
    __shared__ float red[256];
    const int c = blockIdx.x;
    float m = 0.f;
    for (int oc = threadIdx.x; oc < OC; oc += 256) {
        const float* wp = w + ((long long)oc * Cin + c * SP_CK) * taps;
        float sum = 0.f;
        for (int e = 0; e < SP_CK * taps; ++e) sum += fabsf(wp[e]);
        m = fmaxf(m, sum);
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) l1[c] = red[0];
}

template <bool VEC>
__global__ __launch_bounds__(256) void absmax_frames_kernel(const float* __restrict__ x, long long batch_stride, long long per_frame,
                                                            unsigned* __restrict__ slots) {
    const float* xn = x + (long long)blockIdx.y * batch_stride;
    float m = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    if constexpr (VEC) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(xn);
        const long long n4 = per_frame >> 2;
        long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const f32x4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
            m = sp_amax4(sp_amax4(sp_amax4(sp_amax4(m, a), b), c), d);
        }
        for (; i < n4; i += stride) m = sp_amax4(m, x4[i]);
        for (long long t = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; t < per_frame; t += stride) m = fmaxf(m, fabsf(xn[t]));
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_frame; i += stride) m = fmaxf(m, fabsf(xn[i]));
    }
    absmax_commit(slots + blockIdx.y, m);
}

int absmax_frames_launch(const float* x, long long batch_stride, int n, long long per_frame, unsigned* slots, hipStream_t stream) {
    if (!x || !slots || n < 1 || per_frame < 1) return KBN_ERR_INVALID_ARGUMENT;
    const bool vec = !((reinterpret_cast<uintptr_t>(x) & 15) || (batch_stride & 3));
    const int blocks = (int)std::min<long long>(std::max(1, 2048 / n), (per_frame + 4095) / 4096);   // >= 16 floats per thread, ~2048 workgroups
    if (vec) hipLaunchKernelGGL(absmax_frames_kernel<true>, dim3(blocks, n), dim3(256), 0, stream, x, batch_stride, per_frame, slots);
    else hipLaunchKernelGGL(absmax_frames_kernel<false>, dim3(blocks, n), dim3(256), 0, stream, x, batch_stride, per_frame, slots);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

template <int N, int NBX>
__device__ __forceinline__ void sp_wait_b(f32x4 (&b)[NBX][2]) {   // vmcnt(N), tied to the registers it guards
    static_assert(NBX == 1 || NBX == 2, "one or two 32-filter blocks per wave");
    if constexpr (NBX == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]) : "n"(N));
    else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0][0]), "+v"(b[0][1]) : "n"(N));
}

// RG row groups x (8 / RG) filter groups of waves; a wave owns MB = TH / RG rows (m-blocks) x two 32-filter n-blocks.
// APART: the two small terms (h1 w2, h2 w1 2^-11; 2^-11 of the sum) accumulate in their own registers, so that the main
// accumulator is rounded once per k-step instead of three times (error vs fp64 / 1.7: the level of the Winograd kernel).
// BLDS: the weights of a chunk (nine taps) are copied into LDS by LDS-DMA, double buffered like the inputs: every global
// access of chunk c+1 is issued at the start of chunk c and awaited once, in front of the barrier that ends chunk c --
// no vmcnt wait sits between MFMAs (waves retire their loads in order: with the weights fetched per tap into registers,
// the tap-2 wait also had to wait for the next chunk's inputs, +28 % on the decoder's concat convs).
// PIN0: source 0 is a pair tensor (its chunks are staged by LDS-DMA; a second, fp32 source is split here as before and the
// accumulators change window between the two); POUT: the output is written as a pair tensor (MFMA operands swapped: a
// lane's accumulator registers run over the FILTERS of one pixel).  Both: MODE 0 with the weights through LDS.
// TP (MODE 0): TRANSPOSED tiles for the last, narrow column of a map whose width leaves 1-16 columns behind the whole 32-column
// tiles (KITTI: 76 = 2 x 32 + 12, 304 = 9 x 32 + 16): 32 rows x 16 columns per workgroup, a 32-pixel MFMA block = two rows of 16
// pixels, the staged region 34 rows x 18 columns (the same 612 granules).  Half the MFMAs of the column tiles it replaces
// (22 x 76: 55 instead of 66 blocks per frame).  The transposed tiles are the LAST workgroups of the same launch
// (conv3x3_split_mixed_kernel): as a launch of their own -- 128 workgroups for deconv4's conv -- they would cost the round of
// workgroups they save.
// MIXED: the grid is p.nblocks whole tiles followed by p.tp_nblocks transposed ones: the body (conv_split_body.inc) is compiled
// twice into the kernel, once per tile form, and a workgroup takes the one its block index selects.
// ONE: the THROUGHPUT-ONLY one-term mode (KBN_FP16_ONE_TERM=1; BASELINE configs[2]'s 16-bit leg and the ablation "same skeleton, a
// third of the MFMAs"): only h1 w1 is issued -- plain fp16 operands, fp32 accumulation -- and the h2 halves of the staged tiles
// are neither fetched (pair sources, weights through LDS) nor read.  Never on the parity-gated path.
template <int MODE, int RG, bool APART, bool BLDS, int NBW = 2, bool PIN0 = false, bool POUT = false, bool MIXED = false, bool ONE = false, bool KSPLIT = false>   // NBW: 32-filter blocks per wave
__global__ __launch_bounds__(SP_THREADS, 1) void conv3x3_split_kernel(const SplitConvParams p) {
    static_assert(!MIXED || (MODE == 0 && BLDS), "transposed tiles: the concat kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (!MIXED || (int)blockIdx.x < p.nblocks) {
        constexpr bool TP = false;
        const int block = blockIdx.x, nblocks = p.nblocks, tilesX = p.tilesX, tilesY = p.tilesY;
#include "conv_split_body.inc"
    } else if constexpr (MIXED) {
        constexpr bool TP = true;
        const int block = (int)blockIdx.x - p.nblocks, nblocks = p.tp_nblocks, tilesX = 1, tilesY = p.tp_tilesY;
#include "conv_split_body.inc"
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Nearest-2x up-conv in its FOLDED form on split operands (MODE 3 of the entry point).  An output pixel (2Y+py, 2X+px)
// of conv3x3(upsample2x(x)) only sees the 2 x 2 low-resolution pixels (Y+py-1+dy, X+px-1+dx), dy, dx in {0, 1}, with
// the 3 x 3 weights summed over the taps that land on the same source pixel (rows: py 0 -> {0}, {1,2}; py 1 -> {0,1},
// {2}; columns alike): four 2 x 2 convs, one per output parity, 16 channel products per low-resolution pixel instead of
// 36 -- 2.25x fewer MFMAs than MODE 1.  M = 32 low-resolution pixels of a row (one parity), N = 32 filters.
// Workgroup = 8 waves = 8 row groups; tile 16 x 32 low-resolution pixels (32 x 64 outputs) x 32 filters; a wave owns two
// low-resolution rows x four parities (eight accumulator blocks).  The 16 (parity, tap) weight sets of a chunk are
// visited grouped by the source offset they read, (ox = px+dx, s = py+dy): the two rows of a wave then need the A
// fragments of staged rows s and s+1 at column offset ox -- 12 fragment reads per chunk serve all 96 MFMAs; weights are
// packed in that visiting order and fetched three sets ahead.  K per output = 4 Cin: a third of the roundings of the
// unfolded form, so ONE accumulator per block keeps the accuracy of the APART kernels above.
constexpr int UF_NT = 32, UF_ITEMS = 16;
struct UfItem { int ox, s, py, dy, px, dx; };
__host__ __device__ constexpr UfItem uf_item(int it) {
    // ox 0: (px,dx) = (0,0); ox 1: (0,1), (1,0); ox 2: (1,1).  Same for s over (py,dy).  Order: ox, s, (py,dy), (px,dx).
    int ox = it < 4 ? 0 : (it < 12 ? 1 : 2);
    int r = it - (ox == 0 ? 0 : (ox == 1 ? 4 : 12));
    const int ncol = ox == 1 ? 2 : 1;                 // (px,dx) combos of this ox
    const int rowidx = r / ncol, colidx = r % ncol;   // rowidx 0..3 over (s, (py,dy)): s0:1, s1:2, s2:1
    const int s = rowidx == 0 ? 0 : (rowidx < 3 ? 1 : 2);
    const int py = s == 0 ? 0 : (s == 2 ? 1 : rowidx - 1);
    const int px = ox == 0 ? 0 : (ox == 2 ? 1 : colidx);
    return UfItem{ox, s, py, s - py, px, ox - px};
}
// folded weight of (py, dy) x (px, dx) from the nine taps of one (filter, channel).  tr = 0: conv3x3(upsample2x(x)) -- the taps that
// land on the same low-resolution pixel are summed.  tr = 1: ConvTranspose2d(kernel 3, stride 2, padding 1, output_padding 1)
// (reference src/net_utils.py:383-390) in the same four-parity form: out[2i - 1 + ky] += in[i] w[ky] gives an even output row (py 0) the one
// tap ky = 1 of row Y (dy 1), an odd one (py 1) ky = 2 of row Y (dy 0) and ky = 0 of row Y + 1 (dy 1); columns alike.  Nine of the sixteen
// folded weights are taps, seven are zero (`w9`: the weight with out_channels leading, i.e. the module's in x out x 3 x 3 weight with its first
// two axes swapped -- the host does that).
__device__ __forceinline__ void uf_taps(int p, int d, int tr, int& k0, int& k1) {
    if (tr) { k0 = p == 0 ? 1 : (d == 0 ? 2 : 0); k1 = (p == 0 && d == 0) ? 0 : k0; return; }   // (0,0): empty range
    k0 = (p == 0) ? (d == 0 ? 0 : 1) : (d == 0 ? 0 : 2);
    k1 = (p == 0) ? (d == 0 ? 0 : 2) : (d == 0 ? 1 : 2);
}
__device__ __forceinline__ float uf_fold(const float* w9, int py, int dy, int px, int dx, int tr = 0) {
    int r0, r1, c0, c1;
    uf_taps(py, dy, tr, r0, r1);
    uf_taps(px, dx, tr, c0, c1);
    float acc = 0.f;
    for (int r = r0; r <= r1; ++r) {
        float row = 0.f;
        for (int c = c0; c <= c1; ++c) row += w9[r * 3 + c];
        acc += row;
    }
    return acc;
}

__global__ void uf_scale_kernel(const float* __restrict__ w, float* __restrict__ inv_scale, int OC, int Cin, int tr) {
    const int oc = blockIdx.x;
    __shared__ float red[256];
    float m = 0.f;
    if (oc < OC)
        for (int i = threadIdx.x; i < Cin * UF_ITEMS; i += 256) {
            const int c = i / UF_ITEMS;
            const UfItem t = uf_item(i % UF_ITEMS);
            m = fmaxf(m, fabsf(uf_fold(w + ((long long)oc * Cin + c) * 9, t.py, t.dy, t.px, t.dx, tr)));
        }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int ex = SP_WEXP;
        if (red[0] > 0.f && red[0] < 3.0e38f) (void)frexpf(red[0], &ex);
        int e = SP_WEXP - ex;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        inv_scale[oc] = ldexpf(1.f, -e);
    }
}

// OIHW fp32 -> [n-tile][chunk][item][part][k-group][32 filters][8 channels] fp16 of the folded weights
__global__ void uf_pack_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale, _Float16* __restrict__ packed,
                               int OC, int Cin, int nchunks, long long total, int tr) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    constexpr int per_item = 2 * 2 * UF_NT * 8, per_chunk = UF_ITEMS * per_item;
    int r = (int)(e % per_chunk);
    const long long q = e / per_chunk;
    const int chunk = (int)(q % nchunks), nt = (int)(q / nchunks);
    const int item = r / per_item; r -= item * per_item;
    const int part = r / (2 * UF_NT * 8); r -= part * 2 * UF_NT * 8;
    const int g = r / (UF_NT * 8); r -= g * UF_NT * 8;
    const int n = r >> 3, k = r & 7;
    const int c = chunk * SP_CK + g * 8 + k, oc = nt * UF_NT + n;
    _Float16 h = (_Float16)0.f;
    if (c < Cin && oc < OC) {
        const UfItem t = uf_item(item);
        const float ws = uf_fold(w + ((long long)oc * Cin + c) * 9, t.py, t.dy, t.px, t.dx, tr) * (1.f / inv_scale[oc]);
        const _Float16 w1 = (_Float16)ws;
        h = part == 0 ? w1 : (_Float16)(ws - (float)w1);
    }
    packed[e] = h;
}

template <int N>
__device__ __forceinline__ void uf_wait_b(f32x4 (&b)[2]) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0]), "+v"(b[1]) : "n"(N));
}

// BLDS: the sixteen weight sets of a chunk (32 KiB) are copied into LDS by LDS-DMA, double buffered, like the nine taps
// of the concat convs: eight waves fetching every set straight from L1 move 256 KiB per chunk through the CU's vector
// memory pipe (42 B/clk of its 64 beside the input loads); through LDS it is 32 KiB, every global access of chunk c+1
// is issued at the start of chunk c and awaited once, late in it.
template <bool BLDS>
__global__ __launch_bounds__(SP_THREADS, 1) void upconv2x_split_kernel(const SplitConvParams p) {
    constexpr int ROWS = 18, COLS = 34, NPIX = ROWS * COLS, A_PART = 2 * NPIX * 16, A_BYTES = 2 * A_PART, PR = 3, NA_ALL = PR * 8;
    constexpr int B_ITEM = 2 * 2 * UF_NT * 16, NBL = 2, D = 3;       // bytes per weight set; loads per set; sets fetched ahead
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0");   // fp16 results flush subnormals (see conv3x3_split_kernel)
    const int tid = threadIdx.x, lane = tid & 63;
    const int rg = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave = row group: low-resolution rows 2 rg, 2 rg + 1
    const int lm = lane & 31, g = lane >> 5;
    int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int nt = bid % p.nTilesN;
    bid /= p.nTilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int oy0 = ty * 16, ox0 = tx * 32;                            // low-resolution tile origin
    const int H = p.H, W = p.W, sH = p.sH, sW = p.sW;
    const long long plane = (long long)sH * sW;
    const int nchunks = p.Cin / SP_CK;
    float prescale, unscale;
    sp_act_scale(p, n, prescale, unscale);

    const int kg_st = rg >> 2, t256 = tid & 255;
    int goff[PR];
#pragma unroll
    for (int u = 0; u < PR; ++u) {
        const int pix = u * 256 + t256;
        const int r = pix / COLS, c = pix - r * COLS;
        const int Y = oy0 - 1 + r, X = ox0 - 1 + c;
        goff[u] = (pix < NPIX && Y >= 0 && Y < sH && X >= 0 && X < sW) ? (Y * sW + X) * 4 : -1;
    }
    const _Float16* wp_nt = p.wp + (long long)nt * nchunks * (UF_ITEMS * B_ITEM / 2);

    float va[PR][8];
    auto load_chunk = [&](int chunk) {
        const float* base = p.src[0] + (long long)n * p.src_bstride[0] + (long long)(chunk * SP_CK + kg_st * 8) * plane;
#pragma unroll
        for (int u = 0; u < PR; ++u) {
            const unsigned voff = goff[u] < 0 ? 0u : (unsigned)goff[u];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* sb = base + (long long)k * plane;
                asm volatile("global_load_dword %0, %1, %2" : "=v"(va[u][k]) : "v"(voff), "s"(sb) : "memory");
            }
        }
    };
    auto store_round = [&](int buf, int u) {
        unsigned char* A = smem + buf * A_BYTES + kg_st * NPIX * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(va[u][k]));
        const int pix = u * 256 + t256;
        if (pix < NPIX) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = goff[u] >= 0 ? va[u][k] : 0.f;
            sph8 h1, h2;
            sp_split8(v, prescale, h1, h2);
            *reinterpret_cast<sph8*>(A + pix * 16) = h1;
            *reinterpret_cast<sph8*>(A + A_PART + pix * 16) = h2;
        }
    };
    const unsigned boff = (unsigned)((g * UF_NT + lm) * 16);
    auto load_b = [&](f32x4 (&b)[2], int chunk, int item) {
        const unsigned char* base = reinterpret_cast<const unsigned char*>(wp_nt + ((long long)chunk * UF_ITEMS + item) * (B_ITEM / 2));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned char* sb = base + t * 2 * UF_NT * 16;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[t]) : "v"(boff), "s"(sb) : "memory");
        }
    };

    spf16 acc[2][2][2];   // [low-resolution row of the wave][py][px]
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a >> 2][(a >> 1) & 1][a & 1][i] = 0.f;

    const unsigned char* const aptr = smem + (g * NPIX + 2 * rg * COLS + lm) * 16;
    sph8 af[4][2];        // A fragments of staged rows 2 rg + 0..3 at the current column offset (two split terms each)
    auto load_arow = [&](int abuf, int ry, int ox) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            af[ry][t] = *reinterpret_cast<const sph8*>(aptr + abuf + t * A_PART + (ry * COLS + ox) * 16);
    };

    if constexpr (BLDS) {
        constexpr int B_CHUNK = UF_ITEMS * B_ITEM;
        constexpr int WAIT_IT = 9;   // the set whose MFMAs follow the wait for chunk c+1's accesses (issued ahead of set 0)
        static_assert(WAIT_IT + PR < UF_ITEMS, "the staging rounds follow the wait inside the chunk");
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(reinterpret_cast<const float*>(smem)));
        auto stage_b = [&](int bbuf, int chunk) {
            const float* src = reinterpret_cast<const float*>(wp_nt + (long long)chunk * (B_CHUNK / 2));
            const unsigned dst = lds0 + (unsigned)(2 * A_BYTES + bbuf * B_CHUNK);
            constexpr int n4 = B_CHUNK / 16;
            static_assert(n4 % SP_THREADS == 0, "whole rounds of the workgroup");
#pragma unroll
            for (int e0 = 0; e0 < n4; e0 += SP_THREADS) {
                const int eb = e0 + rg * 64;
                lds_dma16_s(src + eb * 4, (unsigned)(lane * 16), dst + eb * 16);
            }
        };
        const unsigned char* const bptr = smem + 2 * A_BYTES + boff;
        auto body = [&](int c, auto more_tag, auto chk_tag) {
            constexpr bool MORE = decltype(more_tag)::value, CHK = decltype(chk_tag)::value;
            const int abuf = (c & 1) * A_BYTES;
            const unsigned char* B = bptr + (c & 1) * B_CHUNK;
            if (MORE) {
                stage_b((c & 1) ^ 1, c + 1);
                load_chunk(c + 1);
            }
            load_arow(abuf, 0, 0);
            load_arow(abuf, 1, 0);
            sph8 bwq[2][2];   // (w1, w2) of the current / next set
            bwq[0][0] = *reinterpret_cast<const sph8*>(B);
            bwq[0][1] = *reinterpret_cast<const sph8*>(B + 2 * UF_NT * 16);
#pragma unroll
            for (int it = 0; it < UF_ITEMS; ++it) {
                const UfItem t = uf_item(it);
                const bool first_of_group = it == 0 || uf_item(it - 1).s != t.s || uf_item(it - 1).ox != t.ox;
                if (first_of_group) {   // fetch what the NEXT group reads and this one does not hold
                    if (t.s == 0) load_arow(abuf, 2, t.ox);
                    else if (t.s == 1) load_arow(abuf, 3, t.ox);
                    else if (t.ox < 2) { load_arow(abuf, 0, t.ox + 1); load_arow(abuf, 1, t.ox + 1); }
                }
                if (it + 1 < UF_ITEMS) {
                    bwq[(it + 1) & 1][0] = *reinterpret_cast<const sph8*>(B + (it + 1) * B_ITEM);
                    bwq[(it + 1) & 1][1] = *reinterpret_cast<const sph8*>(B + (it + 1) * B_ITEM + 2 * UF_NT * 16);
                }
                sph8 bw[3];
                bw[0] = bwq[it & 1][0];
                bw[1] = bwq[it & 1][1];
                bw[2] = bw[0] * (_Float16)0.00048828125f;
                if (MORE && it == WAIT_IT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next chunk's weights (DMA) and inputs
                __builtin_amdgcn_sched_barrier(0);
                constexpr int TA[3] = {0, 0, 1};
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        if (CHK && oy0 + 2 * rg + mb >= sH) continue;     // low-resolution row below the map: no MFMAs (wave-uniform)
                        acc[mb][t.py][t.px] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mb + t.s][TA[k]], bw[k], acc[mb][t.py][t.px], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (MORE && it >= WAIT_IT && it - WAIT_IT < PR) store_round((c & 1) ^ 1, it - WAIT_IT);
            }
            __syncthreads();
        };
        load_chunk(0);
        stage_b(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < PR; ++u) store_round(0, u);
        __syncthreads();
        auto k_loop = [&](auto chk_tag) {
            for (int c = 0; c + 1 < nchunks; ++c) body(c, std::true_type{}, chk_tag);
            body(nchunks - 1, std::false_type{}, chk_tag);
        };
        if (!KBN_SPLIT_STRAIGHT || oy0 + 16 > sH) k_loop(std::true_type{});   // tile with rows below the map (workgroup-uniform)
        else k_loop(std::false_type{});
    } else {
    f32x4 bq[4][2];       // weight sets in flight: set `it` lives in bq[it % 4]
    auto chunk_body = [&](int c, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        constexpr int NA = MORE ? NA_ALL : 0;
        const int abuf = (c & 1) * A_BYTES;
        load_arow(abuf, 0, 0);
        load_arow(abuf, 1, 0);
#pragma unroll
        for (int it = 0; it < UF_ITEMS; ++it) {
            const UfItem t = uf_item(it);
            const bool first_of_group = it == 0 || uf_item(it - 1).s != t.s || uf_item(it - 1).ox != t.ox;
            if (first_of_group) {   // fetch what the NEXT group reads and this one does not hold
                if (t.s == 0) load_arow(abuf, 2, t.ox);
                else if (t.s == 1) load_arow(abuf, 3, t.ox);
                else if (t.ox < 2) { load_arow(abuf, 0, t.ox + 1); load_arow(abuf, 1, t.ox + 1); }
            }
            f32x4 (&bc)[2] = bq[it % 4];
            if (it + D < UF_ITEMS) load_b(bq[(it + D) % 4], c, it + D);
            else if (MORE) load_b(bq[(it + D) % 4], c + 1, it + D - UF_ITEMS);
            if (it == 0 && MORE) load_chunk(c + 1);
            // outstanding, oldest first: b(it) b(it+1) b(it+2) [b(it+3) | inputs in issue order]
            if (it <= D) uf_wait_b<D * NBL + NA>(bc);
            else if (MORE || it + D < UF_ITEMS) uf_wait_b<D * NBL>(bc);
            else if (it == UF_ITEMS - 3) uf_wait_b<2 * NBL>(bc);
            else if (it == UF_ITEMS - 2) uf_wait_b<NBL>(bc);
            else uf_wait_b<0>(bc);
            sph8 bw[3];
            bw[0] = __builtin_bit_cast(sph8, bc[0]);
            bw[1] = __builtin_bit_cast(sph8, bc[1]);
            bw[2] = bw[0] * (_Float16)0.00048828125f;
            __builtin_amdgcn_sched_barrier(0);
            constexpr int TA[3] = {0, 0, 1};
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    if (oy0 + 2 * rg + mb >= sH) continue;            // low-resolution row below the map: no MFMAs (wave-uniform)
                    acc[mb][t.py][t.px] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mb + t.s][TA[k]], bw[k], acc[mb][t.py][t.px], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (MORE && it > D + 1 && it - D - 2 < PR) store_round((c & 1) ^ 1, it - D - 2);   // the wait of set D+1 covered the inputs
        }
        __syncthreads();
    };

    load_chunk(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < PR; ++u) store_round(0, u);
#pragma unroll
    for (int it = 0; it < D; ++it) load_b(bq[it], 0, it);
    __syncthreads();
    for (int c = 0; c + 1 < nchunks; ++c) chunk_body(c, std::true_type{});
    chunk_body(nchunks - 1, std::false_type{});
    }

    // ---- epilogue: acc[mb][py][px][i]: low-resolution x = 8 (i / 4) + 4 g + (i % 4), filter lm; outputs (2 Y + py, 2 x + px)
    const long long oplane = (long long)H * W;
    const int oc = nt * UF_NT + lm;
    const float inv = p.inv_scale[oc] * unscale;
    float* outc = p.out + (long long)n * p.out_bstride + (long long)oc * oplane;
    const float slope = p.act ? p.slope : 1.f;
    float amax = 0.f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int Y = oy0 + 2 * rg + mb;
        if (Y >= sH || oc >= p.OC) continue;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            float* orow = outc + (long long)(2 * Y + py) * W;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int X = 2 * (ox0 + 8 * q4 + 4 * g);              // first output column of this lane's 8
                f32x4 v0, v1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = acc[mb][py][j & 1][q4 * 4 + (j >> 1)] * inv;
                    const float b = acc[mb][py][j & 1][q4 * 4 + 2 + (j >> 1)] * inv;
                    v0[j] = a > 0.f ? a : a * slope;
                    v1[j] = b > 0.f ? b : b * slope;
                }
                if (X < W) { *reinterpret_cast<f32x4*>(orow + X) = v0; amax = sp_amax4(amax, v0); }
                if (X + 4 < W) { *reinterpret_cast<f32x4*>(orow + X + 4) = v1; amax = sp_amax4(amax, v1); }
            }
        }
    }
    if (p.out_amax) absmax_commit(p.out_amax + n, amax);
}


// ------------------------------------------------------------------------------------------------------------------
// The folded up-conv for NARROW layers (at most 16 filters, Cin % 32 == 0: deconv0's 64 -> 12 up-conv at full
// resolution, reference src/net_utils.py:484-499 with n_filters_decoder[-1] = 12): 16-filter tiles on
// v_mfma_f32_16x16x32_f16 instead of 32-filter tiles on 32x32x16 -- 12 of 16 columns live instead of 12 of 32.  Same
// arithmetic as upconv2x_split_kernel (sixteen folded 2 x 2 weight sets, three fp16 products per fp32 product, one
// accumulator per block).  M = 16 low-resolution pixels of a row, N = 16 filters, K = 32 channels per MFMA; chunk = 32
// channels.  Workgroup = 8 waves = 4 row groups x 2 column halves; tile 16 x 32 low-resolution pixels; a wave owns four
// low-resolution rows x 16 pixels x four parities (sixteen 16 x 16 accumulator blocks).  A in LDS as
// [part][k-group (4)][pixel][8 fp16]; a group of weight sets (ox, s) reads the staged rows s .. s+3 at column offset
// ox: rows stream through eight register slots (rows 2 and 3 have two: the last group of one column offset still
// reads them while the first of the next is being fetched).  Weights: [chunk][set][part][k-group][16 filters][8
// channels] fp16, one 1 KiB wave-wide load per (set, part), fetched three sets ahead.
constexpr int U16_NT = 16, U16_CK = 32;

__host__ __device__ constexpr bool uf_narrow(int out_channels, int in_channels) {
    return out_channels <= U16_NT && (in_channels % U16_CK) == 0;
}

// OIHW fp32 -> [n-tile][chunk][set][part][k-group (4)][16 filters][8 channels] fp16 of the folded weights
__global__ void uf16_pack_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale, _Float16* __restrict__ packed,
                                 int OC, int Cin, int nchunks, long long total, int tr) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    constexpr int per_item = 2 * 4 * U16_NT * 8, per_chunk = UF_ITEMS * per_item;
    int r = (int)(e % per_chunk);
    const long long q = e / per_chunk;
    const int chunk = (int)(q % nchunks), nt = (int)(q / nchunks);
    const int item = r / per_item; r -= item * per_item;
    const int part = r / (4 * U16_NT * 8); r -= part * 4 * U16_NT * 8;
    const int g = r / (U16_NT * 8); r -= g * U16_NT * 8;
    const int n = r >> 3, k = r & 7;
    const int c = chunk * U16_CK + g * 8 + k, oc = nt * U16_NT + n;
    _Float16 h = (_Float16)0.f;
    if (c < Cin && oc < OC) {
        const UfItem t = uf_item(item);
        const float ws = uf_fold(w + ((long long)oc * Cin + c) * 9, t.py, t.dy, t.px, t.dx, tr) * (1.f / inv_scale[oc]);
        const _Float16 w1 = (_Float16)ws;
        h = part == 0 ? w1 : (_Float16)(ws - (float)w1);
    }
    packed[e] = h;
}

// NW waves per workgroup.  8: tile 16 x 32 low-resolution pixels, the staged chunk double buffered (157 KB of LDS, one
// workgroup per CU).  4 (the launch default): tile 8 x 32, ONE staging buffer (43.5 KB) refilled from registers between
// two barriers, two workgroups per CU -- with only Cin / 32 = 2 chunks per tile the first fetch and the stores are most
// of a workgroup's life, and a second resident workgroup multiplies meanwhile: 700 -> 616 us for deconv0's up-conv.
// (Measured and not kept, DESIGN.md round 3: persistent workgroups, with the weights from L2 as here or resident in LDS.)
// PIN: the input is a pair tensor, staged by LDS-DMA (see upconv2x_split64_kernel); POUT: the output is written as one with
// 16 channels (two k-groups; channels past OC are zero) -- the decoder tail (csrc/tail.hip) stages it by DMA
template <int NW, bool PIN, bool POUT = false, bool ONE = false>   // ONE: h1 w1 alone (KBN_FP16_ONE_TERM, throughput only; see conv3x3_split_kernel)
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void upconv2x_split16_kernel(const SplitConvParams p) {
    static_assert(NW == 8 || NW == 4, "8 waves (16-row tiles) or 4 waves (8-row tiles)");
    constexpr bool DB = NW == 8;
    constexpr int ROWS = 2 * NW, TPG = 16 * NW;                        // low-resolution rows per tile; threads per k-group in staging
    constexpr int COLS = 34, NPIX = (ROWS + 2) * COLS, KG = 4;
    // plane pitch of a k-group, padded to a multiple of 16 granules: ds_read_b128 serves lanes {0-3, 12-15, 20-27} together, i.e. the kq = 0 and
    // kq = 1 halves of an A fragment, conflict-free only when they sit 0 (mod 256 B) apart (340 granules: SQ_LDS_BANK_CONFLICT / IDX_ACTIVE 0.50)
    constexpr int NPP = (NPIX + 15) / 16 * 16;
    constexpr int A_PART = KG * NPP * 16, A_BYTES = 2 * A_PART;       // [part][k-group][pixel (pitch NPP)][8 fp16]
    constexpr int PR = (NPIX + TPG - 1) / TPG;                         // staging rounds of a quarter of the threads (one k-group each)
    // pair input: a staged chunk is 8 planes (term, k-group) x NPIX granules; wave-wide DMA id = plane * NR + round
    constexpr int NR = (NPIX + 63) / 64, NDMA = (ONE ? 1 : 2) * KG * NR, DPW = NDMA / NW;   // ONE: the h1 planes only
    static_assert(NDMA % NW == 0, "the same number of DMAs in every wave (the vmcnt arithmetic counts them)");
    constexpr int NA_ALL = PIN ? DPW : PR * 8;                         // vector-memory operations of a wave per staged chunk
    constexpr int B_ITEM = 2 * KG * U16_NT * 16, NBL = 2, D = 3;      // bytes per weight set; loads per set; sets fetched ahead
    static_assert(D * NBL + NA_ALL < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0");   // fp16 results flush subnormals (see conv3x3_split_kernel)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave >> 1, mblk = wave & 1;                         // low-resolution rows 4 rg .. 4 rg + 3, pixels 16 mblk .. + 15
    const int lp = lane & 15, kq = lane >> 4;
    int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int nt = bid % p.nTilesN;
    bid /= p.nTilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int oy0 = ty * ROWS, ox0 = tx * 32;                            // low-resolution tile origin
    const int H = p.H, W = p.W, sH = p.sH, sW = p.sW;
    const long long plane = (long long)sH * sW;
    const int nchunks = p.Cin / U16_CK;
    float prescale, unscale;
    if constexpr (PIN) { prescale = 0.f; unscale = 1.f / p.pair_src_scale[n]; }
    else sp_act_scale(p, n, prescale, unscale);

    const int kg_st = wave / (NW / 4), t128 = tid & (TPG - 1);         // staging: NW / 4 waves per k-group
    int goff[PR];
#pragma unroll
    for (int u = 0; u < PR; ++u) {
        const int pix = u * TPG + t128;
        const int r = pix / COLS, c = pix - r * COLS;
        const int Y = oy0 - 1 + r, X = ox0 - 1 + c;
        goff[u] = (pix < NPIX && Y >= 0 && Y < sH && X >= 0 && X < sW) ? (Y * sW + X) * 4 : -1;
    }
    const unsigned char* wp_nt = reinterpret_cast<const unsigned char*>(p.wp) + (long long)nt * nchunks * (UF_ITEMS * B_ITEM);

    float va[PR][8];
    auto load_chunk = [&](int chunk) {
        const float* base = p.src[0] + (long long)n * p.src_bstride[0] + (long long)(chunk * U16_CK + kg_st * 8) * plane;
#pragma unroll
        for (int u = 0; u < PR; ++u) {
            const unsigned voff = goff[u] < 0 ? 0u : (unsigned)goff[u];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* sb = base + (long long)k * plane;         // wave-uniform
                asm volatile("global_load_dword %0, %1, %2" : "=v"(va[u][k]) : "v"(voff), "s"(sb) : "memory");
            }
        }
    };
    auto store_round = [&](int buf, int u) {
        unsigned char* A = smem + buf * A_BYTES + kg_st * NPP * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(va[u][k]));
        const int pix = u * TPG + t128;
        if (pix < NPIX) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = goff[u] >= 0 ? va[u][k] : 0.f;
            sph8 h1, h2;
            sp_split8(v, prescale, h1, h2);
            *reinterpret_cast<sph8*>(A + pix * 16) = h1;
            *reinterpret_cast<sph8*>(A + A_PART + pix * 16) = h2;
        }
    };
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(reinterpret_cast<const float*>(smem)));
    unsigned dvoff[PIN ? DPW : 1];
    if constexpr (PIN) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int pix = ((wave + NW * i) % NR) * 64 + lane;
            const int r = pix / COLS, c = pix - r * COLS;
            const int Y = oy0 - 1 + r, X = ox0 - 1 + c;
            dvoff[i] = (pix < NPIX && Y >= 0 && Y < sH && X >= 0 && X < sW) ? (unsigned)(Y * sW + X) * 16u : (unsigned)(sH * sW) * 16u;
        }
    }
    const long long pplane = pair_plane_halves(sH, sW);
    auto dma_chunk = [&](int buf, int chunk) {
        const _Float16* pn = p.pair_src + (long long)n * p.pair_src_bstride + (long long)(KG * chunk) * 2 * pplane;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int id = wave + NW * i, plane = id / NR, j = id - plane * NR;
            const int t = plane / KG, kgl = plane - t * KG;
            const unsigned long long mask = (j == NR - 1 && (NPIX & 63)) ? ((1ull << (NPIX & 63)) - 1) : ~0ull;
            lds_dma16_sm(reinterpret_cast<const float*>(pn + (long long)(kgl * 2 + t) * pplane), dvoff[i],
                         lds0 + (unsigned)(buf * A_BYTES + t * A_PART + (kgl * NPP + j * 64) * 16), mask);
        }
    };
    const unsigned boff = (unsigned)(lane * 16);                       // [k-group kq][filter lp][8 channels]
    auto load_b = [&](f32x4 (&b)[2], int chunk, int item) {
        const unsigned char* base = wp_nt + ((long long)chunk * UF_ITEMS + item) * B_ITEM;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned char* sb = base + t * (B_ITEM / 2);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[t]) : "v"(boff), "s"(sb) : "memory");
        }
    };

    spf4 acc[4][2][2];    // [low-resolution row of the wave][py][px]
#pragma unroll
    for (int a = 0; a < 16; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[a >> 2][(a >> 1) & 1][a & 1][i] = 0.f;

    // staged row r (0..5 of the wave's six) at column offset ox lives in register slot r, rows 2 and 3 at odd ox in 6 and 7
    const unsigned char* const aptr = smem + (kq * NPP + 4 * rg * COLS + 16 * mblk + lp) * 16;
    sph8 af[8][2];
    auto slot = [](int r, int ox) constexpr { return (r == 2 || r == 3) && (ox & 1) ? r + 4 : r; };
    auto load_arow = [&](int abuf, int r, int ox) {
#pragma unroll
        for (int t = 0; t < (ONE ? 1 : 2); ++t)
            af[slot(r, ox)][t] = *reinterpret_cast<const sph8*>(aptr + abuf + t * A_PART + (r * COLS + ox) * 16);
    };

    f32x4 bq[4][2];       // weight sets in flight: set `it` lives in bq[it % 4]
    auto chunk_body = [&](int c, auto more_tag, auto chk_tag) {
        constexpr bool MORE = decltype(more_tag)::value, CHK = decltype(chk_tag)::value;
        constexpr int NA = (MORE && (DB || !PIN)) ? NA_ALL : 0;   // one buffer + pair input: the DMA follows the chunk's barrier
        const int abuf = DB ? (c & 1) * A_BYTES : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) load_arow(abuf, r, 0);
#pragma unroll
        for (int it = 0; it < UF_ITEMS; ++it) {
            const UfItem t = uf_item(it);
            const bool first_of_group = it == 0 || uf_item(it - 1).s != t.s || uf_item(it - 1).ox != t.ox;
            if (first_of_group) {   // fetch what the NEXT groups read and this one does not hold
                if (t.s == 0) load_arow(abuf, 4, t.ox);
                else if (t.s == 1) load_arow(abuf, 5, t.ox);
                else if (t.ox < 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) load_arow(abuf, r, t.ox + 1);
                }
            }
            f32x4 (&bc)[2] = bq[it % 4];
            if (it + D < UF_ITEMS) load_b(bq[(it + D) % 4], c, it + D);
            else if (MORE) load_b(bq[(it + D) % 4], c + 1, it + D - UF_ITEMS);
            if (it == 0 && MORE) {
                if constexpr (!PIN) load_chunk(c + 1);
                else if (DB) dma_chunk((c & 1) ^ 1, c + 1);   // the other buffer was last read a chunk (a barrier) ago
            }
            // outstanding, oldest first: b(it) b(it+1) b(it+2) [b(it+3) | inputs in issue order]
            if (it <= D) uf_wait_b<D * NBL + NA>(bc);
            else if (MORE || it + D < UF_ITEMS) uf_wait_b<D * NBL>(bc);
            else if (it == UF_ITEMS - 3) uf_wait_b<2 * NBL>(bc);
            else if (it == UF_ITEMS - 2) uf_wait_b<NBL>(bc);
            else uf_wait_b<0>(bc);
            sph8 bw[3];
            bw[0] = __builtin_bit_cast(sph8, bc[0]);
            bw[1] = __builtin_bit_cast(sph8, bc[1]);
            bw[2] = bw[0] * (_Float16)0.00048828125f;
            __builtin_amdgcn_sched_barrier(0);
            constexpr int TA[3] = {0, 0, 1};
#pragma unroll
            for (int k = 0; k < (ONE ? 1 : 3); ++k)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    if (CHK && oy0 + 4 * rg + mb >= sH) continue;      // low-resolution row below the map: no MFMAs (wave-uniform)
                    acc[mb][t.py][t.px] = POUT ? __builtin_amdgcn_mfma_f32_16x16x32_f16(bw[k], af[slot(mb + t.s, t.ox)][TA[k]],
                                                                                         acc[mb][t.py][t.px], 0, 0, 0)
                                               : __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot(mb + t.s, t.ox)][TA[k]], bw[k],
                                                                                         acc[mb][t.py][t.px], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (!PIN && DB && MORE && it > D + 1 && it - D - 2 < PR) store_round((c & 1) ^ 1, it - D - 2);   // the wait of set D+1 covered the inputs
        }
        __syncthreads();
        if (!DB && MORE) {       // one buffer: every wave has read its last fragment of chunk c; the inputs arrived under set D+1's wait
            if constexpr (PIN) {
                // the first weight sets of chunk c+1 (fetched above, MORE) are in flight too: vmcnt(0) covers both
                dma_chunk(0, c + 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int u = 0; u < PR; ++u) store_round(0, u);
            }
            __syncthreads();
        }
    };

    if constexpr (PIN) dma_chunk(0, 0);
    else load_chunk(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!PIN) {
#pragma unroll
        for (int u = 0; u < PR; ++u) store_round(0, u);
    }
    __syncthreads();
    auto k_loop = [&](auto chk_tag) {
        // the first weight fetches are issued INSIDE the variant that awaits them: a register copy at the branch between an
        // asm load and its vmcnt wait would copy what the register held before the data arrived
#pragma unroll
        for (int it = 0; it < D; ++it) load_b(bq[it], 0, it);
        for (int c = 0; c + 1 < nchunks; ++c) chunk_body(c, std::true_type{}, chk_tag);
        chunk_body(nchunks - 1, std::false_type{}, chk_tag);
    };
    if (!KBN_SPLIT_STRAIGHT || oy0 + ROWS > sH) k_loop(std::true_type{});   // tile with rows below the map (workgroup-uniform)
    else k_loop(std::false_type{});

    const float slope = p.act ? p.slope : 1.f;
    if constexpr (POUT) {
        // ---- pair epilogue: acc[mb][py][px][i]: low-resolution x = 16 mblk + lp, filter 4 kq + i: a lane holds half a granule
        // (channels 4 (kq & 1) ..) of k-group kq >> 1 of the outputs (2 Y + py, 2 x + px); 8-byte stores, two lanes per granule
        const float ps_out = sp_pair_out_scale(p, n);
        const long long oph = pair_plane_halves(H, W);
        _Float16* const pn = p.pair_out + (long long)n * p.pair_out_bstride;
        if (tid == 0) p.pair_out_scale[n] = ps_out;
        if (tx == 0 && ty == 0 && wave == 0 && lane < 4)    // the zero granules of the two k-groups x two terms
            *reinterpret_cast<f32x4*>(pn + (long long)lane * oph + (long long)H * W * 8) = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 inv4 = *reinterpret_cast<const f32x4*>(p.inv_scale + 4 * kq) * unscale;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * kq + i >= p.OC) inv4[i] = 0.f;          // channels past the last filter: zeros
        const int x = ox0 + 16 * mblk + lp;
        _Float16* const k0 = pn + (long long)((kq >> 1) * 2) * oph;
        float amax = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int Y = oy0 + 4 * rg + mb;
            if (Y >= sH) continue;                          // wave-uniform
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                f32x4 v0, v1;                               // this lane's four channels of outputs (2 x, 2 x + 1)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float a = acc[mb][py][0][i] * inv4[i], b = acc[mb][py][1][i] * inv4[i];
                    v0[i] = a > 0.f ? a : a * slope;
                    v1[i] = b > 0.f ? b : b * slope;
                }
                if (x < sW) amax = sp_amax4(sp_amax4(amax, v0), v1);
                sph4 a1, a2, b1, b2;
                sp_split4(v0 * ps_out, a1, a2);
                sp_split4(v1 * ps_out, b1, b2);
                const spu4 g1 = sp_pair_exchange16(a1, b1), g2 = sp_pair_exchange16(a2, b2);   // every lane takes part
                if (x < sW) {                               // even kq: the whole granule of pixel 2 x, odd kq: of pixel 2 x + 1
                    const long long o = ((long long)(2 * Y + py) * W + 2 * x + (kq & 1)) * 8;
                    *reinterpret_cast<spu4*>(k0 + o) = g1;
                    if constexpr (!ONE) *reinterpret_cast<spu4*>(k0 + oph + o) = g2;   // (the one-term consumer never fetches the h2 planes)
                }
            }
        }
        if (p.out_amax) absmax_commit(p.out_amax + n, amax);
        return;
    }
    // ---- epilogue: acc[mb][py][px][i]: low-resolution x = 16 mblk + 4 kq + i, filter lp; outputs (2 Y + py, 2 x + px)
    const long long oplane = (long long)H * W;
    const int oc = nt * U16_NT + lp;
    const float inv = p.inv_scale[oc] * unscale;                    // the table is padded to whole n-tiles
    float* outc = p.out + (long long)n * p.out_bstride + (long long)oc * oplane;
    const int X = 2 * (ox0 + 16 * mblk + 4 * kq);                      // first of this lane's 8 output columns
    float amax = 0.f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int Y = oy0 + 4 * rg + mb;
        if (Y >= sH || oc >= p.OC) continue;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            float* orow = outc + (long long)(2 * Y + py) * W;
            f32x4 v0, v1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = acc[mb][py][j & 1][j >> 1] * inv;
                const float b = acc[mb][py][j & 1][2 + (j >> 1)] * inv;
                v0[j] = a > 0.f ? a : a * slope;
                v1[j] = b > 0.f ? b : b * slope;
            }
            if (X < W) { *reinterpret_cast<f32x4*>(orow + X) = v0; amax = sp_amax4(amax, v0); }
            if (X + 4 < W) { *reinterpret_cast<f32x4*>(orow + X + 4) = v1; amax = sp_amax4(amax, v1); }
        }
    }
    if (p.out_amax) absmax_commit(p.out_amax + n, amax);
}


// ------------------------------------------------------------------------------------------------------------------
// The folded up-conv with 64-FILTER tiles (layers whose filter count fills whole 64-wide tiles: KBNet's four wide
// up-convs, 256 / 128 / 128 / 64 filters).  upconv2x_split_kernel's wave owns two low-resolution rows x one 32-filter
// block x four parities; here it owns ONE row x TWO 32-filter blocks x four parities -- the same eight accumulator
// blocks -- so a workgroup covers 8 x 32 low-resolution pixels x 64 filters: per MFMA it stages and splits 340 pixels
// instead of 612, and every input tile is staged by half as many filter tiles.  On random operands the two kernels
// take the same time (1770 vs 1768 us over the four up-convs, tools/split_bench.py); inside a KITTI forward this one is
// 3 % faster (tools/layer_profile.py: 2330 vs 2400 us for the five up-convs).  Eight-row tiles also fit the 11- and 22-row maps better.  A in LDS as before ([part][k-group]
// [pixel][8 fp16], double buffered, 43 KiB); the sixteen weight sets of a chunk are 64 KiB now, so they go through LDS
// in HALVES of eight sets (32 KiB, two buffers): the DMA of the next half flies while the current half multiplies; two
// barriers per chunk.  Weights: [n-tile][chunk][set][part][k-group][64 filters][8 channels] fp16.
constexpr int U64_NT = 64;
__host__ __device__ constexpr bool uf_wide(int out_channels) {   // whole 64-wide tiles, no more padding than 32-wide ones
    return out_channels >= U64_NT && (ceil_div(out_channels, 32) & 1) == 0;
}

__global__ void uf64_pack_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale, _Float16* __restrict__ packed,
                                 int OC, int Cin, int nchunks, long long total, int tr) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    constexpr int per_item = 2 * 2 * U64_NT * 8, per_chunk = UF_ITEMS * per_item;
    int r = (int)(e % per_chunk);
    const long long q = e / per_chunk;
    const int chunk = (int)(q % nchunks), nt = (int)(q / nchunks);
    const int item = r / per_item; r -= item * per_item;
    const int part = r / (2 * U64_NT * 8); r -= part * 2 * U64_NT * 8;
    const int g = r / (U64_NT * 8); r -= g * U64_NT * 8;
    const int n = r >> 3, k = r & 7;
    const int c = chunk * SP_CK + g * 8 + k, oc = nt * U64_NT + n;
    _Float16 h = (_Float16)0.f;
    if (c < Cin && oc < OC) {
        const UfItem t = uf_item(item);
        const float ws = uf_fold(w + ((long long)oc * Cin + c) * 9, t.py, t.dy, t.px, t.dx, tr) * (1.f / inv_scale[oc]);
        const _Float16 w1 = (_Float16)ws;
        h = part == 0 ? w1 : (_Float16)(ws - (float)w1);
    }
    packed[e] = h;
}

// PIN: the input is a pair tensor (staged by LDS-DMA, nothing to split); POUT: the output is written as one (the MFMA
// operands swap roles, so that a lane's accumulator registers run over FILTERS of one pixel: four consecutive channels
// = half a granule per store).
template <bool PIN, bool POUT, bool MIXED = false, bool ONE = false, bool KSPLIT = false>   // MIXED: p.nblocks whole tiles, then p.tp_nblocks transposed ones; ONE, KSPLIT: see conv3x3_split_kernel
__global__ __launch_bounds__(SP_THREADS, 1) void upconv2x_split64_kernel(const SplitConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (!MIXED || (int)blockIdx.x < p.nblocks) {
        constexpr bool TP = false;
        const int block = blockIdx.x, nblocks = p.nblocks, tilesX = p.tilesX, tilesY = p.tilesY;
#include "upconv64_split_body.inc"
    } else if constexpr (MIXED) {
        constexpr bool TP = true;
        const int block = (int)blockIdx.x - p.nblocks, nblocks = p.tp_nblocks, tilesX = 1, tilesY = p.tp_tilesY;
#include "upconv64_split_body.inc"
    }
}


// ------------------------------------------------------------------------------------------------------------------
// 1x1 stride-2 conv (+ LeakyReLU) on split operands: conv_fused of the KB block, reference src/net_utils.py:1337-1343 and
// :1366-1368 (cat[image, xyz, fused] -> Conv2d(kernel 1, stride 2)).  The tensor channels (image, fused: multiples of
// 16) go through the matrix core like the 3x3 kernels' -- one "tap", M = 32 output pixels of a row (input pixels
// (2y, 2x)), N = 32 filters, K = 16 channels -- the three backprojection channels K^-1 [x y 1]^T z, computed once per
// block by kb_xyz_s2_kernel, enter in fp32 in the epilogue (three FMAs per output).  Workgroup = 8 waves = 4 row groups
// x 2 filter groups, tile 8 rows x 32 pixels x 128 filters, main and small-term accumulators as in conv3x3_split_kernel.
// A chunk is only twelve MFMAs per wave, so the K loop is a short software pipeline: weights of chunk c+1 and inputs of
// chunk c+2 are issued at the top of chunk c (two register sets by chunk parity), the inputs of chunk c+1 are split and
// written to the other A buffer after the MFMAs of chunk c; one barrier per chunk; vmcnt waits count the loads in
// issue order (b(c) | inputs(c+1) | b(c+1) | inputs(c+2)).  Past the last chunk the fetches repeat the last chunk (never
// used) and the MFMAs are skipped: the vmcnt arithmetic is the same in every iteration and the kernel holds two copies
// of the body (seven tail variants spilled).  The loop is bound by memory latency, not by its MFMAs (a wave-private
// variant without LDS and barriers, every lane fetching its own fragment, measured 15 % slower: twice the loads).
template <int N, int NBX>
__device__ __forceinline__ void c1_wait_b(f32x4 (&b)[NBX][2]) {
    static_assert(NBX == 1 || NBX == 2, "one or two 32-filter blocks per wave");
    if constexpr (NBX == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]) : "n"(N));
    else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0][0]), "+v"(b[0][1]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void c1_wait_a(float (&v)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "n"(N));
}

template <int RG, int NB, bool ONE = false>   // RG row groups x 8 / RG filter groups of waves; a wave: 8 / RG rows x NB 32-filter blocks (RG * NB = 4: 128 filters per tile); ONE: h1 w1 alone (throughput only)
__global__ __launch_bounds__(SP_THREADS, 1) void conv1x1s2_split_kernel(const SplitConvParams p) {
    constexpr int TH = 8, MB = TH / RG, NT = 128, NPIX = TH * SP_TW;
    static_assert(32 * NB * (8 / RG) == NT, "128 filters per workgroup");
    constexpr int A_PART = 2 * NPIX * 16, A_BYTES = 2 * A_PART;          // [part][k-group][pixel][8 fp16]
    constexpr int B_CHUNK = 2 * 2 * NT * 16;                             // bytes: [part][k-group][filter][8 fp16]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0");      // fp16 results flush subnormals (see conv3x3_split_kernel)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave % RG, fg = wave / RG;
    const int lm = lane & 31, g = lane >> 5;
    int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int nt = bid % p.nTilesN;
    bid /= p.nTilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int oy0 = ty * TH, ox0 = tx * SP_TW;
    const int H = p.H, W = p.W, sH = p.sH, sW = p.sW;
    const long long plane = (long long)sH * sW;
    const int nchunks = p.Cin / SP_CK, last = nchunks - 1;
    float prescale, unscale;
    sp_act_scale(p, n, prescale, unscale);

    // staging: waves 0-3 take k-group 0 of a chunk, waves 4-7 k-group 1; one pixel per thread: output (r, c) reads input (2 r, 2 c)
    const int kg_st = wave >> 2, t256 = tid & 255;
    const int sy = 2 * (oy0 + (t256 >> 5)), sx = 2 * (ox0 + (t256 & 31));
    const int goff = (sy < sH && sx < sW) ? (sy * sW + sx) * 4 : -1;
    // a source that was written at the even pixels only (p.sub0: the stride-2 split conv's fp32 side output): same validity
    const int goff_sub = goff >= 0 ? ((sy >> 1) * W + (sx >> 1)) * 4 : -1;
    const long long plane_sub = (long long)H * W;
    const unsigned char* wp_nt = reinterpret_cast<const unsigned char*>(p.wp) + (long long)nt * nchunks * B_CHUNK;

    float va[2][8];
    auto load_chunk = [&](float (&v)[8], int chunk) {
        int c = chunk * SP_CK, s = 0;
        if (p.nsrc > 1 && c >= p.srcC[0]) { c -= p.srcC[0]; s = 1; }
        const bool sub = p.sub0 && s == 0;                 // launch- / wave-uniform
        const long long pl = sub ? plane_sub : plane;
        const float* base = p.src[s] + (long long)n * p.src_bstride[s] + (long long)(c + kg_st * 8) * pl;
        const int go = sub ? goff_sub : goff;
        const unsigned voff = go < 0 ? 0u : (unsigned)go;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float* sb = base + (long long)k * pl;   // wave-uniform
            asm volatile("global_load_dword %0, %1, %2" : "=v"(v[k]) : "v"(voff), "s"(sb) : "memory");
        }
    };
    auto store_chunk = [&](int buf, const float (&vin)[8]) {
        unsigned char* A = smem + buf * A_BYTES + kg_st * NPIX * 16;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = goff >= 0 ? vin[k] : 0.f;
        sph8 h1, h2;
        sp_split8(v, prescale, h1, h2);
        *reinterpret_cast<sph8*>(A + t256 * 16) = h1;
        *reinterpret_cast<sph8*>(A + A_PART + t256 * 16) = h2;
    };
    const unsigned boff = (unsigned)((g * NT + fg * 32 * NB + lm) * 16);
    auto load_b = [&](f32x4 (&b)[NB][2], int chunk) {
        const unsigned char* base = wp_nt + (long long)chunk * B_CHUNK;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned char* sb = base + (t * 2 * NT + nb * 32) * 16;   // wave-uniform
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[nb][t]) : "v"(boff), "s"(sb) : "memory");
            }
    };

    spf16 acc[MB][NB], lo[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[mb][nb][i] = 0.f; lo[mb][nb][i] = 0.f; }

    const unsigned char* const aptr = smem + (g * NPIX + MB * rg * SP_TW + lm) * 16;
    f32x4 bq[2][NB][2];
    constexpr int NLB = 2 * NB, NLA = 8;   // loads per weight fetch / per input fetch
    auto body = [&](int c, auto par_tag) {   // PAR: parity of c (register sets, A buffer)
        constexpr int PAR = decltype(par_tag)::value;
        load_b(bq[PAR ^ 1], min(c + 1, last));
        load_chunk(va[PAR], min(c + 2, last));
        // outstanding, oldest first: b(c) | inputs(c+1) | b(c+1) | inputs(c+2)
        c1_wait_b<NLA + NLB + NLA>(bq[PAR]);
        if (c <= last) {
            sph8 bw[NB][2], a[MB][2];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                bw[nb][0] = __builtin_bit_cast(sph8, bq[PAR][nb][0]);
                bw[nb][1] = __builtin_bit_cast(sph8, bq[PAR][nb][1]);
            }
            const int abuf = PAR * A_BYTES;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int t = 0; t < 2; ++t) a[mb][t] = *reinterpret_cast<const sph8*>(aptr + abuf + t * A_PART + mb * SP_TW * 16);
            constexpr int TA[3] = {0, 0, 1}, TBP[3] = {0, 1, 0};   // h1 w1 | h1 (w2 2^11), h2 w1
#pragma unroll
            for (int t = 0; t < (ONE ? 1 : 3); ++t)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        spf16& d = t > 0 ? lo[mb][nb] : acc[mb][nb];
                        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb][TA[t]], bw[nb][TBP[t]], d, 0, 0, 0);
                    }
        }
        c1_wait_a<NLB + NLA>(va[PAR ^ 1]);   // inputs of chunk c+1 (fetched a chunk and a half ago)
        store_chunk(PAR ^ 1, va[PAR ^ 1]);
        __syncthreads();
    };

    load_chunk(va[0], 0);
    c1_wait_a<0>(va[0]);
    store_chunk(0, va[0]);
    load_b(bq[0], 0);
    load_chunk(va[1], min(1, last));
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
        body(c, std::integral_constant<int, 0>{});
        body(c + 1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the repeated fetches of the last iterations land before their registers are reused

    // ---- epilogue: acc[mb][nb][i]: pixel x = 8 (i / 4) + 4 g + (i % 4) of row MB rg + mb, filter fg * 64 + nb * 32 + lm
    const long long oplane = (long long)H * W;
    float* outn = p.out + (long long)n * p.out_bstride;
    const float* xyzn = p.xyz ? p.xyz + (long long)n * p.xyz_bstride : nullptr;
    const float slope = p.act ? p.slope : 1.f;
    const bool vec4 = p.vec4 != 0;
    float amax = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int oc = nt * NT + fg * 32 * NB + nb * 32 + lm;
        const float inv = p.inv_scale[oc] * unscale;               // 2^-e 2^-k; the table is padded to whole n-tiles
        if (oc >= p.OC) continue;
        float wx[3] = {0.f, 0.f, 0.f};
        if (xyzn) { wx[0] = p.wxyz[oc * 3]; wx[1] = p.wxyz[oc * 3 + 1]; wx[2] = p.wxyz[oc * 3 + 2]; }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int Y = oy0 + MB * rg + mb;
            if (Y >= H) continue;
            float* o = outn + (long long)oc * oplane + (long long)Y * W + ox0 + 4 * g;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int X = ox0 + 8 * q4 + 4 * g;
                if (X >= W) continue;
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = __builtin_fmaf(lo[mb][nb][q4 * 4 + j], 0.00048828125f, acc[mb][nb][q4 * 4 + j]) * inv;
                    if (xyzn && X + j < W) {
                        const float* xp = xyzn + (long long)Y * W + X + j;
                        t += wx[0] * xp[0] + wx[1] * xp[oplane] + wx[2] * xp[2 * oplane];
                    }
                    v[j] = t > 0.f ? t : t * slope;
                }
                if (vec4) {
                    *reinterpret_cast<f32x4*>(o + 8 * q4) = v;
                    amax = sp_amax4(amax, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (X + j < W) { o[8 * q4 + j] = v[j]; amax = fmaxf(amax, fabsf(v[j])); }
                }
            }
        }
    }
    if (p.out_amax) absmax_commit(p.out_amax + n, amax);
}

// the KB block's backprojection at the positions its stride-2 1x1 conv reads: xyz[:, j, y, x] = (K^-1 [2x 2y 1]^T)_j z,
// z = act(proj_weight . depth[:, 2y, 2x]) (reference src/net_utils.py:1352-1359; the same expressions as the in-kernel
// synthesis of the fp32 conv kernels, conv_dma_impl.h)
__global__ void kb_xyz_s2_kernel(const float* __restrict__ depth, long long dbs, int Cd, int H, int W, const float* __restrict__ proj,
                                 const float* __restrict__ kinv, int act, float slope, float* __restrict__ xyz, long long xbs,
                                 int oh, int ow) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    if (idx >= oh * ow) return;
    const int oy = idx / ow, ox = idx - oy * ow;
    const int Y = 2 * oy, X = 2 * ox;
    const long long HW = (long long)H * W;
    const float* db = depth + (long long)n * dbs + (long long)Y * W + X;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int c = 0;
    for (; c + 3 < Cd; c += 4) {
        a0 = fmaf(proj[c], db[(long long)c * HW], a0);
        a1 = fmaf(proj[c + 1], db[(long long)(c + 1) * HW], a1);
        a2 = fmaf(proj[c + 2], db[(long long)(c + 2) * HW], a2);
        a3 = fmaf(proj[c + 3], db[(long long)(c + 3) * HW], a3);
    }
    for (; c < Cd; ++c) a0 = fmaf(proj[c], db[(long long)c * HW], a0);
    const float a = (a0 + a1) + (a2 + a3);
    const float z = act ? leaky_relu(a, slope) : a;
    const float* ki = kinv + (long long)n * 9;
    float* o = xyz + (long long)n * xbs + idx;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        o[(long long)j * oh * ow] = (fmaf(ki[j * 3 + 1], (float)Y, ki[j * 3 + 0] * (float)X) + ki[j * 3 + 2]) * z;
}

// Sum of the split-K partial planes (conv3x3_split_kernel<.., KSPLIT>), in split order, + activation + the frame's max |out|:
// ws [ksplit][n][OC][H W] -> out (frames out_bstride apart).  A thread owns `kr` items of four consecutive pixels (VEC) or of one, 256
// threads apart (the launcher keeps about 512 blocks per frame, at most 8 items per thread); the block's maximum meets in LDS so that
// ONE wave per block touches the frame's slot (a commit per wave is 13 k agent-scope accesses of one address for deconv2's conv: 110 us
// of a 60 us launch).
template <bool VEC>
__global__ __launch_bounds__(256) void ksplit_reduce_kernel(const float* __restrict__ ws, long long ks_stride, int ksplit, float* __restrict__ out,
                                                            long long out_bstride, long long per_frame, float slope, unsigned* __restrict__ out_amax, int kr) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    float m = 0.f;
    for (int r = 0; r < kr; ++r) {
        const long long i = (((long long)blockIdx.x * kr + r) * 256 + threadIdx.x) * (VEC ? 4 : 1);
        if (i < per_frame) {
            const float* src = ws + (long long)n * per_frame + i;
            if constexpr (VEC) {
                f32x4 a = *reinterpret_cast<const f32x4*>(src);
                for (int k = 1; k < ksplit; ++k) a += *reinterpret_cast<const f32x4*>(src + (long long)k * ks_stride);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = a[j] > 0.f ? a[j] : a[j] * slope;
                *reinterpret_cast<f32x4*>(out + (long long)n * out_bstride + i) = a;
                m = sp_amax4(m, a);
            } else {
                float a = src[0];
                for (int k = 1; k < ksplit; ++k) a += src[(long long)k * ks_stride];
                a = a > 0.f ? a : a * slope;
                out[(long long)n * out_bstride + i] = a;
                m = fmaxf(m, fabsf(a));
            }
        }
    }
    if (out_amax) {   // launch-uniform
        const unsigned b = wave_max_bits(m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = __uint_as_float(b);
        __syncthreads();
        absmax_commit(out_amax + n, threadIdx.x < 64 ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : 0.f);
    }
}

__global__ void copy_wxyz_kernel(const float* __restrict__ w, float* __restrict__ wxyz, int OC, int cin_total, int xyz_offset) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < OC * 3) wxyz[e] = w[(long long)(e / 3) * cin_total + xyz_offset + e % 3];
}

}  // namespace kbn

extern "C" {

// filters per workgroup; the folded up-conv takes 16-filter tiles for narrow layers (upconv2x_split16_kernel)
static int split_nt(int mode, int out_channels, int in_channels) {
    if (mode == 3 && kbn::uf_narrow(out_channels, in_channels)) return kbn::U16_NT;
    // a function of the layer's shape ONLY: the blob layout follows it, so no run-time knob may enter here (a switch read
    // at pack time and again at launch time would let kbn_reload_env() in between walk a blob with the wrong tiling)
    if (mode == 3 && kbn::uf_wide(out_channels)) return kbn::U64_NT;
    return mode == 2 ? 128 : (mode == 3 ? kbn::UF_NT : 64);
}

size_t kbn_conv3x3_split_packed_weight_bytes(int out_channels, int in_channels, int mode) {
    using namespace kbn;
    if (mode == 4) mode = 3;   // the transposed conv runs the folded up-conv's kernels on its own folded weights: same blob layout
    if (out_channels < 1 || in_channels < 1 || (in_channels % SP_CK) != 0 || mode < 0 || mode > 3) return 0;
    const int nt = split_nt(mode, out_channels, in_channels), tiles = ceil_div(out_channels, nt);
    return (size_t)tiles * nt * 4 + (size_t)tiles * (in_channels / SP_CK) * ((mode == 3 ? UF_ITEMS : 9) * 2 * 2 * nt * 16)   // per 16 channels: [set][part][2 k-groups][nt][8] fp16
           + (size_t)(in_channels / SP_CK) * 4;                                                                                  // the bound table of the pair format (split_l1_kernel)
}

int kbn_conv3x3_split_pack_weight(const float* weight, void* packed, int out_channels, int in_channels, int mode,
                                  kbn_stream_t stream) {
    using namespace kbn;
    const size_t bytes = kbn_conv3x3_split_packed_weight_bytes(out_channels, in_channels, mode);
    if (!weight || !packed || bytes == 0) return KBN_ERR_INVALID_ARGUMENT;
    const int tr = mode == 4 ? 1 : 0;   // ConvTranspose2d taps instead of the nearest-2x fold (uf_fold)
    if (tr) mode = 3;
    const int nt = split_nt(mode, out_channels, in_channels), ocpad = ceil_div(out_channels, nt) * nt;
    float* inv = static_cast<float*>(packed);
    _Float16* wp = reinterpret_cast<_Float16*>(inv + ocpad);
    const size_t l1_bytes = (size_t)(in_channels / SP_CK) * 4;
    const long long total = (long long)((bytes - (size_t)ocpad * 4 - l1_bytes) / 2);
    hipLaunchKernelGGL(split_l1_kernel, dim3(in_channels / SP_CK), dim3(256), 0, (hipStream_t)stream, weight,
                       reinterpret_cast<float*>(static_cast<unsigned char*>(packed) + bytes - l1_bytes), out_channels, in_channels, 9);
    if (mode == 3) {
        hipLaunchKernelGGL(uf_scale_kernel, dim3(ocpad), dim3(256), 0, (hipStream_t)stream, weight, inv, out_channels, in_channels, tr);
        if (uf_narrow(out_channels, in_channels)) {
            hipLaunchKernelGGL(uf16_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, inv,
                               wp, out_channels, in_channels, in_channels / U16_CK, total, tr);
            KBN_CHECK_LAUNCH();
            return KBN_OK;
        }
        if (nt == U64_NT) {
            hipLaunchKernelGGL(uf64_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, inv,
                               wp, out_channels, in_channels, in_channels / SP_CK, total, tr);
            KBN_CHECK_LAUNCH();
            return KBN_OK;
        }
        hipLaunchKernelGGL(uf_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, inv, wp,
                           out_channels, in_channels, in_channels / SP_CK, total, tr);
        KBN_CHECK_LAUNCH();
        return KBN_OK;
    }
    hipLaunchKernelGGL(split_scale_kernel, dim3(ocpad), dim3(256), 0, (hipStream_t)stream, weight, inv, out_channels, in_channels * 9);
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight,
                       inv, wp, out_channels, in_channels, in_channels / SP_CK, nt, total, 9, in_channels, 0);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_absmax_frames(const float* x, long long batch_stride, int n, long long per_frame, unsigned* slots, kbn_stream_t stream) {
    return kbn::absmax_frames_launch(x, batch_stride, n, per_frame, slots, (hipStream_t)stream);
}

static int conv3x3_split_impl(const kbn_conv_src* srcs, int n_src, const void* packed_weight, float* out,
                              long long out_batch_stride, int n, int out_channels, int height, int width, int mode,
                              int act_exponent, int apply_activation, float negative_slope, unsigned* out_absmax,
                              void* pair_out, long long pair_out_batch_stride, float* pair_out_scale, int ksplit, float* workspace,
                              kbn_stream_t stream) {
    using namespace kbn;
    if (act_exponent < -60 || act_exponent > 60) return KBN_ERR_INVALID_ARGUMENT;
    if (!srcs || n_src < 1 || n_src > 2 || !packed_weight || (!out && !pair_out) || n < 1 || out_channels < 1 || height < 1 || width < 1)
        return KBN_ERR_INVALID_ARGUMENT;
    if (mode == 4) mode = 3;   // ConvTranspose2d(3, stride 2, padding 1, output_padding 1): the folded kernels on a blob packed with mode 4
    if (mode < 0 || mode > 3) return KBN_ERR_INVALID_ARGUMENT;
    if (knob(KNOB_NO_SPLIT)) return KBN_ERR_UNSUPPORTED;
    const bool vec4 = pair_out || !((width & 3) || (reinterpret_cast<uintptr_t>(out) & 15) || (out_batch_stride & 3));
    if (!vec4 && (mode == 1 || mode == 3)) return KBN_ERR_UNSUPPORTED;   // the up-convs only store whole quads
    if ((mode == 1 || mode == 3) && (n_src != 1 || (height & 1) || (width & 1))) return KBN_ERR_UNSUPPORTED;
    SplitConvParams p{};
    int cin = 0;
    for (int s = 0; s < n_src; ++s) {
        const kbn_conv_src& a = srcs[s];
        const bool pair = a.kind == KBN_SRC_PAIR;
        if ((a.kind != KBN_SRC_TENSOR && !pair) || !a.data || a.channels < 1 || (a.channels % SP_CK) != 0) return KBN_ERR_UNSUPPORTED;
        if (s == 0) { p.sH = a.src_height; p.sW = a.src_width; }
        if (a.src_height != p.sH || a.src_width != p.sW) return KBN_ERR_INVALID_ARGUMENT;
        if (pair) {   // source 0 of the concat conv / the input of a folded up-conv, written by a split-operand producer
            if (s != 0 || mode == 1) return KBN_ERR_UNSUPPORTED;
            if (mode == 2 && n_src != 1) return KBN_ERR_UNSUPPORTED;
            // the concat kernel's K loop: a pair source beside an fp32 one, at least two 16-channel chunks each
            if (mode == 0 && (n_src != 2 || a.channels < 2 * SP_CK || srcs[1].kind != KBN_SRC_TENSOR || srcs[1].channels < 2 * SP_CK))
                return KBN_ERR_UNSUPPORTED;
            if (!a.scale || (reinterpret_cast<uintptr_t>(a.data) & 15) || (a.batch_stride & 7) ||
                a.batch_stride < (long long)(a.channels / 8) * 2 * pair_plane_halves(a.src_height, a.src_width))
                return KBN_ERR_INVALID_ARGUMENT;
            if ((long long)a.src_height * a.src_width >= 0x0fffffffLL) return KBN_ERR_UNSUPPORTED;   // 32-bit DMA offsets
            p.pair_src = reinterpret_cast<const _Float16*>(a.data); p.pair_src_bstride = a.batch_stride; p.pair_src_scale = a.scale;
        } else {
            p.src[s] = a.data; p.src_bstride[s] = a.batch_stride;
        }
        p.srcC[s] = a.channels;
        cin += a.channels;
    }
    if (p.pair_src && !p.src[0]) { p.src[0] = p.src[1]; p.src_bstride[0] = p.src_bstride[1]; }   // never dereferenced: the kernels stage source 0 by DMA
    const bool dims_ok = mode == 0 ? (p.sH == height && p.sW == width)
                       : mode != 2 ? (2 * p.sH == height && 2 * p.sW == width)
                                   : (ceil_div(p.sH, 2) == height && ceil_div(p.sW, 2) == width);
    if (!dims_ok) return KBN_ERR_INVALID_ARGUMENT;
    if ((long long)p.sH * p.sW > 0x1fffffffLL || (long long)height * width > 0x1fffffffLL) return KBN_ERR_UNSUPPORTED;
    if (n_src == 1) { p.src[1] = p.src[0]; p.src_bstride[1] = p.src_bstride[0]; p.srcC[1] = 0; }
    p.nsrc = n_src;
    // the exponent follows the data when EVERY source brings its slots; otherwise the static act_exponent serves
    if (srcs[0].absmax && (n_src == 1 || srcs[1].absmax)) { p.amax[0] = srcs[0].absmax; p.amax[1] = n_src > 1 ? srcs[1].absmax : nullptr; }
    else if (p.pair_src && n_src > 1 && srcs[1].absmax) p.amax[1] = srcs[1].absmax;   // the fp32 source beside a pair source: its own window
    p.out_amax = out_absmax;
    const int ntf = split_nt(mode, out_channels, cin);
    p.nTilesN = ceil_div(out_channels, ntf);
    p.inv_scale = static_cast<const float*>(packed_weight);
    p.wp = reinterpret_cast<const _Float16*>(p.inv_scale + p.nTilesN * ntf);
    p.l1 = reinterpret_cast<const float*>(static_cast<const unsigned char*>(packed_weight) +
                                          kbn_conv3x3_split_packed_weight_bytes(out_channels, cin, mode) - (size_t)(cin / SP_CK) * 4);
    if (pair_out) {   // the output as a pair tensor: concat convs and the 64-filter folded up-convs; its 2^k needs every source's slot
        // the narrow up-conv writes 16 channels (two k-groups, zeros past out_channels): its pair tensor feeds the decoder tail
        const bool narrow = mode == 3 && uf_narrow(out_channels, cin) && !(knob(KNOB_DEBUG) & 128);
        const bool kernel_ok = (mode == 0) || (mode == 3 && ntf == U64_NT && !uf_narrow(out_channels, cin)) ||
                               (mode == 2 && n_src == 1) || narrow;
        if (!kernel_ok || (!narrow && (out_channels & 7))) return KBN_ERR_UNSUPPORTED;
        const int pair_channels = narrow ? U16_NT : out_channels;
        if (!p.amax[0] || (n_src > 1 && !p.amax[1]) || !pair_out_scale || (reinterpret_cast<uintptr_t>(pair_out) & 15) ||
            (pair_out_batch_stride & 7) || pair_out_batch_stride < (long long)(pair_channels / 8) * 2 * pair_plane_halves(height, width))
            return KBN_ERR_INVALID_ARGUMENT;
        p.pair_out = static_cast<_Float16*>(pair_out); p.pair_out_bstride = pair_out_batch_stride; p.pair_out_scale = pair_out_scale;
    }
    if (p.pair_src && mode == 3 && !(ntf == U64_NT || uf_narrow(out_channels, cin))) return KBN_ERR_UNSUPPORTED;   // the 32-filter up-conv kernel splits its inputs itself
    if (p.pair_src && mode == 3 && uf_narrow(out_channels, cin) && (srcs[0].channels % U16_CK)) return KBN_ERR_UNSUPPORTED;
    p.out = out; p.out_bstride = out_batch_stride;
    p.N = n; p.OC = out_channels; p.Cin = cin; p.H = height; p.W = width;
    p.tilesX = ceil_div(width, SP_TW); p.tilesY = ceil_div(height, mode == 2 ? SpGeom<2>::TH : SpGeom<0>::TH);
    if (mode == 3) { p.tilesX = ceil_div(p.sW, 32); p.tilesY = ceil_div(p.sH, 16); }   // tiles of 16 x 32 low-resolution pixels
    const long long blocks = (long long)p.tilesX * p.tilesY * n * p.nTilesN;
    if (blocks > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.nblocks = (int)blocks;
    p.act = apply_activation ? 1 : 0; p.slope = negative_slope;
    p.prescale = ldexpf(1.f, act_exponent); p.unscale = ldexpf(1.f, -act_exponent);
    p.vec4 = vec4 ? 1 : 0;
    p.ksplit = 1;
    if (ksplit > 1) {
        // The latency form: a layer whose tiles cannot fill the chip (deconv4's conv on ONE KITTI frame: 24 workgroups of 48 chunks each)
        // spreads every tile's K loop over `ksplit` workgroups; partial sums go to `workspace` [ksplit][n][out_channels][height x width]
        // and ksplit_reduce_kernel adds them in split order.  Another summation order than the one-workgroup form: same 1e-4 parity,
        // other low bits -- opt-in (KBNetModel.latency_mode), never mixed with the default form inside one model.
        const bool up64 = mode == 3 && ntf == U64_NT && !uf_narrow(out_channels, cin);   // the 64-filter folded up-conv
        if ((mode != 0 && mode != 2 && !up64) || p.pair_src || pair_out || !workspace || !out || knob(KNOB_FP16_ONE_TERM)) return KBN_ERR_UNSUPPORTED;
        const int nchunks = cin / SP_CK;
        if (ksplit > nchunks || (long long)(ceil_div(nchunks, ksplit)) * (ksplit - 1) >= nchunks) return KBN_ERR_INVALID_ARGUMENT;   // an empty range
        if (blocks * ksplit > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
        const long long per_frame = (long long)out_channels * height * width;
        SplitConvParams q = p;
        q.ksplit = ksplit;
        q.out = workspace; q.out_bstride = per_frame; q.ks_stride = per_frame * n;
        q.out_amax = nullptr; q.act = 0;
        q.vec4 = !((width & 3) || (reinterpret_cast<uintptr_t>(workspace) & 15)) ? 1 : 0;
        q.nblocks = (int)(blocks * ksplit);
        static DeviceOnce ok0, ok2, ok3;
        if (up64) {   // 8 x 32 low-resolution pixels x 64 filters per workgroup, whole tiles only
            q.tilesX = ceil_div(p.sW, 32); q.tilesY = ceil_div(p.sH, 8);
            const long long b64 = (long long)q.tilesX * q.tilesY * n * p.nTilesN * ksplit;
            if (b64 > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
            q.nblocks = (int)b64;
            auto kern = upconv2x_split64_kernel<false, false, false, false, true>;
            if (int r = set_max_dynamic_lds(ok3, reinterpret_cast<const void*>(kern), 160 * 1024)) return r;
            hipLaunchKernelGGL(kern, dim3(q.nblocks), dim3(SP_THREADS), 2 * (2 * 2 * 10 * 34 * 16) + 2 * (8 * 2 * 2 * 64 * 16), (hipStream_t)stream, q);
        } else if (mode == 0) {
            auto kern = conv3x3_split_kernel<0, 8, true, true, 2, false, false, false, false, true>;
            if (int r = set_max_dynamic_lds(ok0, reinterpret_cast<const void*>(kern), 160 * 1024)) return r;
            hipLaunchKernelGGL(kern, dim3(q.nblocks), dim3(SP_THREADS), SpGeom<0>::LDS + 2 * 9 * 2 * 2 * 64 * 16, (hipStream_t)stream, q);
        } else {
            auto kern = conv3x3_split_kernel<2, 2, true, false, 1, false, false, false, false, true>;
            if (int r = set_max_dynamic_lds(ok2, reinterpret_cast<const void*>(kern), 160 * 1024)) return r;
            hipLaunchKernelGGL(kern, dim3(q.nblocks), dim3(SP_THREADS), SpGeom<2>::LDS, (hipStream_t)stream, q);
        }
        KBN_CHECK_LAUNCH();
        const float slope = apply_activation ? negative_slope : 1.f;
        const bool vec = !(per_frame & 3) && !(out_batch_stride & 3) && !((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(workspace)) & 15);
        const long long items = vec ? per_frame / 4 : per_frame;
        long long kr = items / (256LL * 512);
        kr = kr < 1 ? 1 : (kr > 8 ? 8 : kr);
        const dim3 grid((unsigned)((items + 256 * kr - 1) / (256 * kr)), (unsigned)n);
        if (vec) hipLaunchKernelGGL(ksplit_reduce_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, workspace, q.ks_stride, ksplit, out, out_batch_stride, per_frame, slope, out_absmax, (int)kr);
        else hipLaunchKernelGGL(ksplit_reduce_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, workspace, q.ks_stride, ksplit, out, out_batch_stride, per_frame, slope, out_absmax, (int)kr);
        KBN_CHECK_LAUNCH();
        return KBN_OK;
    }
    // THROUGHPUT-ONLY (KBN_FP16_ONE_TERM=1): the concat convs, the 64-filter folded up-convs and the stride-2 convs issue h1 w1 alone
    const bool one_term = knob(KNOB_FP16_ONE_TERM) != 0;
    auto launch = [&](auto kern, size_t lds, DeviceOnce& once) -> int {
        if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
        hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(SP_THREADS), lds, (hipStream_t)stream, p);
        return KBN_OK;
    };
    static DeviceOnce o[5];
    int rc;
    if (mode == 3 && uf_narrow(out_channels, cin)) {
        if (knob(KNOB_DEBUG) & 128) {   // A/B: 16-row tiles, one workgroup per CU
            static DeviceOnce o8p;
            rc = p.pair_src ? launch(upconv2x_split16_kernel<8, true>, 2 * 2 * 4 * ((18 * 34 + 15) / 16 * 16) * 16, o8p)
                            : launch(upconv2x_split16_kernel<8, false>, 2 * 2 * 4 * ((18 * 34 + 15) / 16 * 16) * 16, o[4]);
        } else {             // 8 x 32 low-resolution pixels per workgroup of 4 waves, two workgroups per CU
            p.tilesY = ceil_div(p.sH, 8);
            const long long blocks8 = (long long)p.tilesX * p.tilesY * n * p.nTilesN;
            if (blocks8 > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
            p.nblocks = (int)blocks8;
            static DeviceOnce o16, o16p, o16o, o16po;
            auto launch4 = [&](auto kern, DeviceOnce& once) -> int {
                if (int r = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return r;
                hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), 2 * 4 * ((10 * 34 + 15) / 16 * 16) * 16, (hipStream_t)stream, p);
                return KBN_OK;
            };
            if (one_term && p.pair_src && p.pair_out) {   // THROUGHPUT-ONLY: the decoder's pair chain in one-term mode (the shipped form of deconv0's up-conv)
                static DeviceOnce o16po1;
                rc = launch4(upconv2x_split16_kernel<4, true, true, true>, o16po1);
            } else if (p.pair_out) rc = p.pair_src ? launch4(upconv2x_split16_kernel<4, true, true>, o16po) : launch4(upconv2x_split16_kernel<4, false, true>, o16o);
            else rc = p.pair_src ? launch4(upconv2x_split16_kernel<4, true>, o16p) : launch4(upconv2x_split16_kernel<4, false>, o16);
        }
        if (rc != KBN_OK) return rc;
        KBN_CHECK_LAUNCH();
        return KBN_OK;
    }
    if (mode == 3 && ntf == U64_NT) {   // 8 x 32 low-resolution pixels x 64 filters per workgroup
        p.tilesX = ceil_div(p.sW, 32); p.tilesY = ceil_div(p.sH, 8);
        const long long blocks64 = (long long)p.tilesX * p.tilesY * n * p.nTilesN;
        if (blocks64 > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
        p.nblocks = (int)blocks64;
        constexpr size_t lds64 = 2 * (2 * 2 * 10 * 34 * 16) + 2 * (8 * 2 * 2 * 64 * 16);
        // a low-resolution map whose width leaves 1-16 columns behind the whole 32-column tiles: transposed tiles (16 rows x 16
        // columns) for that column, in the same launch (KBN_DEBUG & 512: off)
        const int wrem = p.sW % 32;
        const bool tp64 = p.sW >= 32 && wrem >= 1 && wrem <= 16 && !(knob(KNOB_DEBUG) & 512);
        if (tp64) {
            p.tilesX = p.sW / 32;
            p.nblocks = p.tilesX * p.tilesY * n * p.nTilesN;
            p.tp_x0 = 32 * p.tilesX;
            p.tp_tilesY = ceil_div(p.sH, 16);
            p.tp_nblocks = p.tp_tilesY * n * p.nTilesN;
        }
        auto launch64 = [&](auto kern, DeviceOnce& once) -> int {
            if (int r = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return r;
            hipLaunchKernelGGL(kern, dim3(p.nblocks + (tp64 ? p.tp_nblocks : 0)), dim3(SP_THREADS), lds64, (hipStream_t)stream, p);
            return KBN_OK;
        };
        auto pick64 = [&](auto one_tag) -> int {   // (static locals of a generic lambda: one set per instantiation)
            constexpr bool ONE = decltype(one_tag)::value;
            static DeviceOnce o64[4], o64t[4];
            if (tp64) {
                if (p.pair_src) return p.pair_out ? launch64(upconv2x_split64_kernel<true, true, true, ONE>, o64t[3]) : launch64(upconv2x_split64_kernel<true, false, true, ONE>, o64t[2]);
                return p.pair_out ? launch64(upconv2x_split64_kernel<false, true, true, ONE>, o64t[1]) : launch64(upconv2x_split64_kernel<false, false, true, ONE>, o64t[0]);
            }
            if (p.pair_src) return p.pair_out ? launch64(upconv2x_split64_kernel<true, true, false, ONE>, o64[3]) : launch64(upconv2x_split64_kernel<true, false, false, ONE>, o64[2]);
            return p.pair_out ? launch64(upconv2x_split64_kernel<false, true, false, ONE>, o64[1]) : launch64(upconv2x_split64_kernel<false, false, false, ONE>, o64[0]);
        };
        rc = one_term ? pick64(std::true_type{}) : pick64(std::false_type{});
        if (rc != KBN_OK) return rc;
        KBN_CHECK_LAUNCH();
        return KBN_OK;
    }
    switch (mode) {
        case 0:
            // (a 16x16x32 form of this kernel was built and measured in round 2 -- profiles/r02/HISTORY.md: 1-7 % slower
            // inside the forward -- and removed in round 6)
            {
                constexpr size_t lds0 = SpGeom<0>::LDS + 2 * 9 * 2 * 2 * 64 * 16;
                // a map whose width leaves 1-16 columns behind the whole 32-column tiles: that column goes to a launch of
                // transposed tiles (32 rows x 16 columns), half the MFMAs of the tiles it replaces (KBN_DEBUG & 512: off)
                const int wrem = width % SP_TW;
                const bool tp = width >= SP_TW && wrem >= 1 && wrem <= 16 && !(knob(KNOB_DEBUG) & 512);
                if (tp) {
                    p.tilesX = width / SP_TW;
                    p.nblocks = p.tilesX * p.tilesY * n * p.nTilesN;
                    p.tp_x0 = SP_TW * p.tilesX;
                    p.tp_tilesY = ceil_div(height, 2 * SpGeom<0>::TH);
                    p.tp_nblocks = p.tp_tilesY * n * p.nTilesN;
                }
                auto launch0 = [&](auto kern, DeviceOnce& once) -> int {
                    if (int r = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return r;
                    hipLaunchKernelGGL(kern, dim3(p.nblocks + (tp ? p.tp_nblocks : 0)), dim3(SP_THREADS), lds0, (hipStream_t)stream, p);
                    return KBN_OK;
                };
                auto pick0 = [&](auto one_tag) -> int {
                    constexpr bool ONE = decltype(one_tag)::value;
                    static DeviceOnce o0p[4], o0t[4];
                    if (tp) {
                        if (p.pair_src) return p.pair_out ? launch0(conv3x3_split_kernel<0, 8, true, true, 2, true, true, true, ONE>, o0t[3])
                                                          : launch0(conv3x3_split_kernel<0, 8, true, true, 2, true, false, true, ONE>, o0t[2]);
                        return p.pair_out ? launch0(conv3x3_split_kernel<0, 8, true, true, 2, false, true, true, ONE>, o0t[1])
                                          : launch0(conv3x3_split_kernel<0, 8, true, true, 2, false, false, true, ONE>, o0t[0]);
                    }
                    if (p.pair_src) return p.pair_out ? launch0(conv3x3_split_kernel<0, 8, true, true, 2, true, true, false, ONE>, o0p[3])
                                                      : launch0(conv3x3_split_kernel<0, 8, true, true, 2, true, false, false, ONE>, o0p[2]);
                    return p.pair_out ? launch0(conv3x3_split_kernel<0, 8, true, true, 2, false, true, false, ONE>, o0p[1])
                                      : launch0(conv3x3_split_kernel<0, 8, true, true, 2, false, false, false, ONE>, o0p[0]);
                };
                rc = one_term ? pick0(std::true_type{}) : pick0(std::false_type{});
            }
            break;
        case 1: return KBN_ERR_UNSUPPORTED;   // the nine-tap up-conv form (rounds 2-5; superseded by the folded form, mode 3) is no longer built
        case 2:
            // 2 row groups x 4 filter groups of waves (a wave: 4 rows x ONE 32-filter block): every weight fragment is fetched from
            // L2 by two waves instead of four -- half the vector-memory traffic of a chunk -- for twice the A fragment reads
            // from LDS.  Inside the forward (KB2 / KB3 / KB4 / conv5 image / conv5 depth, 32 KITTI frames): 397 / 337 / 314 /
            // 247 / 60 us against 414 / 360 / 322 / 281 / 69 with 4 x 2 waves of 2 rows x two blocks
            {
                auto pick2 = [&](auto one_tag) -> int {
                    constexpr bool ONE = decltype(one_tag)::value;
                    static DeviceOnce o2p[4];
                    if (p.pair_src) return p.pair_out ? launch(conv3x3_split_kernel<2, 2, true, false, 1, true, true, false, ONE>, SpGeom<2>::LDS, o2p[3])
                                                      : launch(conv3x3_split_kernel<2, 2, true, false, 1, true, false, false, ONE>, SpGeom<2>::LDS, o2p[2]);
                    return p.pair_out ? launch(conv3x3_split_kernel<2, 2, true, false, 1, false, true, false, ONE>, SpGeom<2>::LDS, o2p[1])
                                      : launch(conv3x3_split_kernel<2, 2, true, false, 1, false, false, false, ONE>, SpGeom<2>::LDS, o2p[0]);
                };
                rc = one_term ? pick2(std::true_type{}) : pick2(std::false_type{});
            }
            break;
        default:
            // two A buffers (18 x 34 pixels x 16 channels x two fp16 terms) + two buffers of sixteen weight sets
            rc = launch(upconv2x_split_kernel<true>, 2 * (2 * 2 * 18 * 34 * 16) + 2 * UF_ITEMS * (2 * 2 * UF_NT * 16), o[3]);
            break;
    }
    if (rc != KBN_OK) return rc;
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}


int kbn_conv3x3_split_forward(const kbn_conv_src* srcs, int n_src, const void* packed_weight, float* out,
                              long long out_batch_stride, int n, int out_channels, int height, int width, int mode,
                              int act_exponent, int apply_activation, float negative_slope, unsigned* out_absmax,
                              void* pair_out, long long pair_out_batch_stride, float* pair_out_scale, kbn_stream_t stream) {
    return conv3x3_split_impl(srcs, n_src, packed_weight, out, out_batch_stride, n, out_channels, height, width, mode, act_exponent,
                              apply_activation, negative_slope, out_absmax, pair_out, pair_out_batch_stride, pair_out_scale, 1, nullptr, stream);
}

int kbn_conv3x3_split_forward_ksplit(const kbn_conv_src* srcs, int n_src, const void* packed_weight, float* out,
                                     long long out_batch_stride, int n, int out_channels, int height, int width, int mode,
                                     int act_exponent, int apply_activation, float negative_slope, unsigned* out_absmax,
                                     int ksplit, float* workspace, kbn_stream_t stream) {
    if (ksplit < 1) return KBN_ERR_INVALID_ARGUMENT;
    return conv3x3_split_impl(srcs, n_src, packed_weight, out, out_batch_stride, n, out_channels, height, width, mode, act_exponent,
                              apply_activation, negative_slope, out_absmax, nullptr, 0, nullptr, ksplit, workspace, stream);
}


// ---- 1x1 stride-2 conv on split operands (conv_fused of the KB block) --------------------------------------------------
// blob: [inv_scale: tiles x 128 floats][fp16 panel: tiles x (cin / 16) x 8 KiB][wxyz: out_channels x 3 floats, if any]
static size_t c1_panel_bytes(int out_channels, int cin) {
    return (size_t)kbn::ceil_div(out_channels, 128) * (128 * 4 + (size_t)(cin / kbn::SP_CK) * (2 * 2 * 128 * 16));
}

size_t kbn_conv1x1s2_split_packed_weight_bytes(int out_channels, int tensor_channels, int has_xyz) {
    if (out_channels < 1 || tensor_channels < 1 || (tensor_channels % kbn::SP_CK) != 0) return 0;
    return c1_panel_bytes(out_channels, tensor_channels) + (has_xyz ? (size_t)out_channels * 3 * 4 : 0);
}

int kbn_conv1x1s2_split_pack_weight(const float* weight, void* packed, int out_channels, int in_channels, int xyz_offset,
                                    kbn_stream_t stream) {
    using namespace kbn;
    const bool has_xyz = xyz_offset >= 0;
    const int cin = in_channels - (has_xyz ? 3 : 0);
    if (!weight || !packed || (has_xyz && xyz_offset > cin) || kbn_conv1x1s2_split_packed_weight_bytes(out_channels, cin, has_xyz) == 0)
        return KBN_ERR_INVALID_ARGUMENT;
    const int ocpad = ceil_div(out_channels, 128) * 128;
    float* inv = static_cast<float*>(packed);
    _Float16* wp = reinterpret_cast<_Float16*>(inv + ocpad);
    const long long total = (long long)((c1_panel_bytes(out_channels, cin) - (size_t)ocpad * 4) / 2);
    // per-filter exponent over ALL input channels of the filter (the three fp32 ones can only make it more cautious)
    hipLaunchKernelGGL(split_scale_kernel, dim3(ocpad), dim3(256), 0, (hipStream_t)stream, weight, inv, out_channels, in_channels);
    hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, inv, wp,
                       out_channels, cin, cin / SP_CK, 128, total, 1, has_xyz ? xyz_offset : cin, has_xyz ? 3 : 0);
    if (has_xyz) {
        float* wxyz = reinterpret_cast<float*>(static_cast<unsigned char*>(packed) + c1_panel_bytes(out_channels, cin));
        hipLaunchKernelGGL(copy_wxyz_kernel, dim3(ceil_div(out_channels * 3, 256)), dim3(256), 0, (hipStream_t)stream, weight, wxyz,
                           out_channels, in_channels, xyz_offset);
    }
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_conv1x1s2_split_forward(const kbn_conv_src* srcs, int n_src, const void* packed_weight, const float* xyz,
                                long long xyz_batch_stride, float* out, long long out_batch_stride, int n, int out_channels,
                                int height, int width, int act_exponent, int apply_activation, float negative_slope,
                                unsigned* out_absmax, kbn_stream_t stream) {
    using namespace kbn;
    if (act_exponent < -60 || act_exponent > 60) return KBN_ERR_INVALID_ARGUMENT;
    if (!srcs || n_src < 1 || n_src > 2 || !packed_weight || !out || n < 1 || out_channels < 1 || height < 1 || width < 1)
        return KBN_ERR_INVALID_ARGUMENT;
    if (knob(KNOB_NO_SPLIT)) return KBN_ERR_UNSUPPORTED;
    SplitConvParams p{};
    int cin = 0;
    for (int s = 0; s < n_src; ++s) {
        const kbn_conv_src& a = srcs[s];
        if (a.kind != KBN_SRC_TENSOR || !a.data || a.channels < 1 || (a.channels % SP_CK) != 0) return KBN_ERR_UNSUPPORTED;
        // with two sources, source 0 may come pre-subsampled: height x width planes holding the pixels (2y, 2x) of the tensor
        // the reference's conv reads (the fp32 side output of kbn_conv3x3_split_forward(mode 2, pair_out))
        const bool sub = s == 0 && n_src == 2 && a.src_height == height && a.src_width == width &&
                         (srcs[1].src_height != height || srcs[1].src_width != width);
        if (sub) p.sub0 = 1;
        else {
            if (!p.sH) { p.sH = a.src_height; p.sW = a.src_width; }
            if (a.src_height != p.sH || a.src_width != p.sW) return KBN_ERR_INVALID_ARGUMENT;
        }
        p.src[s] = a.data; p.src_bstride[s] = a.batch_stride; p.srcC[s] = a.channels;
        cin += a.channels;
    }
    if (ceil_div(p.sH, 2) != height || ceil_div(p.sW, 2) != width) return KBN_ERR_INVALID_ARGUMENT;
    if ((long long)p.sH * p.sW > 0x1fffffffLL) return KBN_ERR_UNSUPPORTED;
    if (n_src == 1) { p.src[1] = p.src[0]; p.src_bstride[1] = p.src_bstride[0]; p.srcC[1] = 0; }
    p.nsrc = n_src;
    // the exponent follows the data when EVERY source brings its slots; otherwise the static act_exponent serves
    if (srcs[0].absmax && (n_src == 1 || srcs[1].absmax)) { p.amax[0] = srcs[0].absmax; p.amax[1] = n_src > 1 ? srcs[1].absmax : nullptr; }
    p.out_amax = out_absmax;
    p.nTilesN = ceil_div(out_channels, 128);
    p.inv_scale = static_cast<const float*>(packed_weight);
    p.wp = reinterpret_cast<const _Float16*>(p.inv_scale + p.nTilesN * 128);
    p.xyz = xyz; p.xyz_bstride = xyz_batch_stride;
    p.wxyz = xyz ? reinterpret_cast<const float*>(static_cast<const unsigned char*>(packed_weight) + c1_panel_bytes(out_channels, cin)) : nullptr;
    p.out = out; p.out_bstride = out_batch_stride;
    p.N = n; p.OC = out_channels; p.Cin = cin; p.H = height; p.W = width;
    p.tilesX = ceil_div(width, SP_TW); p.tilesY = ceil_div(height, 8);
    const long long blocks = (long long)p.tilesX * p.tilesY * n * p.nTilesN;
    if (blocks > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.nblocks = (int)blocks;
    p.act = apply_activation ? 1 : 0; p.slope = negative_slope;
    p.prescale = ldexpf(1.f, act_exponent); p.unscale = ldexpf(1.f, -act_exponent);
    p.vec4 = !((width & 3) || (reinterpret_cast<uintptr_t>(out) & 15) || (out_batch_stride & 3)) ? 1 : 0;
    // 2 row groups x 4 filter groups (a wave: 4 rows x one 32-filter block): every weight fragment is fetched by two waves
    // instead of four (KBN_DEBUG & 128: the 4 x 2 form, for A/B runs)
    if (knob(KNOB_DEBUG) & 128) hipLaunchKernelGGL((conv1x1s2_split_kernel<4, 2>), dim3(p.nblocks), dim3(SP_THREADS), 2 * 2 * 2 * 256 * 16, (hipStream_t)stream, p);
    else if (knob(KNOB_FP16_ONE_TERM))   // THROUGHPUT-ONLY: h1 w1 alone
        hipLaunchKernelGGL((conv1x1s2_split_kernel<2, 1, true>), dim3(p.nblocks), dim3(SP_THREADS), 2 * 2 * 2 * 256 * 16, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv1x1s2_split_kernel<2, 1>), dim3(p.nblocks), dim3(SP_THREADS), 2 * 2 * 2 * 256 * 16, (hipStream_t)stream, p);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_kb_xyz_s2_forward(const float* depth, long long depth_batch_stride, int depth_channels, int height, int width,
                          const float* proj_weight, const float* kinv, int apply_activation, float negative_slope, float* xyz,
                          long long xyz_batch_stride, int n, kbn_stream_t stream) {
    using namespace kbn;
    if (!depth || !proj_weight || !kinv || !xyz || n < 1 || depth_channels < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    const int oh = ceil_div(height, 2), ow = ceil_div(width, 2);
    hipLaunchKernelGGL(kb_xyz_s2_kernel, dim3(ceil_div(oh * ow, 256), n), dim3(256), 0, (hipStream_t)stream, depth, depth_batch_stride,
                       depth_channels, height, width, proj_weight, kinv, apply_activation ? 1 : 0, negative_slope, xyz,
                       xyz_batch_stride, oh, ow);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

}  // extern "C"
