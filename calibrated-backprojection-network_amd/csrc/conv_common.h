// conv_common.h -- types shared by the two implicit-GEMM conv kernels
// (conv_igemm.hip: register-staged, any shape; conv_dma.hip: LDS-DMA staged, aligned shapes).
#pragma once

#include <array>
#include <functional>

#include "kbn_common.h"

// Second __launch_bounds__ argument of the MFMA conv kernels (minimum waves per SIMD): 2 lets the
// register allocator keep the accumulators in arch VGPRs (+3-6 % on the decoder convs vs the default).
#ifndef KBN_WAVES_PER_SIMD
#define KBN_WAVES_PER_SIMD 2
#endif

namespace kbn {

struct SrcDev {
    const float* data;
    const float* proj;
    const float* coords;
    const float* kinv;
    long long bstride;
    long long coords_bstride;
    int kind, C, H, W, Cd, cstart;
};

struct ConvParams {
    SrcDev src[KBN_MAX_SRC];
    const float* wp;
    float* out;
    long long out_bstride;
    int nsrc, N, OC, Ctot, Cpad;
    int inH, inW, outH, outW;
    int resize;
    int tilesX, tilesY, nTilesN, nblocks;
    int TWB, TH;
    int rowsS, colsS;
    int pitch, plane, PH;
    int act;
    float slope;
    int dbg;  // ablation switches for tools/conv_bench.py (KBN_DEBUG): 1 no A staging, 2 no B staging, 4 no MFMA
    int epi_lds;  // 1: store through LDS, one channel plane per store instruction (store-bound launches)
    unsigned* out_absmax;   // per-frame max |out| slots (kbn_common.h), or null
};

struct ConvPlan {
    int CK, NB, MW, nTilesN, Cpad, NT;
};

__host__ __device__ inline ConvPlan make_plan(int oc, int cin, int ks, int stride, int force_ck = 0) {
    ConvPlan pl;
    // Channels per K chunk (the packed weight layout depends on it, hence on the stride the
    // weight is used with).  Stride-2 3x3 convs stage 4x the pixels per output pixel: CK = 4
    // halves their LDS stage so that two workgroups stay resident per CU.  Narrow inputs (<= 16 channels:
    // conv0_depth, deconv0's 12 -> 12) also take 4: no K padding (12 = 3 x 4) and staging of chunk c+1
    // overlaps the MFMAs of chunk c even in a two- or three-chunk loop (-8 % / -20 % on those layers).
    pl.CK = (ks == 1) ? 16 : ((cin <= 16 || stride == 2) ? 4 : 8);
    if (force_ck == 4 && ks == 3) pl.CK = 4;  // (experiment hook of rounds 1-5; no caller passes it any more)
    if (force_ck == 8 && ks == 3 && cin > 4) pl.CK = 8;
    int nblk = ceil_div(oc, 16);
    // pick NB in 1..4 minimising padded n-blocks, ties -> larger NB
    int best = 1, bestpad = 1 << 30;
    for (int nb = 1; nb <= 4; ++nb) {
        int pad = ceil_div(nblk, nb) * nb;
        if (pad < bestpad || (pad == bestpad && nb > best)) { best = nb; bestpad = pad; }
    }
    pl.NB = best;
    // 16-channel outputs (NB = 1) measured 18% faster with MW = 4 than with 8 (tools/conv_bench.py):
    // their workgroups are staging-bound and a smaller tile keeps more of them in flight.
    pl.MW = (best >= 3 || best == 1) ? 4 : 8;
    pl.nTilesN = ceil_div(nblk, best);
    pl.NT = best * 16;
    pl.Cpad = round_up(cin, pl.CK);
    return pl;
}

__host__ __device__ constexpr int conv_maxpos(int KS, int STRIDE, int MW) {
    // staged positions per thread, worst case over the tile shapes conv2d_launch may pick
    return (KS == 1) ? (MW >= 4 ? MW / 4 : 1)
           : (STRIDE == 1) ? (MW == 1 ? 1 : (MW <= 4 ? 2 : 3))
                           : (MW == 1 ? 2 : (MW == 2 ? 3 : (MW == 4 ? 5 : 9)));
}

// Which source / plane feeds concat channel c (all wave-uniform -> scalar registers).
struct ChanRef {
    const float* ptr;  // tensor plane of this frame, or nullptr
    int kind;          // kbn_src_kind, or -1 for zero padding
    int j;             // channel index inside the source
};

__device__ __forceinline__ ChanRef chan_lookup(const ConvParams& p, int n, int c) {
    ChanRef r{nullptr, -1, 0};
#pragma unroll
    for (int s = 0; s < KBN_MAX_SRC; ++s) {
        if (s < p.nsrc && c >= p.src[s].cstart && c < p.src[s].cstart + p.src[s].C) {
            r.kind = p.src[s].kind;
            r.j = c - p.src[s].cstart;
            if (r.kind == KBN_SRC_TENSOR)
                r.ptr = p.src[s].data + (long long)n * p.src[s].bstride + (long long)r.j * (p.src[s].H * p.src[s].W);
        }
    }
    return r;
}


// Epilogue shared by the direct conv kernels: each lane holds 4 consecutive pixels (rows 4*(l>>4)+r of
// the m-block) of output channel (l&15) of each n-block; fused LeakyReLU; float4 stores.
struct StoreDst {
    float* out;
    long long out_bstride;
    int outH, outW, OC, TWB, act;
    float slope;
    unsigned* absmax = nullptr;   // per-frame max |out| slots, or null
};

template <int NB, int MW>
__device__ __forceinline__ void store_tile_dst(const StoreDst& p, const f32x4 (&acc)[MW][NB], int n, int nt,
                                               int oy0, int ox0, int wave, int li, int lk) {
    constexpr int NT = NB * 16;
    const int HWo = p.outH * p.outW;
    float* outn = p.out + (long long)n * p.out_bstride;
    const bool vec_ok = ((p.outW & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0) &&
                        ((p.out_bstride & 3) == 0);
    float amax = 0.f;   // max |stored value| of this thread (folded into p.absmax[n] below)
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        int mb = wave * MW + mi;
        int oyl = mb / p.TWB;
        int seg = mb - oyl * p.TWB;
        int oy = oy0 + oyl;
        int ox = ox0 + seg * 16 + lk * 4;
        if (oy >= p.outH || ox >= p.outW) continue;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            int oc = nt * NT + nb * 16 + li;
            if (oc >= p.OC) continue;
            f32x4 v = acc[mi][nb];
            if (p.act) {
                v[0] = leaky_relu(v[0], p.slope); v[1] = leaky_relu(v[1], p.slope);
                v[2] = leaky_relu(v[2], p.slope); v[3] = leaky_relu(v[3], p.slope);
            }
            float* o = outn + (long long)oc * HWo + (long long)oy * p.outW + ox;
            if (vec_ok && ox + 3 < p.outW) {
                *reinterpret_cast<f32x4*>(o) = v;
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ox + r < p.outW) { o[r] = v[r]; amax = fmaxf(amax, fabsf(v[r])); }
            }
        }
    }
    if (p.absmax) absmax_commit(p.absmax + n, amax);   // launch-uniform
}

// Store-bound launches (full-resolution layers with few input channels: conv0, deconv0) use this epilogue instead:
// the plain one above spreads every store instruction over 16 channel planes x 64 bytes (16 pages, half cache
// lines), measured 4.0 TB/s on conv0_image's 657 MB against 6.8 TB/s for a linear fill.  Here one n-block at a time
// goes through LDS ([16 channels][4*MW m-blocks][16 pixels], channel stride padded by 4 floats: conflict-free
// 128-bit writes), and comes back so that a wave's store instruction covers 16 consecutive m-blocks of ONE channel:
// whole cache lines, one page.  `scratch` = 16 * (64 * MW + 4) floats of the (now idle) stage buffers; all 256
// threads must call it.
template <int NB, int MW>
__device__ __forceinline__ void store_tile_lds(const StoreDst& p, const f32x4 (&acc)[MW][NB], int n, int nt,
                                               int oy0, int ox0, int wave, int li, int lk, float* scratch, int tid) {
    constexpr int NT = NB * 16;
    constexpr int CS = 64 * MW + 4;                       // channel stride in floats
    const int HWo = p.outH * p.outW;
    float* outn = p.out + (long long)n * p.out_bstride;
    const bool vec_ok = ((p.outW & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0) &&
                        ((p.out_bstride & 3) == 0);
    float amax = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        __syncthreads();                                  // the K loop's / the previous pass's LDS reads are done
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            f32x4 v = acc[mi][nb];
            if (p.act) {
                v[0] = leaky_relu(v[0], p.slope); v[1] = leaky_relu(v[1], p.slope);
                v[2] = leaky_relu(v[2], p.slope); v[3] = leaky_relu(v[3], p.slope);
            }
            *reinterpret_cast<f32x4*>(scratch + li * CS + (wave * MW + mi) * 16 + lk * 4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MW; ++j) {
            const int f = j * 256 + tid;                  // float4 index: [channel][m-block][quarter]
            const int c = f / (16 * MW), rem = f - c * (16 * MW);
            const int mb = rem >> 2, q = rem & 3;
            const int oyl = mb / p.TWB, seg = mb - oyl * p.TWB;
            const int oy = oy0 + oyl, ox = ox0 + seg * 16 + q * 4;
            const int oc = nt * NT + nb * 16 + c;
            if (oy >= p.outH || ox >= p.outW || oc >= p.OC) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + c * CS + mb * 16 + q * 4);
            float* o = outn + (long long)oc * HWo + (long long)oy * p.outW + ox;
            if (vec_ok && ox + 3 < p.outW) {
                *reinterpret_cast<f32x4*>(o) = v;
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ox + r < p.outW) { o[r] = v[r]; amax = fmaxf(amax, fabsf(v[r])); }
            }
        }
    }
    if (p.absmax) absmax_commit(p.absmax + n, amax);   // launch-uniform; every thread of the workgroup gets here
}

template <int NB, int MW>
__device__ __forceinline__ void store_tile(const ConvParams& p, const f32x4 (&acc)[MW][NB], int n, int nt,
                                           int oy0, int ox0, int wave, int li, int lk) {
    const StoreDst d{p.out, p.out_bstride, p.outH, p.outW, p.OC, p.TWB, p.act, p.slope, p.out_absmax};
    store_tile_dst<NB, MW>(d, acc, n, nt, oy0, ox0, wave, li, lk);
}

// One 16-byte-per-lane LDS-DMA: lane l copies gsrc[0..3] to LDS byte address lds_base + 16*l.
// Inline asm on purpose: hipcc's waitcnt pass makes every later ds_read wait vmcnt(0) for a
// builtin LDS-DMA it cannot disambiguate, which would serialise staging and MFMAs; issued from
// asm the DMA is invisible to it and completion is awaited by the kernels' own
// `s_waitcnt vmcnt(0)` in front of the stage barrier.  M0 (the DMA's LDS base) is compiler
// reserved, so it is saved and restored inside the statement (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_base_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_base_uniform)
        : "memory");
}

// Same with a wave-uniform 64-bit base in SGPRs and a per-lane 32-bit byte offset (the
// instruction's saddr + vaddr form): no 64-bit VALU address arithmetic per DMA.
__device__ __forceinline__ void lds_dma16_s(const float* sbase_uniform, unsigned voff_bytes, unsigned lds_base_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff_bytes), "s"(sbase_uniform), "s"(lds_base_uniform)
        : "memory");
}

// Same, executed under an explicit lane mask (wave-uniform 64-bit value): EXEC is swapped inside the
// statement, so masked-off lanes need no branch around the DMA and the caller's code stays one
// basic block.  A zero mask makes the instruction a no-op.
__device__ __forceinline__ void lds_dma16_sm(const float* sbase_uniform, unsigned voff_bytes, unsigned lds_base_uniform,
                                             unsigned long long lane_mask_uniform) {
    unsigned keep;
    unsigned long long keepx;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %5\n\ts_mov_b64 exec, %4\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(keepx)
        : "v"(voff_bytes), "s"(sbase_uniform), "s"(lane_mask_uniform), "s"(lds_base_uniform)
        : "memory");
}

// 4-byte granules (lane l copies gsrc[voff] to LDS lds_base + 4*l): any alignment, e.g. maps whose
// width is not a multiple of 4.
__device__ __forceinline__ void lds_dma4_sm(const float* sbase_uniform, unsigned voff_bytes, unsigned lds_base_uniform,
                                            unsigned long long lane_mask_uniform) {
    unsigned keep;
    unsigned long long keepx;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %5\n\ts_mov_b64 exec, %4\n\ts_nop 0\n\t"
        "global_load_lds_dword %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(keepx)
        : "v"(voff_bytes), "s"(sbase_uniform), "s"(lane_mask_uniform), "s"(lds_base_uniform)
        : "memory");
}

// Pins a wave-uniform pointer into SGPRs (for the "s" operands above when the compiler's divergence
// analysis cannot prove uniformity).
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

// fp16 elements of one (k-group, term) plane of a PAIR tensor (include/kbnet_hip.h): H * W pixels + the zero granule
__host__ __device__ constexpr long long pair_plane_halves(int h, int w) { return ((long long)h * w + 1) * 8; }

__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

struct TileChoice { int MW, TWB; };

// tune.hip: first-use tuning of launch geometry.  `key` = {kernel family, problem shape...};
// candidates are 0..ncand-1, `launch(c)` launches the real problem with candidate c (KBN_OK, or an error
// for an invalid candidate), `model` is the analytic choice (used while capturing / when disabled).
typedef std::array<int, 10> TuneKey;
bool tune_enabled();
bool tune_lookup(const TuneKey& key, int* cand, int ncand);   // false when nothing is cached or the cached index is not in 0..ncand-1
int tune_pick(const TuneKey& key, int ncand, int model, const std::function<int(int)>& launch, hipStream_t stream);

// conv_wino.hip: Winograd F(2x2,3x3) path for wide 3x3 stride-1 convs.  Eligibility by shape only
// (pack time and launch time must agree): the transformed weights live behind the direct-conv
// weights in the packed blob.
struct WinoPlan { int ok, nTilesN; };
__host__ __device__ inline WinoPlan wino_plan(int oc, int cin, int ks, int stride) {
    WinoPlan wp;
    wp.ok = (ks == 3 && stride == 1 && cin >= 32 && (cin % 16) == 0 && oc >= 32) ? 1 : 0;  // even number of 8-channel chunks
    wp.nTilesN = ceil_div(oc, 64);
    return wp;
}
long long wino_packed_floats(int oc, int cin, int ks, int stride);
int wino_pack(const float* weight, float* packed, int oc, int cin, hipStream_t stream);
// Launches the Winograd kernel for a prepared ConvParams (sources, weights, output, act) or
// returns KBN_ERR_UNSUPPORTED (computed / unaligned sources, W % 4 != 0, ...).
int conv_wino_launch(const ConvParams& p, hipStream_t stream);
// workgroups (0 = not eligible) and region shape the launch would use
int wino_query(int n, int oc, int cin, int H, int W, int* RT, int* CT);

// conv_dma.hip: LDS-DMA staged kernel for 16-byte aligned, non-resized tensor sources.
// Fills the staging geometry of `p` itself.  Returns KBN_ERR_UNSUPPORTED if not eligible.
int conv_dma_launch(ConvParams& p, const ConvPlan& pl, TileChoice tc, int kernel_size, int stride,
                    hipStream_t stream);

// kb_pair.hip: conv_image (3x3 s2) and conv_fused (1x1 s2) of a KB block in ONE launch -- they read the same
// image tile -- with conv_depth (3x3 s2 on cat[depth, coordinates]) riding along when its filters fit
// (*depth_done).  Returns KBN_ERR_UNSUPPORTED when the shapes do not qualify (the caller then launches the convs).
struct KbPairArgs {
    const float *image, *fused, *depth, *coords, *kinv, *proj, *wp_image, *wp_fused, *wp_depth;
    float *out_image, *out_fused, *out_depth;
    unsigned *absmax_image, *absmax_fused, *absmax_depth;   // per-frame max |out| slots of the three outputs, or null
    long long image_bstride, fused_bstride, depth_bstride, coords_bstride, out_image_bstride, out_fused_bstride,
        out_depth_bstride;
    int n, height, width, channels_image, channels_depth, channels_fused, filters, filters_depth;
    float slope;
};
int kb_pair_launch(const KbPairArgs& a, hipStream_t stream, bool* depth_done);

}  // namespace kbn
