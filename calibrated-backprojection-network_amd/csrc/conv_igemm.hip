// conv_igemm.hip -- fp32 implicit-GEMM convolution on the gfx950 matrix cores.
//
// Replaces net_utils.Conv2d.forward (reference src/net_utils.py:120-141) together with
// the torch.cat / interpolate(nearest) / coordinate-channel ops the reference runs in
// front of it (src/net_utils.py:497, 1351-1368, 1483).  NCHW fp32 in, NCHW fp32 out.
//
// GEMM view:  M = output pixels, N = output channels, K = (input channel, tap).
//   D[pixel][oc] += A[pixel][k] * B[k][oc]       v_mfma_f32_16x16x4_f32 (exact fp32)
// A workgroup (256 threads = 4 waves) owns a TH x TW pixel tile (TW a multiple of 16,
// TH*TW/16 = 4*MW "m-blocks" of 16 consecutive pixels of one row) times NT = 16*NB output
// channels.  Each wave owns MW m-blocks x all NB n-blocks (MW*NB accumulators of 4 VGPRs).
//
// K loop: channels are consumed CK at a time.  Per chunk the workgroup stages
//   As[CK][plane]   the input tile incl. halo for CK channels (zero padded), and
//   Bs[CK*TAPS*NT]  the pre-packed weight slice (a straight 16-byte copy),
// then runs TAPS * CK/4 MFMA k-steps out of LDS.  All 9 taps re-use one staged tile.
//
// LDS layout (bank-conflict free for ds_read_b32, which is serviced per 32-lane half):
//   A: lane (i = l&15, k = l>>4) reads As[(c4*4+k)*plane + row*pitch + col + i]; plane = 16
//      (mod 32) puts the k=0 / k=1 rows of a half-wave on disjoint 16-bank groups.
//      Stride-2 3x3 convs de-interleave the staged columns by parity so that the 16 lanes
//      of a fragment still read consecutive words.
//   B: fragment order [k>>1][n][k&1] -> the 32 lanes of a half-wave read 32 consecutive words.
#include <stdlib.h>

#include "conv_common.h"

namespace kbn {

// ---------------------------------------------------------------- weight packing
// packed[nt][chunk][tap][c4][k>>1][n][k&1]  (zero padded in both c and oc)
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ packed, int OC,
                                   int Cin, int taps, ConvPlan pl, long long total) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int NT = pl.NT, CK = pl.CK;
    long long per_nt = (long long)pl.Cpad * taps * NT;
    int nt = (int)(e / per_nt);
    int rem = (int)(e - (long long)nt * per_nt);
    int per_chunk = CK * taps * NT;
    int chunk = rem / per_chunk;
    rem -= chunk * per_chunk;
    int per_tap = CK * NT;  // (CK/4) * 4 * NT
    int tap = rem / per_tap;
    rem -= tap * per_tap;
    int c4 = rem / (4 * NT);
    rem -= c4 * 4 * NT;
    int khalf = rem / (2 * NT);
    rem -= khalf * 2 * NT;
    int nn = rem >> 1, klow = rem & 1;
    int c = chunk * CK + c4 * 4 + khalf * 2 + klow;
    int oc = nt * NT + nn;
    float v = 0.f;
    if (c < Cin && oc < OC) v = w[((long long)oc * Cin + c) * taps + tap];
    packed[e] = v;
}

// ------------------------------------------------------------------- the kernel
template <int KS, int STRIDE, int CK, int NB, int MW, int MAXPOS, bool PIPE>
__global__ __launch_bounds__(256, KBN_WAVES_PER_SIMD) void conv_igemm_kernel(const ConvParams p) {
    constexpr int TAPS = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int STEP = (KS == 1) ? STRIDE : 1;  // spacing of staged positions in the input
    constexpr bool S2 = (KS == 3 && STRIDE == 2);
    constexpr int NT = NB * 16;
    constexpr int NC4 = CK / 4;
    constexpr int B_FLOATS = CK * TAPS * NT;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int a_floats = CK * p.plane;
    const int buf_floats = a_floats + B_FLOATS;  // one stage: [As | Bs]; PIPE uses two stages

    const int tid = threadIdx.x;
    int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int nt = bid % p.nTilesN;
    bid /= p.nTilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int TW = p.TWB * 16;
    const int oy0 = ty * p.TH, ox0 = tx * TW;

    // ---- staging table: which input element / LDS word each thread moves -------------
    int goff[MAXPOS], loff[MAXPOS];
    {
        const int Y0 = oy0 * STRIDE - PAD, X0 = ox0 * STRIDE - PAD;
        const int npos = p.rowsS * p.colsS;
        const int srcH = p.src[0].H, srcW = p.src[0].W;
#pragma unroll
        for (int u = 0; u < MAXPOS; ++u) {
            int pos = tid + u * 256;
            int g = -1, l = -1;
            if (pos < npos) {
                int r = pos / p.colsS;
                int ci = pos - r * p.colsS;
                int Y = Y0 + r * STEP, X = X0 + ci * STEP;
                l = S2 ? (r * p.pitch + (ci & 1) * p.PH + (ci >> 1)) : (r * p.pitch + ci);
                if (Y >= 0 && Y < p.inH && X >= 0 && X < p.inW) {
                    if (p.resize) {
                        g = nearest_src_index(Y, srcH, p.inH) * srcW + nearest_src_index(X, srcW, p.inW);
                    } else {
                        g = Y * p.inW + X;
                    }
                }
            }
            goff[u] = g;
            loff[u] = l;
        }
    }
    // synthesized source (KB layer), if any: coordinates K^-1 [x y 1]^T, optionally times
    // z = act(proj . depth[:, y, x]) -- reference src/net_utils.py:1351-1360
    int syn = -1;
#pragma unroll
    for (int s = 0; s < KBN_MAX_SRC; ++s)
        if (s < p.nsrc && p.src[s].kind != KBN_SRC_TENSOR) syn = s;

    // ---- per-lane fragment addressing ----------------------------------------------
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    int mbase[MW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        int mb = wave * MW + mi;
        int oy = mb / p.TWB;
        int seg = mb - oy * p.TWB;
        mbase[mi] = (S2 ? 2 * oy : oy) * p.pitch + seg * 16 + li + lk * p.plane;
    }
    const int boff = (lk >> 1) * 2 * NT + li * 2 + (lk & 1);

    f32x4 acc[MW][NB];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* wp_nt = p.wp + (long long)nt * p.Cpad * TAPS * NT;

    // Values of the CK chunk channels at every staged position of this thread (zero outside the
    // image and in the channel padding).  Fast path: the chunk lies inside one tensor source
    // (or runs off the end of the last one into padding) -> MAXPOS*CK unconditional, independent
    // loads that the compiler can issue back to back and wait for late.  Generic path: the chunk
    // straddles sources or holds synthesized KB channels.
    auto load_chunk = [&](int c0, float (&va)[MAXPOS][CK]) -> int {
        int s = 0;
#pragma unroll
        for (int t = 1; t < KBN_MAX_SRC; ++t)
            if (t < p.nsrc && c0 >= p.src[t].cstart) s = t;
        const int cend = p.src[s].cstart + p.src[s].C;
        const bool fast = (c0 < cend) && p.src[s].kind == KBN_SRC_TENSOR && (c0 + CK <= cend || cend == p.Ctot);
        if (fast) {
            const int HW = p.src[s].H * p.src[s].W;
            const float* base = p.src[s].data + (long long)n * p.src[s].bstride +
                                (long long)(c0 - p.src[s].cstart) * HW;
            const int nvalid = (cend - c0 < CK) ? (cend - c0) : CK;
#pragma unroll
            for (int u = 0; u < MAXPOS; ++u) {
                const int g = goff[u];
                const int gi = g < 0 ? 0 : g;
#pragma unroll
                for (int q = 0; q < CK; ++q) {
                    const int qq = (q < nvalid) ? q : 0;
                    va[u][q] = base[qq * HW + gi];  // raw: masked in store_pos, so that nothing
                }                                   // consumes the load before the MFMAs have run
            }
            return nvalid;
        }
#pragma unroll
        for (int u = 0; u < MAXPOS; ++u) {
            const int g = goff[u];
            float cv[3] = {0.f, 0.f, 0.f};
            float z = 1.f;
            if (syn >= 0 && c0 + CK > p.src[syn].cstart && c0 < p.src[syn].cstart + 3 && g >= 0) {
                const SrcDev& sd = p.src[syn];
                const int HW = p.inH * p.inW;
                if (sd.kind == KBN_SRC_XYZ && sd.coords) {
                    const float* cb = sd.coords + (long long)n * sd.coords_bstride + g;
                    cv[0] = cb[0]; cv[1] = cb[HW]; cv[2] = cb[2 * HW];
                } else {
                    const float* kinv = sd.kinv + (long long)n * 9;
                    const int Y = g / p.inW, X = g - Y * p.inW;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        cv[j] = fmaf(kinv[j * 3 + 1], (float)Y, kinv[j * 3 + 0] * (float)X) + kinv[j * 3 + 2];
                }
                if (sd.kind == KBN_SRC_XYZ) {
                    const float* db = sd.data + (long long)n * sd.bstride + g;
                    float a = 0.f;
                    for (int c = 0; c < sd.Cd; ++c) a = fmaf(sd.proj[c], db[(long long)c * HW], a);
                    z = p.act ? leaky_relu(a, p.slope) : a;
                }
            }
#pragma unroll
            for (int q = 0; q < CK; ++q) {
                const ChanRef cr = chan_lookup(p, n, c0 + q);
                float val = 0.f;
                if (g >= 0) {
                    if (cr.kind == KBN_SRC_TENSOR) val = cr.ptr[g];
                    else if (cr.kind >= 0) val = (cr.j == 0 ? cv[0] : (cr.j == 1 ? cv[1] : cv[2])) * z;
                }
                va[u][q] = val;
            }
        }
        return CK;
    };
    auto store_pos = [&](float* As, int u, const float (&v)[CK], int nvalid) {
        if (loff[u] >= 0) {
            const bool inb = goff[u] >= 0;
#pragma unroll
            for (int q = 0; q < CK; ++q) As[q * p.plane + loff[u]] = (inb && q < nvalid) ? v[q] : 0.f;
        }
    };
    // packed weight slice of chunk c0 -> LDS (a straight copy; PIPE: LDS-DMA, no VGPR round trip)
    auto stage_B = [&](float* Bs, int c0) {
        constexpr int CNT4 = B_FLOATS / 4;
        const float4* s4 = reinterpret_cast<const float4*>(wp_nt + (long long)c0 * TAPS * NT);
        if constexpr (PIPE) {
            const unsigned bs = __builtin_amdgcn_readfirstlane(lds_addr(Bs));
#pragma unroll
            for (int e0 = 0; e0 < CNT4; e0 += 256) {
                const int eb = e0 + wave * 64;  // wave-uniform
                if (eb + lane < CNT4)
                    lds_dma16_s(reinterpret_cast<const float*>(s4 + eb), (unsigned)(lane * 16), bs + eb * 16);
            }
        } else {
            float4* d4 = reinterpret_cast<float4*>(Bs);
            for (int e = tid; e < CNT4; e += 256) d4[e] = s4[e];
        }
    };
    auto compute = [&](const float* As, const float* Bs) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            const int toff = (KS == 1) ? 0
                             : (S2 ? (ky * p.pitch + (kx == 1 ? p.PH : (kx == 2 ? 1 : 0)))
                                   : (ky * p.pitch + kx));
#pragma unroll
            for (int c4 = 0; c4 < NC4; ++c4) {
                const float* Ab = As + c4 * 4 * p.plane + toff;
                const float* Bb = Bs + (tap * NC4 + c4) * 4 * NT + boff;
                float a[MW], b[NB];
#pragma unroll
                for (int mi = 0; mi < MW; ++mi) a[mi] = Ab[mbase[mi]];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b[nb] = Bb[nb * 32];
#pragma unroll
                for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[nb], acc[mi][nb], 0, 0, 0);
            }
        }
    };

    if constexpr (PIPE) {
        // Double-buffered: the next chunk's global loads (A -> VGPRs, B -> LDS by DMA) are in
        // flight while the MFMAs of the current chunk run; one barrier per chunk.
        float va[MAXPOS][CK];
        stage_B(smem + a_floats, 0);
        int nv = load_chunk(0, va);
#pragma unroll
        for (int u = 0; u < MAXPOS; ++u) store_pos(smem, u, va[u], nv);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int cur = 0;
        for (int c0 = 0; c0 < p.Cpad; c0 += CK) {
            float* curA = smem + cur * buf_floats;
            float* nxtA = smem + (cur ^ 1) * buf_floats;
            const bool more = (c0 + CK < p.Cpad);
            if (more) {
                if (!(p.dbg & 1)) nv = load_chunk(c0 + CK, va);  // A loads first: the DMA below has no register
                if (!(p.dbg & 2)) stage_B(nxtA + a_floats, c0 + CK);  // result: nothing waits on it before the barrier
            }
            if (!(p.dbg & 4)) compute(curA, curA + a_floats);
            if (more && !(p.dbg & 1)) {
#pragma unroll
                for (int u = 0; u < MAXPOS; ++u) store_pos(nxtA, u, va[u], nv);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
    } else {
        for (int c0 = 0; c0 < p.Cpad; c0 += CK) {
            __syncthreads();  // previous chunk's fragments are consumed
            stage_B(smem + a_floats, c0);
            {
                float va[MAXPOS][CK];
                const int nv = load_chunk(c0, va);
#pragma unroll
                for (int u = 0; u < MAXPOS; ++u) store_pos(smem, u, va[u], nv);
            }
            __syncthreads();
            compute(smem, smem + a_floats);
        }
    }

    store_tile<NB, MW>(p, acc, n, nt, oy0, ox0, wave, li, lk);
}

// ----------------------------------------------------------------- host dispatch
template <int KS, int STRIDE, int CK, int NB, int MW>
static int launch_variant(const ConvParams& p, size_t stage_bytes, hipStream_t stream) {
    constexpr int MAXPOS = conv_maxpos(KS, STRIDE, MW);
    constexpr bool PIPE = (MAXPOS * CK <= 48);
    auto kern = conv_igemm_kernel<KS, STRIDE, CK, NB, MW, MAXPOS, PIPE>;
    static DeviceOnce once;
    if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
    if (p.rowsS * p.colsS > MAXPOS * 256) return KBN_ERR_UNSUPPORTED;
    size_t lds_bytes = stage_bytes * (PIPE ? 2 : 1);
    if (lds_bytes > 160 * 1024) return KBN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), lds_bytes, stream, p);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

template <int KS, int STRIDE, int CK, int NB>
static int launch_mw(const ConvParams& p, int MW, size_t lds, hipStream_t st) {
    if constexpr (NB >= 3) {
        switch (MW) {
            case 1: return launch_variant<KS, STRIDE, CK, NB, 1>(p, lds, st);
            case 2: return launch_variant<KS, STRIDE, CK, NB, 2>(p, lds, st);
            default: return launch_variant<KS, STRIDE, CK, NB, 4>(p, lds, st);
        }
    } else {
        switch (MW) {
            case 1: return launch_variant<KS, STRIDE, CK, NB, 1>(p, lds, st);
            case 2: return launch_variant<KS, STRIDE, CK, NB, 2>(p, lds, st);
            case 4: return launch_variant<KS, STRIDE, CK, NB, 4>(p, lds, st);
            default: return launch_variant<KS, STRIDE, CK, NB, 8>(p, lds, st);
        }
    }
}

template <int KS, int STRIDE, int CK>
static int launch_nb(const ConvParams& p, int NB, int MW, size_t lds, hipStream_t st) {
    switch (NB) {
        case 1: return launch_mw<KS, STRIDE, CK, 1>(p, MW, lds, st);
        case 2: return launch_mw<KS, STRIDE, CK, 2>(p, MW, lds, st);
        case 3: return launch_mw<KS, STRIDE, CK, 3>(p, MW, lds, st);
        default: return launch_mw<KS, STRIDE, CK, 4>(p, MW, lds, st);
    }
}

// Tile choice: 4*MW m-blocks of 16 pixels arranged as TH rows x TWB segments.  Cost model:
// the chip runs ~256 workgroups at a time (one per CU, MFMA bound), each costing MW units
// plus a fixed staging overhead; small feature maps pick small tiles so all CUs get work.
static TileChoice choose_tile(int outH, int outW, int n, int nTilesN, int kernel_size, int stride, int max_mw,
                              int CK, int NT) {
    // experiment hooks (tools/conv_bench.py): KBN_FORCE_MW / KBN_FORCE_TWB pin the tile
    const int force_mw = knob(KNOB_FORCE_MW), force_twb = knob(KNOB_FORCE_TWB);
    const bool s2 = (kernel_size == 3 && stride == 2);
    double best_cost = 1e300;
    TileChoice best{max_mw, 2};
    for (int mw = max_mw; mw >= 1; mw /= 2) {
        if (force_mw && mw != force_mw && !(force_mw > max_mw && mw == max_mw)) continue;
        const int mblocks = 4 * mw;
        for (int twb = 1; twb <= 4 && twb <= mblocks; twb *= 2) {
            if (force_twb && twb != force_twb) continue;
            int th = mblocks / twb, tw = twb * 16;
            long long tiles = (long long)ceil_div(outW, tw) * ceil_div(outH, th) * n * nTilesN;
            int rows = (kernel_size == 1) ? th : (s2 ? 2 * th + 1 : th + 2);
            int cols = (kernel_size == 1) ? stride * tw : (s2 ? 2 * tw + 4 : tw + 8);
            // double-buffered stage size -> resident workgroups per CU (a lone workgroup cannot
            // overlap its barrier / staging bubbles with another one's MFMAs)
            double lds = 2.0 * 4.0 * (CK * (rows * cols + 32.0) + CK * kernel_size * kernel_size * NT);
            int resident = (int)(160.0 * 1024.0 / lds);
            double occ_penalty = resident >= 2 ? 1.0 : 1.18;
            double rounds = (double)((tiles + 255) / 256);
            double cost = rounds * (mw * 64.0 + 12.0 + 0.02 * rows * cols) * occ_penalty;
            if (cost < best_cost * 0.98 || (cost < best_cost && mw == best.MW)) {
                best_cost = cost;
                best = TileChoice{mw, twb};
            }
        }
    }
    return best;
}

// candidate index <-> tile shape for the tuner (tune.hip): MW in {1,2,4,8} x TWB in {1,2,4}
static TileChoice tile_of_cand(int c) { return TileChoice{1 << (c / 3), 1 << (c % 3)}; }
static int cand_of_tile(TileChoice t) {
    int a = 0, b2 = 0;
    while ((1 << a) < t.MW) ++a;
    while ((1 << b2) < t.TWB) ++b2;
    return a * 3 + b2;
}

int conv2d_launch(const kbn_conv_src* srcs, int n_src, const float* packed_weight, float* out,
                  long long out_batch_stride, int n, int out_channels, int kernel_size, int stride,
                  int in_height, int in_width, int resize, int apply_activation, float negative_slope,
                  unsigned* out_absmax, hipStream_t stream) {
    if (!srcs || !packed_weight || !out) return KBN_ERR_INVALID_ARGUMENT;
    if (n_src < 1 || n_src > KBN_MAX_SRC || n < 1 || out_channels < 1 || in_height < 1 || in_width < 1)
        return KBN_ERR_INVALID_ARGUMENT;
    if ((kernel_size != 1 && kernel_size != 3) || (stride != 1 && stride != 2)) return KBN_ERR_UNSUPPORTED;
    if (resize != KBN_RESIZE_NONE && resize != KBN_RESIZE_NEAREST) return KBN_ERR_INVALID_ARGUMENT;
    if (resize == KBN_RESIZE_NEAREST && (n_src != 1 || srcs[0].kind != KBN_SRC_TENSOR)) return KBN_ERR_UNSUPPORTED;
    if (in_height > 32767 || in_width > 32767) return KBN_ERR_UNSUPPORTED;

    ConvParams p;
    int ctot = 0;
    for (int s = 0; s < n_src; ++s) {
        const kbn_conv_src& a = srcs[s];
        SrcDev& d = p.src[s];
        if (a.channels < 1) return KBN_ERR_INVALID_ARGUMENT;
        d.kind = a.kind; d.C = a.channels; d.cstart = ctot;
        d.data = a.data; d.bstride = a.batch_stride;
        d.H = in_height; d.W = in_width; d.Cd = 0;
        d.proj = nullptr; d.coords = nullptr; d.kinv = nullptr; d.coords_bstride = 0;
        if (a.kind == KBN_SRC_TENSOR) {
            if (!a.data) return KBN_ERR_INVALID_ARGUMENT;
            if (resize == KBN_RESIZE_NEAREST) {
                if (a.src_height < 1 || a.src_width < 1) return KBN_ERR_INVALID_ARGUMENT;
                d.H = a.src_height; d.W = a.src_width;
            } else if ((a.src_height && a.src_height != in_height) || (a.src_width && a.src_width != in_width)) {
                return KBN_ERR_INVALID_ARGUMENT;
            }
        } else if (a.kind == KBN_SRC_COORDS) {
            if (!a.kinv || a.channels != 3) return KBN_ERR_INVALID_ARGUMENT;
            d.kinv = a.kinv;
        } else if (a.kind == KBN_SRC_XYZ) {
            if (!a.data || !a.proj_weight || a.aux_channels < 1 || a.channels != 3) return KBN_ERR_INVALID_ARGUMENT;
            if (!a.coordinates && !a.kinv) return KBN_ERR_INVALID_ARGUMENT;
            d.proj = a.proj_weight; d.Cd = a.aux_channels;
            d.coords = a.coordinates; d.coords_bstride = a.coordinates_batch_stride; d.kinv = a.kinv;
        } else {
            return KBN_ERR_INVALID_ARGUMENT;
        }
        ctot += a.channels;
    }
    for (int s = n_src; s < KBN_MAX_SRC; ++s) p.src[s] = p.src[0];

    const ConvPlan pl = make_plan(out_channels, ctot, kernel_size, stride, 0);
    p.nsrc = n_src; p.N = n; p.OC = out_channels; p.Ctot = ctot; p.Cpad = pl.Cpad;
    p.wp = packed_weight; p.out = out; p.out_bstride = out_batch_stride;
    p.inH = in_height; p.inW = in_width;
    p.outH = ceil_div(in_height, stride); p.outW = ceil_div(in_width, stride);
    p.resize = resize; p.act = apply_activation ? 1 : 0; p.slope = negative_slope;
    p.nTilesN = pl.nTilesN;
    p.dbg = knob(KNOB_DEBUG);
    p.out_absmax = out_absmax;   // folded in the epilogues of conv_dma_kernel / conv_igemm_kernel
    {   // LDS-transposed epilogue for store-bound launches: few multiply-adds per output (conv0, deconv0's conv).
        // KBN_EPI_LDS = 0 never / 2 always (A/B, tests); default: K = channels x taps <= 128
        const int mode = knob_set(KNOB_EPI_LDS) ? knob(KNOB_EPI_LDS) : 1;
        p.epi_lds = mode == 2 || (mode == 1 && ctot * kernel_size * kernel_size <= 128);
    }

    const bool s2 = (kernel_size == 3 && stride == 2);
    TileChoice tc = choose_tile(p.outH, p.outW, n, pl.nTilesN, kernel_size, stride, pl.MW, pl.CK, pl.NT);
    if (kernel_size == 3 && stride == 1 && !knob(KNOB_NO_WINO)) {  // wide 3x3: Winograd F(2x2,3x3)
        int rc = conv_wino_launch(p, stream);
        if (rc == KBN_OK && out_absmax && !knob(KNOB_NO_SPLIT))   // (KBN_NO_SPLIT: nothing reads slots) the Winograd kernel (a fallback since the split-operand convs) has no slot epilogue
            rc = absmax_frames_launch(out, out_batch_stride, n, (long long)out_channels * p.outH * p.outW, out_absmax, stream);
        if (rc != KBN_ERR_UNSUPPORTED) return rc;
    }
    {  // fast path: LDS-DMA staging (aligned tensor sources, no resize)
        const bool forced = knob(KNOB_FORCE_MW) || knob(KNOB_FORCE_TWB);
        int sig = n_src;
        for (int s = 0; s < n_src; ++s) sig = sig * 4 + srcs[s].kind;
        const TuneKey key{1, n, out_channels, ctot, kernel_size, stride, in_height, in_width, sig, 0};
        int cand = cand_of_tile(tc);
        if (!forced && tune_lookup(key, &cand, 12)) tc = tile_of_cand(cand);    // tuned earlier / preloaded cache
        const bool tuning = !forced && tune_enabled();                      // opt-in (kbn_set_autotune)
        ConvParams q = p;
        int rc = conv_dma_launch(q, pl, tc, kernel_size, stride, stream);   // also the eligibility check
        if (rc == KBN_OK && tuning && !tune_lookup(key, &cand, 12)) {           // first eligible launch of this shape
            cand = tune_pick(key, 12, cand_of_tile(tc), [&](int c) {
                const TileChoice t = tile_of_cand(c);
                if (t.MW > pl.MW || t.TWB > 4 * t.MW) return (int)KBN_ERR_UNSUPPORTED;
                ConvParams r = p;
                return conv_dma_launch(r, pl, t, kernel_size, stride, stream);
            }, stream);
        }
        if (rc != KBN_ERR_UNSUPPORTED) return rc;
    }
    const int mblocks = 4 * tc.MW;
    p.TWB = tc.TWB; p.TH = mblocks / tc.TWB;
    const int TW = tc.TWB * 16;
    p.tilesX = ceil_div(p.outW, TW); p.tilesY = ceil_div(p.outH, p.TH);
    if (kernel_size == 1) { p.rowsS = p.TH; p.colsS = TW; p.PH = 0; p.pitch = TW; }
    else if (s2) { p.rowsS = 2 * p.TH + 1; p.colsS = 2 * TW + 1; p.PH = TW + 1; p.pitch = 2 * p.PH; }
    else { p.rowsS = p.TH + 2; p.colsS = TW + 2; p.PH = 0; p.pitch = TW + 2; }
    int plane = p.rowsS * p.pitch;
    plane = ((plane + 15) / 32) * 32 + 16;  // smallest value >= plane that is 16 (mod 32)
    p.plane = plane;
    long long nb64 = (long long)p.tilesX * p.tilesY * n * p.nTilesN;
    if (nb64 > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.nblocks = (int)nb64;
    const int taps = kernel_size * kernel_size;
    size_t lds = sizeof(float) * ((size_t)pl.CK * plane + (size_t)pl.CK * taps * pl.NT);

    if (kernel_size == 3 && stride == 1)
        return pl.CK == 4 ? launch_nb<3, 1, 4>(p, pl.NB, tc.MW, lds, stream)
                          : launch_nb<3, 1, 8>(p, pl.NB, tc.MW, lds, stream);
    if (kernel_size == 3 && stride == 2)
        return pl.CK == 4 ? launch_nb<3, 2, 4>(p, pl.NB, tc.MW, lds, stream)
                          : launch_nb<3, 2, 8>(p, pl.NB, tc.MW, lds, stream);
    if (stride == 2) return launch_nb<1, 2, 16>(p, pl.NB, tc.MW, lds, stream);
    return launch_nb<1, 1, 16>(p, pl.NB, tc.MW, lds, stream);
}

}  // namespace kbn

extern "C" {

size_t kbn_conv2d_packed_weight_bytes(int out_channels, int in_channels, int kernel_size, int stride) {
    if (out_channels < 1 || in_channels < 1 || (kernel_size != 1 && kernel_size != 3)) return 0;
    if (stride != 1 && stride != 2) return 0;
    kbn::ConvPlan pl = kbn::make_plan(out_channels, in_channels, kernel_size, stride, 0);
    return sizeof(float) * ((size_t)pl.nTilesN * pl.Cpad * kernel_size * kernel_size * pl.NT +
                            (size_t)kbn::wino_packed_floats(out_channels, in_channels, kernel_size, stride));
}

int kbn_conv2d_pack_weight(const float* weight, float* packed, int out_channels, int in_channels,
                           int kernel_size, int stride, kbn_stream_t stream) {
    if (!weight || !packed || out_channels < 1 || in_channels < 1) return KBN_ERR_INVALID_ARGUMENT;
    if ((kernel_size != 1 && kernel_size != 3) || (stride != 1 && stride != 2)) return KBN_ERR_UNSUPPORTED;
    kbn::ConvPlan pl = kbn::make_plan(out_channels, in_channels, kernel_size, stride, 0);
    int taps = kernel_size * kernel_size;
    long long total = (long long)pl.nTilesN * pl.Cpad * taps * pl.NT;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(kbn::pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight,
                       packed, out_channels, in_channels, taps, pl, total);
    KBN_CHECK_LAUNCH();
    if (kbn::wino_plan(out_channels, in_channels, kernel_size, stride).ok)  // + G g G^T behind it
        return kbn::wino_pack(weight, packed + total, out_channels, in_channels, (hipStream_t)stream);
    return KBN_OK;
}

int kbn_conv2d_query(int n, int out_channels, int in_channels, int kernel_size, int stride, int in_height,
                     int in_width, int resize, int* info) {
    using namespace kbn;
    if (!info || n < 1 || out_channels < 1 || in_channels < 1 || in_height < 1 || in_width < 1)
        return KBN_ERR_INVALID_ARGUMENT;
    if ((kernel_size != 1 && kernel_size != 3) || (stride != 1 && stride != 2)) return KBN_ERR_UNSUPPORTED;
    const ConvPlan pl = make_plan(out_channels, in_channels, kernel_size, stride, 0);
    const int outH = ceil_div(in_height, stride), outW = ceil_div(in_width, stride);
    TileChoice tc = choose_tile(outH, outW, n, pl.nTilesN, kernel_size, stride, pl.MW, pl.CK, pl.NT);
    {   // a tuned choice, if this shape has run already (tensor sources only: the common signatures)
        for (int nsrc = 1; nsrc <= KBN_MAX_SRC; ++nsrc) {
            int sig = nsrc, cand = 0;
            for (int s = 0; s < nsrc; ++s) sig = sig * 4 + KBN_SRC_TENSOR;
            if (tune_lookup(TuneKey{1, n, out_channels, in_channels, kernel_size, stride, in_height, in_width, sig, 0}, &cand, 12)) {
                tc = tile_of_cand(cand);
                break;
            }
        }
    }
    const int th = 4 * tc.MW / tc.TWB, tw = tc.TWB * 16;
    const int maxpos = conv_maxpos(kernel_size, stride, tc.MW);
    info[0] = pl.CK; info[1] = pl.NB; info[2] = tc.MW; info[3] = tc.TWB; info[4] = th;
    info[5] = ceil_div(outW, tw) * ceil_div(outH, th) * n * pl.nTilesN;
    info[6] = maxpos;
    // 2: conv_dma_kernel (assuming 16-byte aligned planes), 1: conv_igemm_kernel pipelined, 0: not pipelined
    info[7] = (!resize && (in_width & 3) == 0) ? 2 : ((maxpos * pl.CK <= 48) ? 1 : 0);
    if (kernel_size == 3 && stride == 1 && !resize && !knob(KNOB_NO_WINO)) {
        int rt = 0, ct = 0;
        const int wgs = wino_query(n, out_channels, in_channels, in_height, in_width, &rt, &ct);
        if (wgs > 0) { info[7] = 3; info[2] = rt; info[3] = ct; info[4] = 2 * rt; info[5] = wgs; }
    }
    return KBN_OK;
}

int kbn_conv2d_forward(const kbn_conv_src* srcs, int n_src, const float* packed_weight, float* out,
                       long long out_batch_stride, int n, int out_channels, int kernel_size,
                       int stride, int in_height, int in_width, int resize, int apply_activation,
                       float negative_slope, unsigned* out_absmax, kbn_stream_t stream) {
    return kbn::conv2d_launch(srcs, n_src, packed_weight, out, out_batch_stride, n, out_channels,
                              kernel_size, stride, in_height, in_width, resize, apply_activation,
                              negative_slope, out_absmax, (hipStream_t)stream);
}

}  // extern "C"
