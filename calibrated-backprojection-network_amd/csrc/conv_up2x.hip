// conv_up2x.hip -- UpConv2d for an exact 2x nearest upsample: interpolate(nearest, 2x) followed
// by a 3x3 conv (reference net_utils.UpConv2d.forward, src/net_utils.py:484-499) evaluated as
// four 2x2 convs on the LOW-resolution input, one per output phase (a, b) = (Y & 1, X & 1):
//
//   out[2y+a, 2x+b] = sum_{dy,dx in {0,1}} W'_{ab}[dy][dx] . in[y-1+a+dy, x-1+b+dx]
//   W'_{ab}[dy][dx] = sum_{ky in S_a(dy)} sum_{kx in S_b(dx)} W[ky][kx]
//   S_0(0) = {0}, S_0(1) = {1,2}, S_1(0) = {0,1}, S_1(1) = {2}
//
// (two of the three taps of the 3x3 window always land on the same low-res pixel).  That is
// 4 instead of 9 MACs per output and channel -- 2.25x fewer MFMAs for the five up-convs of the
// decoder (37.5 of KBNet's 100.7 GFLOP) -- and the upsampled tensor is never formed.  The
// pre-summed weights change rounding at the 1e-7 level only.  Zero padding carries over
// exactly: an out-of-image upsampled row/column corresponds to an out-of-image low-res one.
//
// Same MFMA machinery as conv_igemm.hip (v_mfma_f32_16x16x4_f32, LDS-staged tile + packed
// weights, double buffered: A by register prefetch, B by LDS-DMA).  A workgroup owns a
// TH x TW low-res tile, ONE row phase `a` and BOTH column phases: each lane ends up with 4
// consecutive low-res x for b = 0 and b = 1, i.e. 8 consecutive output pixels -> two 16-byte
// stores.  Weights are packed per row phase as [a][n-tile][chunk][dy][c/4][dx][b][k>>1][n][k&1].
#include <stdlib.h>

#include "conv_common.h"

namespace kbn {

struct Up2xPlan {
    int CK, NB, MW, NT, nTilesN, Cpad;
};

__host__ __device__ inline Up2xPlan make_up2x_plan(int oc, int cin) {
    Up2xPlan pl;
    pl.CK = 8;
    int nblk = ceil_div(oc, 16);
    int best = 1, bestpad = 1 << 30;
    for (int nb = 1; nb <= 4; ++nb) {
        int pad = ceil_div(nblk, nb) * nb;
        if (pad < bestpad || (pad == bestpad && nb > best)) { best = nb; bestpad = pad; }
    }
    pl.NB = best;
    pl.MW = (best >= 3) ? 2 : 4;  // 2 * MW * NB accumulators (both column phases)
    pl.NT = best * 16;
    pl.nTilesN = ceil_div(nblk, best);
    pl.Cpad = round_up(cin, pl.CK);
    return pl;
}

// tr = 1: the four-phase weights of ConvTranspose2d(kernel 3, stride 2, padding 1, output_padding 1) (reference src/net_utils.py:383-390)
// instead of the nearest-2x fold: out[2i - 1 + ky] += in[i] w[ky] puts tap ky = 1 of row Y (dy 1) on an even output row (a = 0) and taps
// ky = 2 of row Y (dy 0), ky = 0 of row Y + 1 (dy 1) on an odd one; columns alike.  `w`: out_channels x in_channels x 3 x 3 either way.
__global__ void pack_up2x_kernel(const float* __restrict__ w, float* __restrict__ packed, int OC, int Cin,
                                 Up2xPlan pl, long long total, int tr) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int NT = pl.NT, CK = pl.CK, NC4 = CK / 4;
    const long long per_nt = (long long)pl.Cpad * 8 * NT;  // 2 dy * 2 dx * 2 b taps per channel
    const long long per_a = per_nt * pl.nTilesN;
    const int a = (int)(e / per_a);
    long long rem = e - a * per_a;
    const int nt = (int)(rem / per_nt);
    int r = (int)(rem - nt * per_nt);
    const int per_chunk = CK * 8 * NT;
    const int chunk = r / per_chunk; r -= chunk * per_chunk;
    const int per_dy = NC4 * 4 * 4 * NT;
    const int dy = r / per_dy; r -= dy * per_dy;
    const int per_c4 = 4 * 4 * NT;
    const int c4 = r / per_c4; r -= c4 * per_c4;
    const int dx = r / (2 * 4 * NT); r -= dx * 2 * 4 * NT;
    const int b = r / (4 * NT); r -= b * 4 * NT;
    const int khalf = r / (2 * NT); r -= khalf * 2 * NT;
    const int nn = r >> 1, klow = r & 1;
    const int c = chunk * CK + c4 * 4 + khalf * 2 + klow;
    const int oc = nt * NT + nn;
    float v = 0.f;
    if (c < Cin && oc < OC) {
        const float* wk = w + ((long long)oc * Cin + c) * 9;
        int ky0 = (a == 0) ? (dy == 0 ? 0 : 1) : (dy == 0 ? 0 : 2);
        int ky1 = (a == 0) ? (dy == 0 ? 0 : 2) : (dy == 0 ? 1 : 2);
        int kx0 = (b == 0) ? (dx == 0 ? 0 : 1) : (dx == 0 ? 0 : 2);
        int kx1 = (b == 0) ? (dx == 0 ? 0 : 2) : (dx == 0 ? 1 : 2);
        if (tr) {   // one tap or none: (phase 0, offset 0) reads nothing
            ky0 = a == 0 ? 1 : (dy == 0 ? 2 : 0); ky1 = (a == 0 && dy == 0) ? 0 : ky0;
            kx0 = b == 0 ? 1 : (dx == 0 ? 2 : 0); kx1 = (b == 0 && dx == 0) ? 0 : kx0;
        }
        for (int ky = ky0; ky <= ky1; ++ky)
            for (int kx = kx0; kx <= kx1; ++kx) v += wk[ky * 3 + kx];
    }
    packed[e] = v;
}

// 3-product form of the two column phases (used by conv_up2x_dma_kernel<..., T3 = true>).  With the row-summed
// column filter (g0, g1, g2) of a row phase / row tap, the two output columns of low-res pixel x are
//     o0 = g0 in[x-1] + (g1 + g2) in[x]   =  (-g0) (in[x] - in[x-1]) + G in[x]
//     o1 = (g0 + g1) in[x] + g2 in[x+1]   =    g2  (in[x+1] - in[x]) + G in[x],      G = g0 + g1 + g2:
// three products (L = -g0 on the left difference, C = G on the pixel, R = g2 on the right difference) instead of
// the four of the plain 4-phase form -- 3/4 of the MFMAs, the differences cost two VALU subtractions per fragment.
// Layout per (row phase, n-tile): [chunk of 8 ch][dy][c4][L, C, R][4 x NT fragment block]; sums in fp64.
__global__ void pack_up2x3_kernel(const float* __restrict__ w, float* __restrict__ packed, int OC, int Cin,
                                  Up2xPlan pl, long long total) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int NT = pl.NT;
    const long long per_nt = (long long)pl.Cpad * 6 * NT;
    const long long per_a = per_nt * pl.nTilesN;
    const int a = (int)(e / per_a);
    long long rem = e - a * per_a;
    const int nt = (int)(rem / per_nt);
    int r = (int)(rem - nt * per_nt);
    const int chunk = r / (48 * NT); r -= chunk * 48 * NT;
    const int dy = r / (24 * NT); r -= dy * 24 * NT;
    const int c4 = r / (12 * NT); r -= c4 * 12 * NT;
    const int m = r / (4 * NT); r -= m * 4 * NT;
    const int khalf = r / (2 * NT); r -= khalf * 2 * NT;
    const int nn = r >> 1, klow = r & 1;
    const int c = chunk * 8 + c4 * 4 + khalf * 2 + klow;
    const int oc = nt * NT + nn;
    float v = 0.f;
    if (c < Cin && oc < OC) {
        const float* wk = w + ((long long)oc * Cin + c) * 9;
        const int ky0 = (a == 0) ? (dy == 0 ? 0 : 1) : (dy == 0 ? 0 : 2);
        const int ky1 = (a == 0) ? (dy == 0 ? 0 : 2) : (dy == 0 ? 1 : 2);
        double g[3] = {0.0, 0.0, 0.0};
        for (int ky = ky0; ky <= ky1; ++ky)
            for (int kx = 0; kx < 3; ++kx) g[kx] += (double)wk[ky * 3 + kx];
        v = (float)(m == 0 ? -g[0] : (m == 1 ? g[0] + g[1] + g[2] : g[2]));
    }
    packed[e] = v;
}

struct Up2xParams {
    const float* src;
    const float* wp;
    const float* wp3;   // 3-product weights (behind the 4-phase ones in the packed blob)
    const float* wp9;   // 9-product weights (behind those)
    float* out;
    long long src_bstride, out_bstride;
    int N, Cin, Cpad, OC, srcH, srcW;
    int tilesX, tilesY, nTilesN, nblocks;
    int TWB, TH, rowsS, colsS, pitch, plane;
    int act;
    float slope;
};

template <int I> struct IC2 { static constexpr int value = I; };

// Epilogue shared by both kernels: interleave the two column phases -> 8 consecutive output pixels per
// lane (two 16-byte stores), fused LeakyReLU.
template <int NB, int MW>
__device__ __forceinline__ void up2x_store(const Up2xParams& p, const f32x4 (&acc)[2][MW][NB], int n, int nt, int a,
                                           int y0, int x0, int twb, int wave, int li, int lk, bool vstack = false) {
    constexpr int NT = NB * 16;
    const int outH = 2 * p.srcH, outW = 2 * p.srcW;
    const long long HWo = (long long)outH * outW;
    float* outn = p.out + (long long)n * p.out_bstride;
    const bool vec_ok = ((outW & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0) && ((p.out_bstride & 3) == 0);
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const int mb = wave * MW + mi;
        // vstack: a wave's m-blocks are MW vertically adjacent rows of one 16-pixel segment (9-product kernel)
        const int oyl = vstack ? (wave / twb) * MW + mi : mb / twb, seg = vstack ? wave % twb : mb - oyl * twb;
        const int y = y0 + oyl;
        const int xl = x0 + seg * 16 + lk * 4;
        if (y >= p.srcH || xl >= p.srcW) continue;
        const int Y = 2 * y + a, X = 2 * xl;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int oc = nt * NT + nb * 16 + li;
            if (oc >= p.OC) continue;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e0 = acc[0][mi][nb][r], e1 = acc[1][mi][nb][r];
                if (p.act) { e0 = leaky_relu(e0, p.slope); e1 = leaky_relu(e1, p.slope); }
                v[2 * r] = e0;
                v[2 * r + 1] = e1;
            }
            float* o = outn + (long long)oc * HWo + (long long)Y * outW + X;
            if (vec_ok && xl + 3 < p.srcW) {
                reinterpret_cast<f32x4*>(o)[0] = (f32x4){v[0], v[1], v[2], v[3]};
                reinterpret_cast<f32x4*>(o)[1] = (f32x4){v[4], v[5], v[6], v[7]};
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (xl + (r >> 1) < p.srcW) o[r] = v[r];
            }
        }
    }
}

template <int CK, int NB, int MW, int MAXPOS>
__global__ __launch_bounds__(256, KBN_WAVES_PER_SIMD) void conv_up2x_kernel(const Up2xParams p) {
    constexpr int NT = NB * 16;
    constexpr int NC4 = CK / 4;
    constexpr int B_FLOATS = CK * 8 * NT;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int a_floats = CK * p.plane;
    const int buf_floats = a_floats + B_FLOATS;

    const int tid = threadIdx.x;
    int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int a = bid & 1;  // row phase
    bid >>= 1;
    const int nt = bid % p.nTilesN;
    bid /= p.nTilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int TW = p.TWB * 16;
    const int y0 = ty * p.TH, x0 = tx * TW;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;

    // ---- staging table (low-res rows y0-1+a .. y0+TH-1+a, cols x0-1 .. x0+TW) ----
    int goff[MAXPOS], loff[MAXPOS];
#pragma unroll
    for (int u = 0; u < MAXPOS; ++u) {
        const int pos = tid + u * 256;
        int g = -1, l = -1;
        if (pos < p.rowsS * p.colsS) {
            const int r = pos / p.colsS, ci = pos - r * p.colsS;
            const int Y = y0 - 1 + a + r, X = x0 - 1 + ci;
            l = r * p.pitch + ci;
            if (Y >= 0 && Y < p.srcH && X >= 0 && X < p.srcW) g = Y * p.srcW + X;
        }
        goff[u] = g;
        loff[u] = l;
    }

    int mbase[MW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const int mb = wave * MW + mi;
        const int oy = mb / p.TWB, seg = mb - oy * p.TWB;
        mbase[mi] = oy * p.pitch + seg * 16 + li + lk * p.plane;
    }
    const int boff = (lk >> 1) * 2 * NT + li * 2 + (lk & 1);

    f32x4 acc[2][MW][NB];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[b][mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int HW = p.srcH * p.srcW;
    const float* srcn = p.src + (long long)n * p.src_bstride;
    const float* wp_a = p.wp + ((long long)a * p.nTilesN + nt) * p.Cpad * 8 * NT;

    auto load_chunk = [&](int c0, float (&va)[MAXPOS][CK]) -> int {
        const int nvalid = (p.Cin - c0 < CK) ? (p.Cin - c0) : CK;
        const float* base = srcn + (long long)c0 * HW;
#pragma unroll
        for (int u = 0; u < MAXPOS; ++u) {
            const int gi = goff[u] < 0 ? 0 : goff[u];
#pragma unroll
            for (int q = 0; q < CK; ++q) va[u][q] = base[(q < nvalid ? q : 0) * HW + gi];  // masked at the store
        }
        return nvalid;
    };
    auto store_chunk = [&](float* As, const float (&va)[MAXPOS][CK], int nvalid) {
#pragma unroll
        for (int u = 0; u < MAXPOS; ++u) {
            if (loff[u] >= 0) {
                const bool inb = goff[u] >= 0;
#pragma unroll
                for (int q = 0; q < CK; ++q) As[q * p.plane + loff[u]] = (inb && q < nvalid) ? va[u][q] : 0.f;
            }
        }
    };
    auto stage_B = [&](float* Bs, int c0) {
        constexpr int CNT4 = B_FLOATS / 4;
        const float4* s4 = reinterpret_cast<const float4*>(wp_a + (long long)c0 * 8 * NT);
        const unsigned bs = __builtin_amdgcn_readfirstlane(lds_addr(Bs));
#pragma unroll
        for (int e0 = 0; e0 < CNT4; e0 += 256) {
            const int eb = e0 + wave * 64;
            if (eb + lane < CNT4) lds_dma16_s(reinterpret_cast<const float*>(s4 + eb), (unsigned)(lane * 16), bs + eb * 16);
        }
    };
    auto compute = [&](const float* As, const float* Bs) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
            for (int c4 = 0; c4 < NC4; ++c4) {
                const float* Ab = As + c4 * 4 * p.plane + dy * p.pitch;
                const float* Bb = Bs + (dy * NC4 + c4) * 16 * NT + boff;
                float av[MW][3], bv[2][2][NB];
#pragma unroll
                for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                    for (int j = 0; j < 3; ++j) av[mi][j] = Ab[mbase[mi] + j];
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) bv[dx][b][nb] = Bb[(dx * 2 + b) * 4 * NT + nb * 32];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[b][mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi][b + dx], bv[dx][b][nb],
                                                                                     acc[b][mi][nb], 0, 0, 0);
            }
        }
    };

    float va[MAXPOS][CK];
    stage_B(smem + a_floats, 0);
    int nv = load_chunk(0, va);
    store_chunk(smem, va, nv);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int c0 = 0; c0 < p.Cpad; c0 += CK) {
        float* curA = smem + cur * buf_floats;
        float* nxtA = smem + (cur ^ 1) * buf_floats;
        const bool more = (c0 + CK < p.Cpad);
        if (more) {
            nv = load_chunk(c0 + CK, va);
            stage_B(nxtA + a_floats, c0 + CK);
        }
        compute(curA, curA + a_floats);
        if (more) store_chunk(nxtA, va, nv);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    up2x_store<NB, MW>(p, acc, n, nt, a, y0, x0, p.TWB, wave, li, lk);
}

// ------------------------------------------------------------------------------------------------
// Fast path: same 4-phase GEMM, but (a) the low-res tile reaches LDS by LDS-DMA as in conv_dma.hip
// (row-major image of the tile, 16-byte granules, out-of-image granules never written into the
// pre-zeroed stages; needs srcW % 4 == 0 and 16-byte aligned planes), (b) the tile geometry (MW, TWB)
// is a template argument and the K loop is unrolled by the stage parity, so that every LDS address is
// a loop-invariant register plus an immediate, and (c) the staging pointers live in SGPRs.  Vector-ALU
// instructions between MFMAs cost matrix-pipe time (tools/probe/issue_probe.hip); the general kernel
// above spends ~0.3 of them per MFMA on addresses and masks, this one none.
// GR = floats per DMA granule: 4 (16-byte granules, maps with W % 4 == 0) or 1 (dword granules, any map --
// e.g. the 11 x 38 latent of a KITTI frame).
template <int NB, int MW, int TWB, int GR, bool T3 = false>
struct Up2xGeom {
    static constexpr int NT = NB * 16;
    static constexpr int TH = 4 * MW / TWB, TW = TWB * 16;
    static constexpr int XPAD = (GR == 4) ? 4 : 1;               // staged columns start at x0 - XPAD
    static constexpr int ROWS = TH + 1, COLS = (GR == 4) ? TW + 8 : TW + 2;
    static constexpr int PLANE = ((ROWS * COLS + 15) / 32) * 32 + 16;
    static constexpr int NF4 = ROWS * COLS / GR;                 // granules per channel
    static constexpr int MAXJ = (NF4 + 63) / 64;
    static constexpr int A_FLOATS = 8 * PLANE, B_FLOATS = 8 * (T3 ? 6 : 8) * NT, BUF = A_FLOATS + B_FLOATS;
};

template <int NB, int MW, int TWB, int GR, bool T3>
__global__ __launch_bounds__(256, KBN_WAVES_PER_SIMD) void conv_up2x_dma_kernel(const Up2xParams p) {
    using G = Up2xGeom<NB, MW, TWB, GR, T3>;
    constexpr int NT = G::NT, PLANE = G::PLANE, PITCH = G::COLS;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int a = bid & 1;  // row phase
    bid >>= 1;
    const int nt = bid % p.nTilesN;
    bid /= p.nTilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int y0 = ty * G::TH, x0 = tx * G::TW;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;

    // this lane's granules of a channel tile: byte offset inside the source plane, or masked off
    unsigned gv[G::MAXJ];
    unsigned long long gm[G::MAXJ];
#pragma unroll
    for (int j = 0; j < G::MAXJ; ++j) {
        const int f = j * 64 + lane;
        int g = -1;
        if (f < G::NF4) {
            const int r = f / (PITCH / GR), cv = f - r * (PITCH / GR);
            const int Y = y0 - 1 + a + r, X = x0 - G::XPAD + cv * GR;
            if (Y >= 0 && Y < p.srcH && X >= 0 && X < p.srcW) g = (Y * p.srcW + X) * 4;
        }
        gv[j] = g < 0 ? 0u : (unsigned)g;
        gm[j] = __ballot(g >= 0);
    }

    int mbase[MW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const int mb = wave * MW + mi;
        const int oy = mb / TWB, seg = mb - oy * TWB;
        mbase[mi] = oy * PITCH + seg * 16 + li + (G::XPAD - 1) + lk * PLANE;   // column x-1 of the lane's pixel
    }
    const int boff = G::A_FLOATS + (lk >> 1) * 2 * NT + li * 2 + (lk & 1);

    constexpr int NACC = T3 ? 3 : 2;   // T3: left-difference, centre and right-difference products
    f32x4 acc[NACC][MW][NB];
#pragma unroll
    for (int b = 0; b < NACC; ++b)
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[b][mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int HW = p.srcH * p.srcW;
    // wave-uniform staging state (SGPRs): plane of channel (c0 + wave) and this wave's weight granules
    const float* aptr = uniform_ptr(p.src + (long long)n * p.src_bstride + (long long)wave * HW);
    const float* bptr = T3 ? uniform_ptr(p.wp3 + ((long long)a * p.nTilesN + nt) * p.Cpad * 6 * NT)
                           : uniform_ptr(p.wp + ((long long)a * p.nTilesN + nt) * p.Cpad * 8 * NT + wave * 256);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const unsigned uv = (unsigned)(lane * 16);

    auto stage = [&](int buf) {   // chunk at aptr / bptr -> stage `buf`; wave w moves channels w and w + 4
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned dst = lds0 + 4u * (unsigned)(buf * G::BUF + (wave + 4 * t) * PLANE);
#pragma unroll
            for (int j = 0; j < G::MAXJ; ++j) {
                if constexpr (GR == 4) lds_dma16_sm(aptr + (long long)(4 * t) * HW, gv[j], dst + j * 1024, gm[j]);
                else lds_dma4_sm(aptr + (long long)(4 * t) * HW, gv[j], dst + j * 256, gm[j]);
            }
        }
        if constexpr (T3) {   // 48 * NT floats: not always a whole number of 1 KiB rounds -> masked tail
            constexpr int n4 = G::B_FLOATS / 4;
            const unsigned bdst = lds0 + 4u * (unsigned)(buf * G::BUF + G::A_FLOATS);
#pragma unroll
            for (int e0 = 0; e0 < n4; e0 += 256) {
                const int eb = e0 + wave * 64;
                if (eb + lane < n4) lds_dma16_s(bptr + eb * 4, uv, bdst + eb * 16);
            }
        } else {
            const unsigned bdst = lds0 + 4u * (unsigned)(buf * G::BUF + G::A_FLOATS + wave * 256);
#pragma unroll
            for (int e = 0; e < G::B_FLOATS / 1024; ++e) lds_dma16_s(bptr + e * 1024, uv, bdst + e * 4096);
        }
        aptr += (long long)8 * HW;
        bptr += G::B_FLOATS;
    };
    auto compute = [&](auto par) {
        constexpr int PAR = decltype(par)::value;
        const float* S = smem + PAR * G::BUF;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4) {
                const float* Ab = S + c4 * 4 * PLANE + dy * PITCH;
                float av[MW][3];
#pragma unroll
                for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                    for (int j = 0; j < 3; ++j) av[mi][j] = Ab[mbase[mi] + j];
                if constexpr (T3) {
                    const float* Bb = S + (dy * 2 + c4) * 12 * NT + boff;
                    float b3[3][NB];
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) b3[m][nb] = Bb[m * 4 * NT + nb * 32];
#pragma unroll
                    for (int mi = 0; mi < MW; ++mi) {
                        const float dl = av[mi][1] - av[mi][0], dr = av[mi][2] - av[mi][1];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            acc[0][mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(dl, b3[0][nb], acc[0][mi][nb], 0, 0, 0);
                            acc[1][mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi][1], b3[1][nb], acc[1][mi][nb], 0, 0, 0);
                            acc[NACC - 1][mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(dr, b3[2][nb], acc[NACC - 1][mi][nb], 0, 0, 0);
                        }
                    }
                    continue;
                }
                const float* Bb = S + (dy * 2 + c4) * 16 * NT + boff;
                float bv[2][2][NB];
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) bv[dx][b][nb] = Bb[(dx * 2 + b) * 4 * NT + nb * 32];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[b][mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi][b + dx], bv[dx][b][nb],
                                                                                     acc[b][mi][nb], 0, 0, 0);
            }
        }
    };

    // clear the A part of both stages once (out-of-image granules are never written afterwards)
    {
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int e = tid * 4; e < G::A_FLOATS; e += 1024) {
            *reinterpret_cast<f32x4*>(smem + e) = zero;
            *reinterpret_cast<f32x4*>(smem + G::BUF + e) = zero;
        }
    }
    __syncthreads();
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nch = p.Cpad / 8;   // even (launcher)
    for (int c = 0; c < nch; c += 2) {
        stage(1);
        compute(IC2<0>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 2 < nch) stage(0);
        compute(IC2<1>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if constexpr (T3) {   // o0 = L + C, o1 = R + C
        f32x4 o[2][MW][NB];
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                o[0][mi][nb] = acc[0][mi][nb] + acc[1][mi][nb];
                o[1][mi][nb] = acc[NACC - 1][mi][nb] + acc[1][mi][nb];
            }
        up2x_store<NB, MW>(p, o, n, nt, a, y0, x0, TWB, wave, li, lk);
    } else {
        up2x_store<NB, MW>(p, acc, n, nt, a, y0, x0, TWB, wave, li, lk);
    }
}

template <int NB, int MW, int TWB, int GR = 4, bool T3 = false>
static int up2x_dma_variant(Up2xParams& p, hipStream_t stream) {
    using G = Up2xGeom<NB, MW, TWB, GR, T3>;
    auto kern = conv_up2x_dma_kernel<NB, MW, TWB, GR, T3>;
    static DeviceOnce once;
    if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
    p.TWB = TWB; p.TH = G::TH;
    p.tilesX = ceil_div(p.srcW, G::TW); p.tilesY = ceil_div(p.srcH, G::TH);
    const long long nb64 = (long long)p.tilesX * p.tilesY * p.N * p.nTilesN * 2;
    if (nb64 > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.nblocks = (int)nb64;
    hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), 2 * sizeof(float) * G::BUF, stream, p);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

template <int NB, int MW, int MAXPOS>
static int up2x_variant(const Up2xParams& p, size_t lds, hipStream_t stream) {
    auto kern = conv_up2x_kernel<8, NB, MW, MAXPOS>;
    static DeviceOnce once;
    if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
    if (p.rowsS * p.colsS > MAXPOS * 256) return KBN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), lds, stream, p);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

// ------------------------------------------------------------------------------------------------
// 9-product form: the 3-product identity (pack_up2x3_kernel) along the rows as well.  A workgroup covers BOTH row
// phases of its low-res tile: per pixel and channel the 3x3 neighbourhood gives 9 values -- rows (in[y]-in[y-1],
// in[y], in[y+1]-in[y]) x columns likewise -- and 9 channel-GEMMs with the weights T w T^t,
// T = [[-1,0,0],[1,1,1],[0,0,1]], accumulate P[i][m]; the four outputs of the pixel are
// o(a,b) = P[ia][jb] + P[ia][C] + P[C][jb] + P[C][C]  (ia = L for the upper output row, R for the lower; jb likewise).
// 9 products per low-res pixel instead of the 12 of two 3-product row phases (16 of the plain 4-phase form); the
// 12 subtractions per fragment run on the vector ALU.  9 accumulator sets limit the tile to 2 n-blocks x 2 m-blocks
// (or 1 x 4 for <= 16 filters).
struct Up2x9Plan { int NB, MW, NT, nTilesN; };
__host__ __device__ inline bool up2x9_eligible(int oc) { return oc <= 16 || (oc % 32) == 0; }
__host__ __device__ inline Up2x9Plan make_up2x9_plan(int oc) {
    Up2x9Plan q;
    if (oc <= 16) { q.NB = 1; q.MW = 4; } else { q.NB = 2; q.MW = 2; }
    q.NT = q.NB * 16;
    q.nTilesN = ceil_div(oc, q.NT);
    return q;
}

// layout per n-tile: [chunk of 8 ch][c4][i*3+m][4 x NT fragment block]
__global__ void pack_up2x9_kernel(const float* __restrict__ w, float* __restrict__ packed, int OC, int Cin, int Cpad,
                                  Up2x9Plan q, long long total) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int NT = q.NT;
    const long long per_nt = (long long)Cpad * 9 * NT;
    const int nt = (int)(e / per_nt);
    int r = (int)(e - nt * per_nt);
    const int chunk = r / (72 * NT); r -= chunk * 72 * NT;
    const int c4 = r / (36 * NT); r -= c4 * 36 * NT;
    const int im = r / (4 * NT); r -= im * 4 * NT;
    const int khalf = r / (2 * NT); r -= khalf * 2 * NT;
    const int nn = r >> 1, klow = r & 1;
    const int c = chunk * 8 + c4 * 4 + khalf * 2 + klow;
    const int oc = nt * NT + nn;
    float v = 0.f;
    if (c < Cin && oc < OC) {
        const float* wk = w + ((long long)oc * Cin + c) * 9;
        const double T[3][3] = {{-1.0, 0.0, 0.0}, {1.0, 1.0, 1.0}, {0.0, 0.0, 1.0}};
        const int i = im / 3, m = im - i * 3;
        double acc = 0.0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) acc += T[i][ky] * (double)wk[ky * 3 + kx] * T[m][kx];
        v = (float)acc;
    }
    packed[e] = v;
}

template <int NB, int MW, int TWB>
struct Up2x9Geom {
    static constexpr int NT = NB * 16;
    static constexpr int TH = 4 * MW / TWB, TW = TWB * 16;
    static constexpr int ROWS = TH + 2, COLS = TW + 8;            // rows y0-1 .. y0+TH, columns x0-4 .. x0+TW+3
    static constexpr int PLANE = ((ROWS * COLS + 15) / 32) * 32 + 16;
    static constexpr int NF4 = ROWS * COLS / 4;
    static constexpr int MAXJ = (NF4 + 63) / 64;
    static constexpr int A_FLOATS = 8 * PLANE, B_FLOATS = 72 * NT, BUF = A_FLOATS + B_FLOATS;
};

template <int NB, int MW, int TWB>
__global__ __launch_bounds__(256, 2) void conv_up2x9_kernel(const Up2xParams p) {
    using G = Up2x9Geom<NB, MW, TWB>;
    constexpr int NT = G::NT, PLANE = G::PLANE, PITCH = G::COLS;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int nt = bid % p.nTilesN;
    bid /= p.nTilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int y0 = ty * G::TH, x0 = tx * G::TW;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;

    unsigned gv[G::MAXJ];
    unsigned long long gm[G::MAXJ];
#pragma unroll
    for (int j = 0; j < G::MAXJ; ++j) {
        const int f = j * 64 + lane;
        int g = -1;
        if (f < G::NF4) {
            const int r = f / (PITCH / 4), cv = f - r * (PITCH / 4);
            const int Y = y0 - 1 + r, X = x0 - 4 + cv * 4;
            if (Y >= 0 && Y < p.srcH && X >= 0 && X < p.srcW) g = (Y * p.srcW + X) * 4;
        }
        gv[j] = g < 0 ? 0u : (unsigned)g;
        gm[j] = __ballot(g >= 0);
    }

    int mbase[MW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const int oy = (wave / TWB) * MW + mi, seg = wave % TWB;  // a wave owns MW vertically adjacent rows of one segment
        mbase[mi] = oy * PITCH + seg * 16 + li + 3 + lk * PLANE;   // (row y-1, column x-1) of the lane's pixel
    }
    const int boff = G::A_FLOATS + (lk >> 1) * 2 * NT + li * 2 + (lk & 1);

    f32x4 acc[9][MW][NB];   // first written by the zero-C products of the tile's first k-step

    const int HW = p.srcH * p.srcW;
    const float* aptr = uniform_ptr(p.src + (long long)n * p.src_bstride + (long long)wave * HW);
    const float* bptr = uniform_ptr(p.wp9 + (long long)nt * p.Cpad * 9 * NT);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const unsigned uv = (unsigned)(lane * 16);

    auto stage = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned dst = lds0 + 4u * (unsigned)(buf * G::BUF + (wave + 4 * t) * PLANE);
#pragma unroll
            for (int j = 0; j < G::MAXJ; ++j) lds_dma16_sm(aptr + (long long)(4 * t) * HW, gv[j], dst + j * 1024, gm[j]);
        }
        constexpr int n4 = G::B_FLOATS / 4;
        const unsigned bdst = lds0 + 4u * (unsigned)(buf * G::BUF + G::A_FLOATS);
#pragma unroll
        for (int e0 = 0; e0 < n4; e0 += 256) {
            const int eb = e0 + wave * 64;
            if (eb + lane < n4) lds_dma16_s(bptr + eb * 4, uv, bdst + eb * 16);
        }
        aptr += (long long)8 * HW;
        bptr += G::B_FLOATS;
    };
    // `first` (first k-step of a tile): the products go into a zero C operand -- no accumulator is ever cleared by
    // hand (9 x MW x NB x 4 v_mov per tile that would sit on the matrix pipe's issue port).
    auto compute = [&](auto par, auto first_c) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool FIRST = decltype(first_c)::value != 0;
        const float* S = smem + PAR * G::BUF;
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            const float* Ab = S + c4 * 4 * PLANE;
            const float* Bb = S + c4 * 36 * NT + boff;
            float d[MW][9], b9[9][NB];
            {
                // The wave's MW m-blocks are vertically adjacent rows of the tile: read the MW + 2 input rows once,
                // take the column differences per input row, then the row differences between neighbouring rows --
                // the lower difference of a row is the upper difference of the next one.  2 (MW + 2) + 3 (MW + 1)
                // subtractions and 3 (MW + 2) reads instead of 12 MW and 9 MW (vector instructions cost MFMA time).
                float cv[MW + 2][3];
#pragma unroll
                for (int i = 0; i < MW + 2; ++i) {
                    const float x0v = Ab[mbase[0] + i * PITCH], x1v = Ab[mbase[0] + i * PITCH + 1], x2v = Ab[mbase[0] + i * PITCH + 2];
                    cv[i][0] = x1v - x0v; cv[i][1] = x1v; cv[i][2] = x2v - x1v;
                }
                float rd[MW + 1][3];
#pragma unroll
                for (int i = 0; i < MW + 1; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) rd[i][j] = cv[i + 1][j] - cv[i][j];
#pragma unroll
                for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        d[mi][0 * 3 + j] = rd[mi][j];        // upper difference: row y - row y-1
                        d[mi][1 * 3 + j] = cv[mi + 1][j];    // centre row
                        d[mi][2 * 3 + j] = rd[mi + 1][j];    // lower difference: row y+1 - row y
                    }
            }
#pragma unroll
            for (int q = 0; q < 9; ++q)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b9[q][nb] = Bb[q * 4 * NT + nb * 32];
#pragma unroll
            for (int q = 0; q < 9; ++q)
#pragma unroll
                for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        if (FIRST && c4 == 0)
                            acc[q][mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[mi][q], b9[q][nb], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        else
                            acc[q][mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[mi][q], b9[q][nb], acc[q][mi][nb], 0, 0, 0);
                    }
        }
    };

    {
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int e = tid * 4; e < G::A_FLOATS; e += 1024) {
            *reinterpret_cast<f32x4*>(smem + e) = zero;
            *reinterpret_cast<f32x4*>(smem + G::BUF + e) = zero;
        }
    }
    __syncthreads();
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nch = p.Cpad / 8;   // even (launcher)
    stage(1);
    compute(IC2<0>{}, IC2<1>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (2 < nch) stage(0);
    compute(IC2<1>{}, IC2<0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 2; c < nch; c += 2) {
        stage(1);
        compute(IC2<0>{}, IC2<0>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 2 < nch) stage(0);
        compute(IC2<1>{}, IC2<0>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

#pragma unroll
    for (int a = 0; a < 2; ++a) {                       // upper / lower output row of every low-res row
        const int ia = a == 0 ? 0 : 2;
        f32x4 o[2][MW][NB];
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f32x4 cc = acc[4][mi][nb];
                o[0][mi][nb] = (acc[ia * 3 + 0][mi][nb] + acc[ia * 3 + 1][mi][nb]) + (acc[3][mi][nb] + cc);
                o[1][mi][nb] = (acc[ia * 3 + 2][mi][nb] + acc[ia * 3 + 1][mi][nb]) + (acc[5][mi][nb] + cc);
            }
        up2x_store<NB, MW>(p, o, n, nt, a, y0, x0, TWB, wave, li, lk, true);
    }
}

template <int NB, int MW, int TWB>
static int up2x9_variant(Up2xParams& p, hipStream_t stream) {
    using G = Up2x9Geom<NB, MW, TWB>;
    auto kern = conv_up2x9_kernel<NB, MW, TWB>;
    static DeviceOnce once;
    if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
    p.TWB = TWB; p.TH = G::TH;
    p.tilesX = ceil_div(p.srcW, G::TW); p.tilesY = ceil_div(p.srcH, G::TH);
    const long long nb64 = (long long)p.tilesX * p.tilesY * p.N * p.nTilesN;
    if (nb64 > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.nblocks = (int)nb64;
    hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), 2 * sizeof(float) * G::BUF, stream, p);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

// (NB, half-size tile, TWB) -> instantiated LDS-DMA kernel, plain 4-phase (T3 = false) or 3-product form
template <bool T3, int GR>
static int up2x_dma_pick(Up2xParams& q, int NB, int half, int t, hipStream_t st) {
    if constexpr (GR == 4) {
        switch (NB * 100 + half * 10 + t) {
            case 101: return up2x_dma_variant<1, 4, 1, 4, T3>(q, st);
            case 102: return up2x_dma_variant<1, 4, 2, 4, T3>(q, st);
            case 201: return up2x_dma_variant<2, 4, 1, 4, T3>(q, st);
            case 202: return up2x_dma_variant<2, 4, 2, 4, T3>(q, st);
            default: break;
        }
    }
    if (NB < 3) return KBN_ERR_UNSUPPORTED;
    switch ((NB == 3 ? 300 : 400) + half * 10 + t) {   // dword granules (GR = 1): wide outputs only
        case 301: return up2x_dma_variant<3, 2, 1, GR, T3>(q, st);
        case 302: return up2x_dma_variant<3, 2, 2, GR, T3>(q, st);
        case 311: return up2x_dma_variant<3, 1, 1, GR, T3>(q, st);
        case 312: return up2x_dma_variant<3, 1, 2, GR, T3>(q, st);
        case 401: return up2x_dma_variant<4, 2, 1, GR, T3>(q, st);
        case 402: return up2x_dma_variant<4, 2, 2, GR, T3>(q, st);
        case 411: return up2x_dma_variant<4, 1, 1, GR, T3>(q, st);
        default: return up2x_dma_variant<4, 1, 2, GR, T3>(q, st);
    }
}

}  // namespace kbn

extern "C" {

size_t kbn_upconv2x_packed_weight_bytes(int out_channels, int in_channels) {
    if (out_channels < 1 || in_channels < 1) return 0;
    kbn::Up2xPlan pl = kbn::make_up2x_plan(out_channels, in_channels);
    size_t floats = 2 * (size_t)pl.nTilesN * pl.Cpad * (8 + 6) * pl.NT;   // 4-phase weights + 3-product weights
    if (kbn::up2x9_eligible(out_channels)) {                               // + 9-product weights
        const kbn::Up2x9Plan q = kbn::make_up2x9_plan(out_channels);
        floats += (size_t)q.nTilesN * pl.Cpad * 9 * q.NT;
    }
    return sizeof(float) * floats;
}

int kbn_upconv2x_pack_weight(const float* weight, float* packed, int out_channels, int in_channels,
                             kbn_stream_t stream) {
    if (!weight || !packed || out_channels < 1 || in_channels < 1) return KBN_ERR_INVALID_ARGUMENT;
    kbn::Up2xPlan pl = kbn::make_up2x_plan(out_channels, in_channels);
    long long total = 2LL * pl.nTilesN * pl.Cpad * 8 * pl.NT;
    hipLaunchKernelGGL(kbn::pack_up2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, weight, packed, out_channels, in_channels, pl, total, 0);
    KBN_CHECK_LAUNCH();
    const long long total3 = 2LL * pl.nTilesN * pl.Cpad * 6 * pl.NT;
    hipLaunchKernelGGL(kbn::pack_up2x3_kernel, dim3((unsigned)((total3 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, weight, packed + total, out_channels, in_channels, pl, total3);
    KBN_CHECK_LAUNCH();
    if (kbn::up2x9_eligible(out_channels)) {
        const kbn::Up2x9Plan q = kbn::make_up2x9_plan(out_channels);
        const long long total9 = (long long)q.nTilesN * pl.Cpad * 9 * q.NT;
        hipLaunchKernelGGL(kbn::pack_up2x9_kernel, dim3((unsigned)((total9 + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, weight, packed + total + total3, out_channels, in_channels, pl.Cpad, q, total9);
        KBN_CHECK_LAUNCH();
    }
    return KBN_OK;
}

static int upconv2x_forward_impl(const float* src, long long src_batch_stride, const float* packed_weight, float* out,
                                 long long out_batch_stride, int n, int in_channels, int out_channels, int src_height,
                                 int src_width, int apply_activation, float negative_slope, kbn_stream_t stream, bool plain = false) {
    // `plain`: the blob holds four-phase weights only (kbn_deconv2x_pack_weight) -- the 3- and 9-product forms rest on the nearest-2x fold
    using namespace kbn;
    if (!src || !packed_weight || !out || n < 1 || in_channels < 1 || out_channels < 1 || src_height < 1 ||
        src_width < 1)
        return KBN_ERR_INVALID_ARGUMENT;
    if (src_height > 16383 || src_width > 16383) return KBN_ERR_UNSUPPORTED;
    const Up2xPlan pl = make_up2x_plan(out_channels, in_channels);
    Up2xParams p;
    p.src = src; p.wp = packed_weight; p.out = out;
    p.wp3 = packed_weight + 2LL * pl.nTilesN * pl.Cpad * 8 * pl.NT;
    p.wp9 = p.wp3 + 2LL * pl.nTilesN * pl.Cpad * 6 * pl.NT;
    p.src_bstride = src_batch_stride; p.out_bstride = out_batch_stride;
    p.N = n; p.Cin = in_channels; p.Cpad = pl.Cpad; p.OC = out_channels;
    p.srcH = src_height; p.srcW = src_width;
    p.act = apply_activation ? 1 : 0; p.slope = negative_slope;
    p.nTilesN = pl.nTilesN;
    hipStream_t st = (hipStream_t)stream;
    // fast path: LDS-DMA staging, compile-time tile geometry
    const bool aligned = (src_width & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
                         (src_batch_stride & 3) == 0 && (pl.Cpad / 8) % 2 == 0 && in_channels == pl.Cpad;
    if (aligned) {
        const int ftw = knob(KNOB_FORCE_TWB);
        int twb = 1;   // 16-wide tiles unless 32-wide ones waste fewer pixels / fill the rounds better
        {
            double best = 1e300;
            for (int t = 1; t <= 2; ++t) {
                if (ftw && ftw != t) continue;
                const int th = 4 * pl.MW / t, tw = t * 16;
                const long long tiles = (long long)ceil_div(src_width, tw) * ceil_div(src_height, th) * n * pl.nTilesN * 2;
                const double cost = (double)((tiles + 511) / 512) * (4 * pl.MW * 16.0 + 0.1 * (th + 1) * (tw + 8));
                if (cost < best) { best = cost; twb = t; }
            }
        }
        // 9-product form (both row phases in one workgroup) where the filter count allows; like the 3-product form a
        // property of the layer (it rounds differently), never a tuner choice -- the tuner only picks the tile width.
        if (up2x9_eligible(out_channels) && !knob(KNOB_NO_UP2X9) && !knob(KNOB_NO_UP2X3) && !plain) {
            const Up2x9Plan q9 = make_up2x9_plan(out_channels);
            auto launch9 = [&](int cand) -> int {   // candidate = TWB - 1
                Up2xParams q = p;
                q.nTilesN = q9.nTilesN;
                if (q9.NB == 1) return cand ? up2x9_variant<1, 4, 2>(q, st) : up2x9_variant<1, 4, 1>(q, st);
                return cand ? up2x9_variant<2, 2, 2>(q, st) : up2x9_variant<2, 2, 1>(q, st);
            };
            int c9 = twb - 1;
            if (!ftw) c9 = tune_pick(TuneKey{7, n, out_channels, in_channels, src_height, src_width, 0, 0, 0, 0}, 2, c9, launch9, st);
            return launch9(c9);
        }
        // The 3-product form (3/4 of the MFMAs, see pack_up2x3_kernel) is a property of the layer, not a tuning
        // choice: it rounds differently from the 4-phase form, and results must not depend on the batch size or on
        // what the tuner measured.  The tuner only picks the tile shape (bit-identical among themselves).
        const int t3 = (knob(KNOB_NO_UP2X3) || plain) ? 0 : 1;
        auto launch = [&](int cand) -> int {   // candidate = (TWB - 1) + 2 * (half-size tile)
            Up2xParams q = p;
            const int t = (cand & 1) + 1, half = (cand >> 1) & 1;
            if (half && pl.NB < 3) return (int)KBN_ERR_UNSUPPORTED;
            return t3 ? up2x_dma_pick<true, 4>(q, pl.NB, half, t, st) : up2x_dma_pick<false, 4>(q, pl.NB, half, t, st);
        };
        int cand = twb - 1;
        if (!ftw) cand = tune_pick(TuneKey{3, n, out_channels, in_channels, src_height, src_width, t3, 0, 0, 0}, 4, cand, launch, st);
        return launch(cand);
    }
    // maps whose rows are not 16-byte aligned: the same kernel with dword DMA granules (wide outputs only)
    if (!aligned && pl.NB >= 3 && (pl.Cpad / 8) % 2 == 0 && in_channels == pl.Cpad) {
        const int t3 = (knob(KNOB_NO_UP2X3) || plain) ? 0 : 1;
        auto launch = [&](int cand) -> int {   // candidate = (TWB - 1) + 2 * (half-size tile)
            Up2xParams q = p;
            const int t = (cand & 1) + 1, half = (cand >> 1) & 1;
            return t3 ? up2x_dma_pick<true, 1>(q, pl.NB, half, t, st) : up2x_dma_pick<false, 1>(q, pl.NB, half, t, st);
        };
        const long long tiles2 = (long long)ceil_div(src_width, 16) * ceil_div(src_height, 4 * pl.MW) * n * pl.nTilesN * 2;
        const int model = tiles2 <= 512 ? 2 : 0;   // small maps: half-size tiles (see below)
        const int cand = tune_pick(TuneKey{4, n, out_channels, in_channels, src_height, src_width, t3, 0, 0, 0}, 4, model, launch, st);
        return launch(cand);
    }
    // tile: 4*MW m-blocks of 16 low-res pixels, 16 or 32 wide; a lone workgroup round is avoided.
    // Small maps (every tile of the launch resident at once) halve the tile: twice the workgroups, and the
    // waves sharing a SIMD finish sooner than one wave with the double tile.
    int mw = pl.MW;
    {
        const long long tiles2 = (long long)ceil_div(src_width, 16) * ceil_div(src_height, 4 * pl.MW) * n * pl.nTilesN * 2;
        if (pl.NB >= 3 && tiles2 <= 512) mw = 1;
        const int fm = 0;
        if (fm && pl.NB >= 3) mw = fm == 1 ? 1 : pl.MW;
    }
    const int mblocks = 4 * mw;
    double best_cost = 1e300;
    int best_twb = 1;
    const int ft = knob(KNOB_FORCE_TWB);
    for (int twb = 1; twb <= 2; ++twb) {
        if (ft && ft != twb) continue;
        const int th = mblocks / twb, tw = twb * 16;
        long long tiles = (long long)ceil_div(src_width, tw) * ceil_div(src_height, th) * n * pl.nTilesN * 2;
        double cost = (double)((tiles + 255) / 256) * (mblocks * 16.0 + 0.1 * (th + 1) * (tw + 2));
        if (cost < best_cost) { best_cost = cost; best_twb = twb; }
    }
    p.TWB = best_twb; p.TH = mblocks / best_twb;
    const int TW = best_twb * 16;
    p.tilesX = ceil_div(src_width, TW); p.tilesY = ceil_div(src_height, p.TH);
    p.rowsS = p.TH + 1; p.colsS = TW + 2; p.pitch = TW + 2;
    int plane = p.rowsS * p.pitch;
    p.plane = ((plane + 15) / 32) * 32 + 16;
    long long nb64 = (long long)p.tilesX * p.tilesY * n * pl.nTilesN * 2;
    if (nb64 > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.nblocks = (int)nb64;
    size_t lds = 2 * sizeof(float) * ((size_t)pl.CK * p.plane + (size_t)pl.CK * 8 * pl.NT);
    switch (pl.NB) {
        case 1: return up2x_variant<1, 4, 2>(p, lds, st);
        case 2: return up2x_variant<2, 4, 2>(p, lds, st);
        case 3: return mw == 1 ? up2x_variant<3, 1, 1>(p, lds, st) : up2x_variant<3, 2, 1>(p, lds, st);
        default: return mw == 1 ? up2x_variant<4, 1, 1>(p, lds, st) : up2x_variant<4, 2, 1>(p, lds, st);
    }
}

int kbn_upconv2x_forward(const float* src, long long src_batch_stride, const float* packed_weight, float* out,
                         long long out_batch_stride, int n, int in_channels, int out_channels, int src_height,
                         int src_width, int apply_activation, float negative_slope, unsigned* out_absmax, kbn_stream_t stream) {
    int rc = upconv2x_forward_impl(src, src_batch_stride, packed_weight, out, out_batch_stride, n, in_channels, out_channels,
                                   src_height, src_width, apply_activation, negative_slope, stream);
    // these fp32 kernels are fallbacks since the folded split-operand up-conv: the slot is filled by a pass of its own
    if (rc == KBN_OK && out_absmax && !kbn::knob(kbn::KNOB_NO_SPLIT))   // KBN_NO_SPLIT: nothing reads slots
        rc = kbn::absmax_frames_launch(out, out_batch_stride, n, 4LL * out_channels * src_height * src_width, out_absmax,
                                       (hipStream_t)stream);
    return rc;
}

int kbn_deconv2x_pack_weight(const float* weight, float* packed, int out_channels, int in_channels, kbn_stream_t stream) {
    if (!weight || !packed || out_channels < 1 || in_channels < 1) return KBN_ERR_INVALID_ARGUMENT;
    kbn::Up2xPlan pl = kbn::make_up2x_plan(out_channels, in_channels);
    const long long total = 2LL * pl.nTilesN * pl.Cpad * 8 * pl.NT;
    hipLaunchKernelGGL(kbn::pack_up2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, weight, packed, out_channels, in_channels, pl, total, 1);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_deconv2x_forward(const float* src, long long src_batch_stride, const float* packed_weight, float* out,
                         long long out_batch_stride, int n, int in_channels, int out_channels, int src_height,
                         int src_width, int apply_activation, float negative_slope, unsigned* out_absmax, kbn_stream_t stream) {
    int rc = upconv2x_forward_impl(src, src_batch_stride, packed_weight, out, out_batch_stride, n, in_channels, out_channels,
                                   src_height, src_width, apply_activation, negative_slope, stream, true);
    if (rc == KBN_OK && out_absmax && !kbn::knob(kbn::KNOB_NO_SPLIT))
        rc = kbn::absmax_frames_launch(out, out_batch_stride, n, 4LL * out_channels * src_height * src_width, out_absmax,
                                       (hipStream_t)stream);
    return rc;
}

/* Which algebraic form kbn_upconv2x_forward runs for a problem and what it executes:
 * info[4] = {channel products per low-resolution pixel (16 = four 2x2 phases, 12 = 3-product columns,
 * 9 = 3-product rows and columns), padded output channels, padded input channels, 0}.
 * Executed MFMA work = 2 * n * src_height * src_width * info[0] * info[1] * info[2] FLOP. */
int kbn_upconv2x_query(int n, int in_channels, int out_channels, int src_height, int src_width, int* info) {
    using namespace kbn;
    if (!info || n < 1 || in_channels < 1 || out_channels < 1 || src_height < 1 || src_width < 1)
        return KBN_ERR_INVALID_ARGUMENT;
    const Up2xPlan pl = make_up2x_plan(out_channels, in_channels);
    const bool chunks_ok = (pl.Cpad / 8) % 2 == 0 && in_channels == pl.Cpad;
    const bool aligned = (src_width & 3) == 0 && chunks_ok;   // plus 16-byte aligned planes, which torch tensors are
    const int t3 = knob(KNOB_NO_UP2X3) ? 0 : 1;
    info[0] = 16; info[1] = pl.nTilesN * pl.NT; info[2] = pl.Cpad; info[3] = 0;
    if (aligned) {
        if (up2x9_eligible(out_channels) && !knob(KNOB_NO_UP2X9) && t3) {
            const Up2x9Plan q9 = make_up2x9_plan(out_channels);
            info[0] = 9; info[1] = q9.nTilesN * q9.NT;
        } else {
            info[0] = t3 ? 12 : 16;
        }
    } else if (!aligned && pl.NB >= 3 && chunks_ok) {
        info[0] = t3 ? 12 : 16;
    }
    return KBN_OK;
}

}  // extern "C"
