// s2d.hip -- fused sparse-to-dense pool (S2D) for gfx950.
//
// Replaces networks.SparseToDensePool.forward (reference src/networks.py:2168-2196):
//   z = x[:, 0]; min-pools over the non-zero depths (999-sentinel semantics, :2175-2181),
//   max-pools (:2183-2186), cat, n_convolution x [conv1x1 + LeakyReLU], cat with x,
//   conv3x3 + LeakyReLU -- one launch, one read of x, one write of the n_filter maps.
//
// What bounds it.  Per pixel the layer moves 40 bytes (2 channels in, 8 out) but evaluates 904 multiply-adds
// (1x1 chain 184, 3x3 conv 720): 45 FLOP/B, above the chip's fp32 balance (157 TFLOP/s : 8 TB/s = 20), so the
// kernel is fp32-ARITHMETIC bound, not HBM bound -- and on gfx950 the fp32-input MFMAs and the vector ALU share
// one multiply-add datapath (tools/probe/mfma4_probe.hip: a matrix-only wave and a v_fma-only wave on one SIMD
// take the SUM of their times), so there is no second pipe to hide the convolutions behind.  The design
// therefore minimises everything that is not a multiply-add and keeps two workgroups resident per CU so that
// one's LDS / latency-bound phases overlap the other's arithmetic.
//
// One workgroup (256 threads) produces a 64 x 16 output tile:
//   P1  the sparse depth tile with a halo of R+1 (R = largest pool radius) is staged in LDS twice: `zmin`
//       (0 -> 999 sentinel, +inf outside the image) and `zmax` (-inf outside): coalesced row reads of x.
//   P2  vertical pass, register blocked: a thread owns one column and 6 consecutive rows, reads the 6+2R
//       values it needs ONCE and runs the nested sweep (one outward sweep yields every pool size) from
//       registers; results go to per-pool planes V.
//   P3  horizontal pass, register blocked: a thread owns 4 consecutive pixels of a row; per pool it reads the
//       2r+4 values of V it needs as 16-byte words and forms the four windows from a shared core + prefix /
//       suffix minima (2r+5 compares instead of 8r).  Exact 999-sentinel semantics, bit-exact pyramid.
//   P4  the 1x1 conv chain on v_mfma_f32_4x4x1_16B_f32, used as "4 multiply-adds per lane": lane = pixel,
//       B = the pixel's input value, A = 4 output-channel weights (lane&3 selects the channel), 2 instructions
//       per input channel for the 8 filters; LeakyReLU on the vector ALU; features + the raw x channels go to
//       LDS (overlaying V) as the 3x3 conv's input tile, zero outside the image like the reference's padding.
//   P5  the 3x3 conv the same way: a wave owns 64 columns x 4 rows; per input channel it loads its 18 A
//       operands (9 taps x 2 filter halves) and the 6 x 3 shifted B rows from LDS, then issues 72 MFMAs.
//       Accumulation order = channels ascending, taps inside, one fp32 FMA chain per output (exact fp32).
#include <math.h>

#include <initializer_list>

#include "front_common.h"
#include "kbn_common.h"
#include "s2d_pools.h"

namespace kbn {

constexpr int S2D_TW = 64, S2D_TH = 16, S2D_THREADS = 256;
constexpr int S2D_FW = S2D_TW + 2, S2D_FH = S2D_TH + 2;      // feature region the 3x3 conv needs (66 x 18)
constexpr int S2D_NQ = (S2D_FW + 3) / 4;                     // 4-pixel groups per feature row (17)
constexpr int S2D_FWP = S2D_NQ * 4;                          // feature row pitch in LDS (68 floats, 16-byte rows)
constexpr int S2D_GS = 6;                                    // rows per thread in the vertical pass (FH = 3 x 6)
constexpr int S2D_MAXPOOL = 8, S2D_MAXF = 8, S2D_MAXCONV = 4, S2D_MAXIN = 2, S2D_MAXR = 15;
constexpr int S2D_MAXCH = S2D_MAXF + S2D_MAXIN;
// LDS weight tables (floats), both in MFMA A-operand order [..][filter half][4 filters]:
//   w3[(ch * 9 + tap) * 8 + half * 4 + j]    = conv.weight[4 half + j][ch][tap]
//   w1[((layer * 8 + q) * 2 + half) * 4 + j] = pool_convs[layer].weight[4 half + j][q]   (0 beyond the fan-in)
constexpr int S2D_W3 = S2D_MAXCH * 9 * 8, S2D_W1 = S2D_MAXCONV * 8 * 8;
constexpr int S2D_WFLOATS = S2D_W3 + S2D_W1;
static_assert(S2D_FH % S2D_GS == 0 && S2D_TH % 4 == 0, "tile geometry");

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
    int nconv, nf, R, Rmin, Rmax;
    int tilesX, tilesY, nblocks;
    int dbg;   // phase ablation for tools/s2d_bench.py (KBN_S2D_DEBUG): 1 no vertical pass, 2 no horizontal pass, 4 no 1x1 chain, 8 no 3x3 conv, 16 no staging
    float slope;
};

// staged depth tile (FW + 2R) x (FH + 2R); V rows hold FWP + 2R columns (pitch a multiple of 4 floats)
__host__ __device__ constexpr int s2d_zw(int R) { return S2D_FW + 2 * R; }
__host__ __device__ constexpr int s2d_zh(int R) { return S2D_FH + 2 * R; }
__host__ __device__ constexpr int s2d_vp(int R) { return (S2D_FWP + 2 * R + 3) / 4 * 4; }

struct DynamicPools {
    static constexpr bool is_static = false;
    static constexpr int RMAXZ = S2D_MAXR;
    __device__ static int R(const S2DParams& p) { return p.R; }
};

// LeakyReLU as max(v, slope v): 2 vector instructions instead of 3 (0 <= slope <= 1, checked by the launcher)
__device__ __forceinline__ float s2d_lrelu(float v, float slope) {
    float t = v * slope, o;
    asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(v), "v"(t));
    return o;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);   // c[i] += A[4 * (lane / 4) + i] * B[lane]
}

template <typename CFG>
__global__ __launch_bounds__(S2D_THREADS, 2) void s2d_kernel(const S2DParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int R = CFG::R(p);                       // compile-time for the static presets
    const int ZW = s2d_zw(R), ZH = s2d_zh(R), VP = s2d_vp(R);
    const int VPLANE = S2D_FH * VP;
    float* w3 = smem;                      // weights, live for the whole kernel
    float* w1 = smem + S2D_W3;
    float* zmin = smem + S2D_WFLOATS;
    float* zmax = zmin + ZH * ZW;
    float* vbuf = smem + S2D_WFLOATS + ((2 * ZH * ZW + 3) & ~3);   // [pool][FH][VP], 16-byte aligned rows
    float* feat = vbuf;                    // [nf + inC][FH][FWP]: overlays V (dead once the pooled values are in registers)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long HW = (long long)p.H * p.W;
    const int nch = p.nf + p.inC;
    int npool = p.npool, nmin = p.nmin;
    if constexpr (CFG::is_static) { npool = CFG::NP; nmin = CFG::NMINP; }
    const float slope = p.slope;

    // Persistent workgroups (two per CU): tile t = blockIdx.x, blockIdx.x + gridDim.x, ...  The weight tables are
    // staged once, and the next tile's depth values are fetched into registers while this tile's passes and
    // convolutions run (the depth images in LDS are dead after the vertical pass).
    struct Tile { int n, oy0, ox0; };
    auto decode = [&](int t) {
        int bid = xcd_remap(t, p.nblocks);
        const int tx = bid % p.tilesX;
        bid /= p.tilesX;
        const int ty = bid % p.tilesY;
        return Tile{bid / p.tilesY, ty * S2D_TH, tx * S2D_TW};
    };
    constexpr int MAXE = (s2d_zh(CFG::RMAXZ) * s2d_zw(CFG::RMAXZ) + S2D_THREADS - 1) / S2D_THREADS;
    float vz[MAXE];
    auto z_load = [&](const Tile& tl) {   // this thread's elements of the depth tile (+halo): -inf outside the image
        const float* src = p.x + (long long)tl.n * p.x_bstride;
        const int Yt = tl.oy0 - 1 - R, Xl = tl.ox0 - 1 - R;
        const bool inside = Yt >= 0 && Yt + ZH <= p.H && Xl >= 0 && Xl + ZW <= p.W;   // block-uniform
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = u * S2D_THREADS + tid;
            const int r = e / ZW, c = e - r * ZW;
            const int Y = Yt + r, X = Xl + c;
            const bool ok = e < ZH * ZW && (inside || (Y >= 0 && Y < p.H && X >= 0 && X < p.W));
            vz[u] = ok ? src[(long long)Y * p.W + X] : -INFINITY;
        }
    };
    auto z_store = [&]() {
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = u * S2D_THREADS + tid;
            if (e < ZH * ZW) {
                zmax[e] = vz[u];                                                              // -inf outside the image
                zmin[e] = (vz[u] == 0.f) ? 999.f : ((vz[u] == -INFINITY) ? INFINITY : vz[u]);   // where(z == 0, 999, z)
            }
        }
    };

    // ---- P0: weights -> LDS in A-operand order -------------------------------------------------
    if (!p.pyramid) {
        for (int e = tid; e < S2D_W3; e += S2D_THREADS) {
            const int f = e & 7, t = (e >> 3) % 9, ch = e / 72;
            w3[e] = (f < p.nf && ch < nch) ? p.wconv[((long long)f * nch + ch) * 9 + t] : 0.f;
        }
        for (int e = tid; e < S2D_W1; e += S2D_THREADS) {
            const int f = e & 7, q = (e >> 3) & 7, i = e >> 6;
            const int cin = (i == 0) ? npool : p.nf;
            w1[e] = (i < p.nconv && f < p.nf && q < cin) ? p.wpool[i][f * cin + q] : 0.f;
        }
    }
    // ---- P1 (first tile): stage the depth tile (+halo) ------------------------------------------------------
    int tile_id = blockIdx.x;
    Tile tl = decode(tile_id);
    if (!(p.dbg & 16)) { z_load(tl); z_store(); }
    __syncthreads();

  for (;;) {   // ---- tile loop ----
    const int n = tl.n, oy0 = tl.oy0, ox0 = tl.ox0;
    const float* xz = p.x + (long long)n * p.x_bstride;  // channel 0 = sparse depth
    // the feature region (+1 halo, padded to 4-pixel groups) lies inside the image: no zero padding to apply
    const bool interior = oy0 >= 1 && oy0 + S2D_TH < p.H && ox0 >= 1 && ox0 - 1 + S2D_FWP <= p.W;   // block-uniform
    const int next_id = tile_id + (int)gridDim.x;
    const bool more = next_id < p.nblocks;

    // ---- P2: vertical pass -> V[pool][feature row][z column] ------------------------------------------
    if (p.dbg & 1) {
    } else if constexpr (CFG::is_static) {
        constexpr int NG = S2D_FH / S2D_GS;
        for (int t = tid; t < ZW * NG; t += S2D_THREADS) {
            const int g = t / ZW, c = t - g * ZW;
            auto sweep = [&](auto is_min_c, const float* zsrc) {
                constexpr bool IS_MIN = decltype(is_min_c)::value != 0;
                constexpr int RS = IS_MIN ? CFG::RMIN : CFG::RMAX;
                if constexpr (RS > 0) {
                    float m[S2D_GS + 2 * RS];
                    const float* s = zsrc + (g * S2D_GS + (CFG::RR - RS)) * ZW + c;
#pragma unroll
                    for (int i = 0; i < S2D_GS + 2 * RS; ++i) m[i] = s[i * ZW];
#pragma unroll
                    for (int j = 0; j < S2D_GS; ++j) {
                        float a = m[j + RS];
                        s2d_for<1, RS + 1>([&](auto dc) {
                            constexpr int d = decltype(dc)::value;
                            a = IS_MIN ? fminf(a, fminf(m[j + RS - d], m[j + RS + d]))
                                       : fmaxf(a, fmaxf(m[j + RS - d], m[j + RS + d]));
                            s2d_for<0, CFG::NP>([&](auto pic) {
                                constexpr int pi = decltype(pic)::value;
                                if constexpr ((pi < CFG::NMINP) == IS_MIN && CFG::radius(pi) == d)
                                    vbuf[pi * VPLANE + (g * S2D_GS + j) * VP + c] = a;
                            });
                        });
                    }
                }
            };
            sweep(IntC<1>{}, zmin);
            sweep(IntC<0>{}, zmax);
        }
    } else {
        for (int t = tid; t < ZW * S2D_FH; t += S2D_THREADS) {
            const int fr = t / ZW, c = t - fr * ZW;
            const float* smin = zmin + (fr + R) * ZW + c;
            const float* smax = zmax + (fr + R) * ZW + c;
            float a = smin[0], b = smax[0];
            for (int d = 1; d <= R; ++d) {
                if (d <= p.Rmin) a = fminf(a, fminf(smin[-d * ZW], smin[d * ZW]));
                if (d <= p.Rmax) b = fmaxf(b, fmaxf(smax[-d * ZW], smax[d * ZW]));
                for (int pi = 0; pi < npool; ++pi)
                    if ((p.ksize[pi] >> 1) == d) vbuf[pi * VPLANE + fr * VP + c] = pi < nmin ? a : b;
            }
        }
    }
    __syncthreads();
    Tile nxt = tl;
    if (more) {   // next tile's depth values -> registers, in flight during P3 .. P5 (zmin / zmax are dead from here on)
        nxt = decode(next_id);
        if constexpr (CFG::is_static) {   // (the run-time pool path has no registers to spare: it loads at the loop tail)
            if (!(p.dbg & 16)) z_load(nxt);
        }
    }

    // ---- P3: horizontal pass: pooled values of 4 consecutive feature pixels per item -> registers ------
    // item = (feature row fr, group q): feature columns 4q .. 4q+3 (columns >= FW are padding, never consumed)
    constexpr int NITEM = S2D_FH * S2D_NQ, NROUND = (NITEM + S2D_THREADS - 1) / S2D_THREADS;
    float pooled[NROUND][4][S2D_MAXPOOL];
    float xin[NROUND][S2D_MAXIN][4];
#pragma unroll
    for (int rd = 0; rd < NROUND; ++rd) {
        const int t = rd * S2D_THREADS + tid;
        const int fr = t / S2D_NQ, q = t - fr * S2D_NQ;
        const bool live = t < NITEM;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pi = 0; pi < S2D_MAXPOOL; ++pi) pooled[rd][j][pi] = 0.f;
        // the raw x channels of these pixels (zero outside the image): issued now, consumed after the 1x1 chain
#pragma unroll
        for (int ci = 0; ci < S2D_MAXIN; ++ci)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int Y = oy0 - 1 + fr, X = ox0 - 1 + 4 * q + j;
                const bool inb = live && ci < p.inC && !p.pyramid && (interior || (Y >= 0 && Y < p.H && X >= 0 && X < p.W));
                xin[rd][ci][j] = inb ? xz[(long long)ci * HW + (long long)Y * p.W + X] : 0.f;
            }
        if (!live || (p.dbg & 2)) continue;
        if constexpr (CFG::is_static) {
            s2d_for<0, CFG::NP>([&](auto pic) {
                constexpr int pi = decltype(pic)::value;
                constexpr int r = CFG::radius(pi);
                constexpr bool IS_MIN = pi < CFG::NMINP;
                constexpr int OFF = CFG::RR - r;              // z column of window element o of pixel j: 4q + j + OFF + o
                constexpr int A0 = OFF & ~3, SH = OFF & 3;    // 16-byte aligned start, shift inside the first word
                constexpr int NV = 2 * r + 4, NB = (SH + NV + 3) / 4;
                static_assert(r >= 2, "the blocked window code needs pool sizes >= 5");
                f32x4 w[NB];
                const float* s = vbuf + pi * VPLANE + fr * VP + 4 * q + A0;
#pragma unroll
                for (int m = 0; m < NB; ++m) w[m] = *reinterpret_cast<const f32x4*>(s + 4 * m);
                auto v = [&](int o) { return w[(SH + o) >> 2][(SH + o) & 3]; };
                auto mn = [&](float a, float b) { return IS_MIN ? fminf(a, b) : fmaxf(a, b); };
                // windows v[j .. j + 2r], j = 0..3: shared core v[3 .. 2r] + prefix / suffix
                float core = v(3);
#pragma unroll
                for (int o = 4; o <= 2 * r; ++o) core = mn(core, v(o));
                const float l1 = mn(v(1), v(2)), l0 = mn(v(0), l1);
                const float h2 = mn(v(2 * r + 1), v(2 * r + 2)), h3 = mn(h2, v(2 * r + 3));
                float o0 = mn(core, l0), o1 = mn(core, mn(l1, v(2 * r + 1))), o2 = mn(core, mn(v(2), h2)), o3 = mn(core, h3);
                if (IS_MIN) {   // where(pool == 999, 0, pool)
                    o0 = (o0 == 999.f) ? 0.f : o0; o1 = (o1 == 999.f) ? 0.f : o1;
                    o2 = (o2 == 999.f) ? 0.f : o2; o3 = (o3 == 999.f) ? 0.f : o3;
                }
                pooled[rd][0][pi] = o0; pooled[rd][1][pi] = o1; pooled[rd][2][pi] = o2; pooled[rd][3][pi] = o3;
            });
        } else {
#pragma unroll
            for (int pi = 0; pi < S2D_MAXPOOL; ++pi) {
                if (pi >= npool) continue;
                const int r = p.ksize[pi] >> 1;
                const bool is_min = pi < nmin;
                const float* s = vbuf + pi * VPLANE + fr * VP + 4 * q + R - r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = s[j];
                    for (int o = 1; o <= 2 * r; ++o) a = is_min ? fminf(a, s[j + o]) : fmaxf(a, s[j + o]);
                    pooled[rd][j][pi] = (is_min && a == 999.f) ? 0.f : a;
                }
            }
        }
    }
    if (p.pyramid) {
#pragma unroll
        for (int rd = 0; rd < NROUND; ++rd) {
            const int t = rd * S2D_THREADS + tid;
            const int fr = t / S2D_NQ, q = t - fr * S2D_NQ;
            if (t >= NITEM || fr < 1 || fr > S2D_TH) continue;
            const int Y = oy0 - 1 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int fx = 4 * q + j, X = ox0 - 1 + fx;
                if (fx < 1 || fx > S2D_TW || Y >= p.H || X >= p.W) continue;
                float* py = p.pyramid + ((long long)n * npool) * HW + (long long)Y * p.W + X;
#pragma unroll
                for (int pi = 0; pi < S2D_MAXPOOL; ++pi)
                    if (pi < npool) py[(long long)pi * HW] = pooled[rd][j][pi];
            }
        }
    }
    __syncthreads();  // every V read is done: `feat` overlays V from here on

    // ---- P4: 1x1 conv chain on the matrix cores (lane = pixel), features + raw x channels -> LDS --------
    if (!p.pyramid) {
#pragma unroll
        for (int rd = 0; rd < NROUND; ++rd) {
            const int t = rd * S2D_THREADS + tid;
            const int fr = t / S2D_NQ, q = t - fr * S2D_NQ;
            if (rd * S2D_THREADS + wave * 64 >= NITEM) continue;   // whole wave idle (wave-uniform: no MFMA under a partial EXEC)
            f32x4 h[4][2];                                         // [pixel][filter half]
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j][0] = h[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < S2D_MAXCONV; ++i) {
                if (i < p.nconv && !(p.dbg & 4)) {
                    float a1[8][2];   // A operands of this layer: [input][filter half], this lane's filter = lane & 3
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        a1[qq][0] = w1[((i * 8 + qq) * 2 + 0) * 4 + (lane & 3)];
                        a1[qq][1] = w1[((i * 8 + qq) * 2 + 1) * 4 + (lane & 3)];
                    }
                    f32x4 g[4][2];
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j][0] = g[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {   // zero weights beyond the real fan-in
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float b = (i == 0) ? pooled[rd][j][qq] : h[j][qq >> 2][qq & 3];
                            g[j][0] = mfma4(a1[qq][0], b, g[j][0]);
                            g[j][1] = mfma4(a1[qq][1], b, g[j][1]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                            for (int f = 0; f < 4; ++f) h[j][hh][f] = s2d_lrelu(g[j][hh][f], slope);
                }
            }
            if (t < NITEM) {
                float* fo = feat + fr * S2D_FWP + 4 * q;
                if (!interior) {   // zero padding of the feature map outside the image (block-uniform, border tiles only)
                    const int Y = oy0 - 1 + fr;
                    const bool rowin = Y >= 0 && Y < p.H;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int X = ox0 - 1 + 4 * q + j;
                        if (!(rowin && X >= 0 && X < p.W)) h[j][0] = h[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int f = 0; f < S2D_MAXF; ++f)
                    if (f < p.nf)
                        *reinterpret_cast<f32x4*>(fo + f * (S2D_FH * S2D_FWP)) =
                            (f32x4){h[0][f >> 2][f & 3], h[1][f >> 2][f & 3], h[2][f >> 2][f & 3], h[3][f >> 2][f & 3]};
#pragma unroll
                for (int ci = 0; ci < S2D_MAXIN; ++ci)
                    if (ci < p.inC)
                        *reinterpret_cast<f32x4*>(fo + (p.nf + ci) * (S2D_FH * S2D_FWP)) =
                            (f32x4){xin[rd][ci][0], xin[rd][ci][1], xin[rd][ci][2], xin[rd][ci][3]};
            }
        }
    }
    __syncthreads();

    // ---- P5: 3x3 conv over [features | x] on the matrix cores + LeakyReLU.  Wave w: rows 4w .. 4w+3 of the
    //      tile; lane = (row l >> 4, 4 consecutive columns 4 (l & 15) ..): the six feature values a row of the
    //      window needs are one 16-byte + one 8-byte LDS read, and every filter leaves as one 16-byte store.
    //      acc[pixel][half][j] = filter 4 half + j. ----------------------------------------------------------
    if (!p.pyramid) {
        constexpr int RW = S2D_TH / 4;     // rows per wave
        static_assert(RW == 4, "lane mapping: 16 lanes x 4 columns per row, 4 rows per wave");
        const int lr = lane >> 4, cg = lane & 15;
        f32x4 acc[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* fbase = feat + (wave * RW + lr) * S2D_FWP + 4 * cg;
        for (int ch = 0; ch < ((p.dbg & 8) ? 0 : nch); ++ch) {
            float a3[9][2], b[3][6];
            const float* wsrc = w3 + ch * 72 + (lane & 3);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) { a3[tp][0] = wsrc[tp * 8]; a3[tp][1] = wsrc[tp * 8 + 4]; }
            const float* fr = fbase + ch * (S2D_FH * S2D_FWP);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(fr + ky * S2D_FWP);
                const f32x2 hi = *reinterpret_cast<const f32x2*>(fr + ky * S2D_FWP + 4);
                b[ky][0] = lo[0]; b[ky][1] = lo[1]; b[ky][2] = lo[2]; b[ky][3] = lo[3]; b[ky][4] = hi[0]; b[ky][5] = hi[1];
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j][0] = mfma4(a3[ky * 3 + kx][0], b[ky][j + kx], acc[j][0]);
                        acc[j][1] = mfma4(a3[ky * 3 + kx][1], b[ky][j + kx], acc[j][1]);
                    }
        }
        const int X = ox0 + 4 * cg, Y = oy0 + wave * RW + lr;
        if (X < p.W && Y < p.H) {
            float* o = p.out + ((long long)n * p.nf) * HW + (long long)Y * p.W + X;
            const bool vec = (X + 3 < p.W) && ((p.W & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
#pragma unroll
            for (int f = 0; f < S2D_MAXF; ++f) {
                if (f >= p.nf) continue;
                const f32x4 v = (f32x4){s2d_lrelu(acc[0][f >> 2][f & 3], slope), s2d_lrelu(acc[1][f >> 2][f & 3], slope),
                                        s2d_lrelu(acc[2][f >> 2][f & 3], slope), s2d_lrelu(acc[3][f >> 2][f & 3], slope)};
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
    if (!more) break;
    tile_id = next_id;
    tl = nxt;
    if constexpr (!CFG::is_static) {
        if (!(p.dbg & 16)) z_load(tl);
    }
    if (!(p.dbg & 16)) z_store();   // the prefetched depth tile -> LDS
    __syncthreads();                // ... and every read of this tile's features is done (V is rewritten next)
  }   // tile loop
}


// ------------------------------------------------------------------------------------------------------------------
// The same layer with its convolutions on the 16-bit matrix core (round 6).  The fp32 form above pays for its 904
// multiply-adds per pixel on the ONE datapath the vector ALU and the fp32-input MFMAs share (0.28 of that pipe, 0.12 of
// HBM); here both convolutions run as v_mfma_f32_16x16x32_f16 over two-term fp16 splits (the arithmetic of
// csrc/conv_split.hip: a 2^k = h1 + 2^-11 h2, w 2^e = w1 + w2, three products), beside the vector ALU, which keeps the
// pool passes, the LeakyReLUs and the splitting.  For KBNet's own shape -- 2 input channels, a 1x1 chain of three layers,
// 8 filters, one of the compiled pool presets; anything else takes the fp32 kernel.  P1-P3 (depth tile, vertical and
// horizontal min / max passes: compare / select only, bit-exact pyramid) are the fp32 kernel's.  Then
//   P3b the pooled values of a pixel (8 channels) -> one 16-byte granule per term, G [term][pixel][8 fp16] (overlays V);
//       the raw channels -> X2 [term][feature row][pixel pair] = (z, v) of the four columns the pair's windows span
//       (overlays the depth tile).  Window: the tile's max(|z|, |v|) (the pooled values are minima / maxima of z).
//   P4  the 1x1 chain, a wave's 32-pixel blocks layer by layer, IN REGISTERS: D rows 0-7 = the filters at pixel
//       32 b + n (B k-groups 0, 1 = its h1, h2), rows 8-15 = the same filters at pixel 32 b + 16 + n (k-groups 2, 3);
//       two MFMAs per block and layer: [w1 | w1 2^-11] . [h1 ; h2] + [w2 | 0] . [h1 ; h2].  A lane ends up with four
//       channels of a pixel; scale, LeakyReLU, split, and one v_permlane16_swap per dword hands lanes kq = 0 / 2 the
//       whole h1 granule and lanes kq = 1 / 3 the whole h2 granule: the next layer's B operand, no LDS round trip.
//       Windows are MEASURED, never chained bounds: between the layers the maximum over the wave's own blocks (its
//       operands never leave the wave), for the chain's output the maximum over the tile (one LDS slot per wave + the
//       barrier that publishes the features anyway).
//   P5  the 3x3 conv over cat[features, z, v] for pixel PAIRS (8 filters fill half of a 16-row MFMA): rows 0-7 the
//       filters at pixel 2p, rows 8-15 at pixel 2p + 1, over the 3 x 4 window the two share: K-step = window row,
//       k-group = window column; the raw channels take a fourth K-step (own accumulators: their window is the input's).
//       LeakyReLU, NCHW stores (a wave's store covers 128 contiguous bytes of two filter planes).
// The weights are split and put in MFMA operand order by every (persistent) workgroup at its start: 274 weights, no
// packed blob, no state -- the ABI call is unchanged.
typedef unsigned spu2x __attribute__((ext_vector_type(2)));
typedef unsigned spu4x __attribute__((ext_vector_type(4)));
constexpr int SM_NB4 = (S2D_FH * S2D_FWP + 31) / 32;                 // 32-pixel blocks of the chain (39)
constexpr int SM_GPART = SM_NB4 * 32 * 16;                           // bytes of one term of G
constexpr int SM_NPAIR = S2D_TW / 2;                                 // pixel pairs per output row (32)
constexpr int SM_X2PART = S2D_FH * SM_NPAIR * 16;                    // bytes of one term of X2
constexpr int SM_NB5 = S2D_TH * SM_NPAIR / 16;                       // 16-pair blocks of the 3x3 conv (32)
constexpr int SM_WCHAIN = 3 * 2 * 64 * 8, SM_WCONV = 4 * 2 * 64 * 8; // halves: [layer][operand][lane][8], [K-step][term][lane][8]
constexpr int SM_TAB = 64;                                           // floats: 2^-e of filter f of layer i at 8 i + f (i = 3: the 3x3 conv)
constexpr int SM_WBYTES = SM_TAB * 4 + (SM_WCHAIN + SM_WCONV) * 2;
constexpr int SM_RED = 16;                                           // floats of reduction slots behind the weights
static_assert(S2D_NQ * 4 == S2D_FWP && S2D_TH * SM_NPAIR % 16 == 0, "tile geometry");

template <typename CFG>
struct S2DMfmaGeom {
    static constexpr int R = CFG::RR, ZW = s2d_zw(R), ZH = s2d_zh(R), VP = s2d_vp(R), VPLANE = S2D_FH * VP;
    static constexpr int ZBYTES = (2 * ZH * ZW * 4 + 15) / 16 * 16, VBYTES = CFG::NP * VPLANE * 4;
    static constexpr int GBYTES = VBYTES > 2 * SM_GPART ? VBYTES : 2 * SM_GPART;
    static constexpr int OFF_RED = SM_WBYTES, OFF_Z = OFF_RED + SM_RED * 4, OFF_G = OFF_Z + ZBYTES;
    static constexpr int LDS = OFF_G + GBYTES;
    static_assert(2 * SM_X2PART <= ZBYTES, "X2 overlays the depth tile");
    static_assert(OFF_Z % 16 == 0 && OFF_G % 16 == 0, "16-byte granules");
};

template <typename CFG>
__global__ __launch_bounds__(S2D_THREADS, 2) void s2d_mfma_kernel(const S2DParams p) {
    using GM = S2DMfmaGeom<CFG>;
    constexpr int R = GM::R, ZW = GM::ZW, ZH = GM::ZH, VP = GM::VP, VPLANE = GM::VPLANE, NP = CFG::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    float* const tab = reinterpret_cast<float*>(smem_b);
    _Float16* const wchain = reinterpret_cast<_Float16*>(smem_b + SM_TAB * 4);
    _Float16* const wconv = wchain + SM_WCHAIN;
    float* const red = reinterpret_cast<float*>(smem_b + GM::OFF_RED);   // [0..3] max |z| of the staged tile, [4..7] max |v|, [8..11] max |feature|
    float* const zmin = reinterpret_cast<float*>(smem_b + GM::OFF_Z);
    float* const zmax = zmin + ZH * ZW;
    unsigned char* const X2 = smem_b + GM::OFF_Z;
    float* const vbuf = reinterpret_cast<float*>(smem_b + GM::OFF_G);
    unsigned char* const G = smem_b + GM::OFF_G;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const long long HW = (long long)p.H * p.W;
    const float slope = p.slope;

    struct Tile { int n, oy0, ox0; };
    auto decode = [&](int t) {
        int bid = xcd_remap(t, p.nblocks);
        const int tx = bid % p.tilesX;
        bid /= p.tilesX;
        const int ty = bid % p.tilesY;
        return Tile{bid / p.tilesY, ty * S2D_TH, tx * S2D_TW};
    };
    constexpr int MAXE = (ZH * ZW + S2D_THREADS - 1) / S2D_THREADS;
    float vz[MAXE];
    auto z_load = [&](const Tile& tl) {
        const float* src = p.x + (long long)tl.n * p.x_bstride;
        const int Yt = tl.oy0 - 1 - R, Xl = tl.ox0 - 1 - R;
        const bool inside = Yt >= 0 && Yt + ZH <= p.H && Xl >= 0 && Xl + ZW <= p.W;   // block-uniform
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = u * S2D_THREADS + tid;
            const int r = e / ZW, c = e - r * ZW;
            const int Y = Yt + r, X = Xl + c;
            const bool ok = e < ZH * ZW && (inside || (Y >= 0 && Y < p.H && X >= 0 && X < p.W));
            vz[u] = ok ? src[(long long)Y * p.W + X] : -INFINITY;
        }
    };
    auto z_store = [&]() {   // -> zmin / zmax, and max |z| over the tile (+halo) into red[wave]
        float tm = 0.f;
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = u * S2D_THREADS + tid;
            if (e < ZH * ZW) {
                zmax[e] = vz[u];
                zmin[e] = (vz[u] == 0.f) ? 999.f : ((vz[u] == -INFINITY) ? INFINITY : vz[u]);
                if (vz[u] != -INFINITY) tm = fmaxf(tm, fabsf(vz[u]));
            }
        }
        const unsigned b = wave_max_bits(tm);
        if (lane == 0) red[wave] = __uint_as_float(b);
    };

    // ---- P0: weights -> split fp16 MFMA operands in LDS (the layouts of csrc/front.hip's s2d_stage_pack_kernel) ----
    float* const escale = tab + 32;   // 2^e of filter f of layer i at 32 + 8 i + f
    if (tid < 32) {
        const int layer = tid >> 3, f = tid & 7;
        const int per = layer == 0 ? NP : (layer == 3 ? 90 : 8);
        const float* w = (layer == 3 ? p.wconv : p.wpool[layer]) + (long long)f * per;
        float m = 0.f;
        for (int i = 0; i < per; ++i) m = fmaxf(m, fabsf(w[i]));
        int ex = FR_WEXP;
        if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &ex);
        int e = FR_WEXP - ex;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        tab[tid] = ldexpf(1.f, -e);
        escale[tid] = ldexpf(1.f, e);
    }
    __syncthreads();
    for (int e = tid; e < SM_WCHAIN; e += S2D_THREADS) {
        const int j = e & 7, ln = (e >> 3) & 63, op = (e >> 9) & 1, layer = e >> 10;
        const int m = ln & 15, q = ln >> 4;
        const int f = m & 7, second = m >> 3;       // rows 8-15: the same filters at the block's second pixel (k-groups 2, 3)
        const int cin = layer == 0 ? NP : 8;
        float v = 0.f;
        if (j < cin && (q >> 1) == second) {
            const float ws = p.wpool[layer][f * cin + j] * escale[layer * 8 + f];
            const _Float16 w1 = (_Float16)ws;
            if (op == 0) v = (q & 1) ? (float)w1 * 0.00048828125f : (float)w1;   // [w1 | w1 2^-11] . [h1 ; h2]
            else v = (q & 1) ? 0.f : ws - (float)w1;                              // [w2 | 0] . [h1 ; h2]
        }
        wchain[e] = (_Float16)v;
    }
    for (int e = tid; e < SM_WCONV; e += S2D_THREADS) {
        const int j = e & 7, ln = (e >> 3) & 63, term = (e >> 9) & 1, s = e >> 10;
        const int m = ln & 15, q = ln >> 4;
        const int f = m & 7, second = m >> 3;
        float ws = 0.f;
        bool live = false;
        if (s < 3) {              // features: window row s, window column q
            const int kx = q - second;
            if (kx >= 0 && kx < 3) { live = true; ws = p.wconv[((f * 10 + j) * 3 + s) * 3 + kx]; }
        } else if (q < 3) {       // raw channels: window row q, window column j >> 1, channel j & 1
            const int kx = (j >> 1) - second;
            if (kx >= 0 && kx < 3) { live = true; ws = p.wconv[((f * 10 + 8 + (j & 1)) * 3 + q) * 3 + kx]; }
        }
        wconv[e] = live ? fr_term(ws * escale[24 + f], term) : (_Float16)0.f;
    }
    int tile_id = blockIdx.x;
    Tile tl = decode(tile_id);
    if (!(p.dbg & 16)) z_load(tl);
    z_store();
    __syncthreads();

  for (;;) {   // ---- tile loop ----
    const int n = tl.n, oy0 = tl.oy0, ox0 = tl.ox0;
    const float* xz = p.x + (long long)n * p.x_bstride;
    const bool interior = oy0 >= 1 && oy0 + S2D_TH < p.H && ox0 >= 1 && ox0 - 1 + S2D_FWP + 2 <= p.W;   // block-uniform: every feature pixel (and the two columns a last pair item reads past its own) inside the image
    const int next_id = tile_id + (int)gridDim.x;
    const bool more = next_id < p.nblocks;
    const float tmz = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));

    // ---- P2: vertical pass (csrc/s2d.hip P2) ----
    if (!(p.dbg & 1)) {
        constexpr int NG = S2D_FH / S2D_GS;
        for (int t = tid; t < ZW * NG; t += S2D_THREADS) {
            const int g = t / ZW, c = t - g * ZW;
            auto sweep = [&](auto is_min_c, const float* zsrc) {
                constexpr bool IS_MIN = decltype(is_min_c)::value != 0;
                constexpr int RS = IS_MIN ? CFG::RMIN : CFG::RMAX;
                if constexpr (RS > 0) {
                    float m[S2D_GS + 2 * RS];
                    const float* s = zsrc + (g * S2D_GS + (CFG::RR - RS)) * ZW + c;
#pragma unroll
                    for (int i = 0; i < S2D_GS + 2 * RS; ++i) m[i] = s[i * ZW];
#pragma unroll
                    for (int j = 0; j < S2D_GS; ++j) {
                        float a = m[j + RS];
                        s2d_for<1, RS + 1>([&](auto dc) {
                            constexpr int d = decltype(dc)::value;
                            a = IS_MIN ? fminf(a, fminf(m[j + RS - d], m[j + RS + d]))
                                       : fmaxf(a, fmaxf(m[j + RS - d], m[j + RS + d]));
                            s2d_for<0, CFG::NP>([&](auto pic) {
                                constexpr int pi = decltype(pic)::value;
                                if constexpr ((pi < CFG::NMINP) == IS_MIN && CFG::radius(pi) == d)
                                    vbuf[pi * VPLANE + (g * S2D_GS + j) * VP + c] = a;
                            });
                        });
                    }
                }
            };
            sweep(IntC<1>{}, zmin);
            sweep(IntC<0>{}, zmax);
        }
    }
    __syncthreads();
    Tile nxt = tl;
    if (more) {   // next tile's depth values -> registers, in flight during P3 .. P5
        nxt = decode(next_id);
        if (!(p.dbg & 16)) z_load(nxt);
    }

    // ---- P3: horizontal pass (csrc/s2d.hip P3): pooled values of 4 consecutive feature pixels per item -> registers;
    //      the raw channels of the item's columns and of the two behind them (the second pair of the item spans 4q + 2 .. 4q + 5)
    constexpr int NITEM = S2D_FH * S2D_NQ, NROUND = (NITEM + S2D_THREADS - 1) / S2D_THREADS;
    float pooled[NROUND][4][8];
    float xr[NROUND][2][6];
    float tmv = 0.f;
#pragma unroll
    for (int rd = 0; rd < NROUND; ++rd) {
        const int t = rd * S2D_THREADS + tid;
        const int fr = t / S2D_NQ, q = t - fr * S2D_NQ;
        const bool live = t < NITEM;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pi = 0; pi < 8; ++pi) pooled[rd][j][pi] = 0.f;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int Y = oy0 - 1 + fr, fx = 4 * q + j, X = ox0 - 1 + fx;
                const bool inb = live && fx < S2D_FW && (interior || (Y >= 0 && Y < p.H && X >= 0 && X < p.W));
                const float v = inb ? xz[(long long)ci * HW + (long long)Y * p.W + X] : 0.f;
                xr[rd][ci][j] = v;
                tmv = fmaxf(tmv, fabsf(v));
            }
        if (!live || (p.dbg & 2)) continue;
        s2d_for<0, CFG::NP>([&](auto pic) {
            constexpr int pi = decltype(pic)::value;
            constexpr int r = CFG::radius(pi);
            constexpr bool IS_MIN = pi < CFG::NMINP;
            constexpr int OFF = CFG::RR - r;
            constexpr int A0 = OFF & ~3, SH = OFF & 3;
            constexpr int NV = 2 * r + 4, NB = (SH + NV + 3) / 4;
            static_assert(r >= 2, "the blocked window code needs pool sizes >= 5");
            f32x4 w[NB];
            const float* s = vbuf + pi * VPLANE + fr * VP + 4 * q + A0;
#pragma unroll
            for (int m = 0; m < NB; ++m) w[m] = *reinterpret_cast<const f32x4*>(s + 4 * m);
            auto v = [&](int o) { return w[(SH + o) >> 2][(SH + o) & 3]; };
            auto mn = [&](float a, float b) { return IS_MIN ? fminf(a, b) : fmaxf(a, b); };
            float core = v(3);
#pragma unroll
            for (int o = 4; o <= 2 * r; ++o) core = mn(core, v(o));
            const float l1 = mn(v(1), v(2)), l0 = mn(v(0), l1);
            const float h2 = mn(v(2 * r + 1), v(2 * r + 2)), h3 = mn(h2, v(2 * r + 3));
            float o0 = mn(core, l0), o1 = mn(core, mn(l1, v(2 * r + 1))), o2 = mn(core, mn(v(2), h2)), o3 = mn(core, h3);
            if (IS_MIN) {   // where(pool == 999, 0, pool)
                o0 = (o0 == 999.f) ? 0.f : o0; o1 = (o1 == 999.f) ? 0.f : o1;
                o2 = (o2 == 999.f) ? 0.f : o2; o3 = (o3 == 999.f) ? 0.f : o3;
            }
            pooled[rd][0][pi] = o0; pooled[rd][1][pi] = o1; pooled[rd][2][pi] = o2; pooled[rd][3][pi] = o3;
        });
    }
    {
        const unsigned b = wave_max_bits(tmv);
        if (lane == 0) red[4 + wave] = __uint_as_float(b);
    }
    __syncthreads();  // every V and depth-tile read is done: G overlays V, X2 the depth tile

    // ---- P3b: split granules of the chain's input (G) and of the raw channels (X2); window: the tile's max(|z|, |v|) ----
    float pre0, un0;
    {
        const float b0 = fmaxf(tmz, fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
        fr_scales(__builtin_amdgcn_readfirstlane(__float_as_uint(b0)), pre0, un0);
    }
#pragma unroll
    for (int rd = 0; rd < NROUND; ++rd) {
        const int t = rd * S2D_THREADS + tid;
        const int fr = t / S2D_NQ, q = t - fr * S2D_NQ;
        if (t >= NITEM || (p.dbg & 64)) continue;
        const int Y = oy0 - 1 + fr;
        const bool rowin = Y >= 0 && Y < p.H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int fx = 4 * q + j, X = ox0 - 1 + fx;
            // the two columns behind the 66 feature columns pad the row to whole 4-pixel items: their windows reach past the staged
            // tile (stale LDS); they and the pixels outside the image are zero -- inputs of the chain, never read by the 3x3 conv
            const bool in = fx < S2D_FW && (interior || (rowin && X >= 0 && X < p.W));
            ff4 a = {pooled[rd][j][0], pooled[rd][j][1], pooled[rd][j][2], pooled[rd][j][3]};
            ff4 b = {pooled[rd][j][4], pooled[rd][j][5], pooled[rd][j][6], pooled[rd][j][7]};
            if (!in) { a = (ff4){0.f, 0.f, 0.f, 0.f}; b = a; }
            fh4 a1, a2, b1, b2;
            fr_split4(a * pre0, a1, a2);
            fr_split4(b * pre0, b1, b2);
            const int e = (fr * S2D_FWP + fx) * 16;
            *reinterpret_cast<fh8*>(G + e) = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<fh8*>(G + SM_GPART + e) = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        if (2 * q + 1 < SM_NPAIR) {   // pairs 2q, 2q + 1 of this row: (z, v) of columns 4q .. 4q + 3 and 4q + 2 .. 4q + 5
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const ff4 a = {xr[rd][0][2 * h], xr[rd][1][2 * h], xr[rd][0][2 * h + 1], xr[rd][1][2 * h + 1]};
                const ff4 b = {xr[rd][0][2 * h + 2], xr[rd][1][2 * h + 2], xr[rd][0][2 * h + 3], xr[rd][1][2 * h + 3]};
                fh4 a1, a2, b1, b2;
                fr_split4(a * pre0, a1, a2);
                fr_split4(b * pre0, b1, b2);
                const int e = (fr * SM_NPAIR + 2 * q + h) * 16;
                *reinterpret_cast<fh8*>(X2 + e) = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                *reinterpret_cast<fh8*>(X2 + SM_X2PART + e) = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    }
    if (tid < SM_NB4 * 32 - S2D_FH * S2D_FWP) {   // the slack behind the last feature pixel: finite values for the last block
        const int e = (S2D_FH * S2D_FWP + tid) * 16;
        const fh8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        *reinterpret_cast<fh8*>(G + e) = zero;
        *reinterpret_cast<fh8*>(G + SM_GPART + e) = zero;
    }
    __syncthreads();

    // ---- P4: the 1x1 chain in registers ----
    constexpr int NU = (SM_NB4 + 3) / 4;   // blocks per wave (10)
    const int term_off = (kq & 1) * SM_GPART;
    float preF, unF;
    {
        fh8 B[NU];
        ff4 v[NU];
        float un_prev = un0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const fh8 A0 = *reinterpret_cast<const fh8*>(wchain + ((i * 2 + 0) * 64 + lane) * 8);
            const fh8 A1 = *reinterpret_cast<const fh8*>(wchain + ((i * 2 + 1) * 64 + lane) * 8);
            const ff4 sc = *reinterpret_cast<const ff4*>(tab + 8 * i + 4 * (kq & 1)) * un_prev;
            float m = 0.f;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int blk = wave + 4 * u;
                if (blk < SM_NB4 && !(p.dbg & 4)) {   // wave-uniform
                    const int px = 32 * blk + 16 * (kq >> 1) + l15;
                    if (i == 0) B[u] = *reinterpret_cast<const fh8*>(G + term_off + px * 16);
                    ff4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, B[u], (ff4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, B[u], d, 0, 0, 0);
                    d *= sc;
                    ff4 t = {s2d_lrelu(d[0], slope), s2d_lrelu(d[1], slope), s2d_lrelu(d[2], slope), s2d_lrelu(d[3], slope)};
                    if (i == 2) {   // zero padding of the 3x3 conv's input: pixels outside the image (and the row padding)
                        const int pr = px / S2D_FWP, pc = px - pr * S2D_FWP;
                        const int Y = oy0 - 1 + pr, X = ox0 - 1 + pc;
                        if (!(interior || (Y >= 0 && Y < p.H && X >= 0 && X < p.W))) t = (ff4){0.f, 0.f, 0.f, 0.f};
                    }
                    v[u] = t;
                    m = sp_amax4f(m, t);
                } else {
                    v[u] = (ff4){0.f, 0.f, 0.f, 0.f};
                }
            }
            float pre, un;
            if (i < 2) {
                fr_scales(wave_max_bits(m), pre, un);          // this wave's blocks: their operands never leave the wave
            } else {
                const unsigned b = wave_max_bits(m);
                if (lane == 0) red[8 + wave] = __uint_as_float(b);
                __syncthreads();                               // (also: every wave is done reading G)
                const float bf = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));
                fr_scales(__builtin_amdgcn_readfirstlane(__float_as_uint(bf)), pre, un);
                preF = pre; unF = un;
            }
            un_prev = un;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int blk = wave + 4 * u;
                if (blk < SM_NB4) {
                    fh4 h1, h2;
                    fr_split4(v[u] * pre, h1, h2);
                    // lanes kq = 0 / 2 take the pixel's whole h1 granule (their channels 0-3 + the partner's 4-7), lanes kq = 1 / 3 its h2 granule
                    const spu2x H1 = __builtin_bit_cast(spu2x, h1), H2 = __builtin_bit_cast(spu2x, h2);
                    const auto r0 = __builtin_amdgcn_permlane16_swap(H1[0], H2[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(H1[1], H2[1], false, false);
                    const spu4x g4 = {r0[0], r1[0], r0[1], r1[1]};
                    if (i < 2) {
                        B[u] = __builtin_bit_cast(fh8, g4);
                    } else {
                        const int px = 32 * blk + 16 * (kq >> 1) + l15;
                        *reinterpret_cast<spu4x*>(G + term_off + px * 16) = g4;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- P5: 3x3 conv over [features | z, v] for pixel pairs -> out ----
    if (!(p.dbg & 8)) {
        fh8 aF[3][2], aR[2];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) aF[s][t] = *reinterpret_cast<const fh8*>(wconv + ((s * 2 + t) * 64 + lane) * 8);
#pragma unroll
        for (int t = 0; t < 2; ++t) aR[t] = *reinterpret_cast<const fh8*>(wconv + ((3 * 2 + t) * 64 + lane) * 8);
        const ff4 inv = *reinterpret_cast<const ff4*>(tab + 24 + 4 * (kq & 1));
        const ff4 scF = inv * unF, scR = inv * un0;
        const bool vec_ok = true;
        (void)vec_ok;
        float* const obase = p.out + ((long long)n * 8 + 4 * (kq & 1)) * HW;
#pragma unroll 2
        for (int u = 0; u < SM_NB5 / 4; ++u) {
            const int b = wave + 4 * u;
            const int y = b >> 1, pp = 16 * (b & 1) + l15;
            const unsigned char* fb = G + (y * S2D_FWP + 2 * pp + kq) * 16;
            const unsigned char* rb = X2 + ((y + (kq < 3 ? kq : 2)) * SM_NPAIR + pp) * 16;
            ff4 mF = {0.f, 0.f, 0.f, 0.f}, sF = mF, mR = mF, sR = mF;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const fh8 b1 = *reinterpret_cast<const fh8*>(fb + s * S2D_FWP * 16);
                const fh8 b2 = *reinterpret_cast<const fh8*>(fb + SM_GPART + s * S2D_FWP * 16);
                mF = __builtin_amdgcn_mfma_f32_16x16x32_f16(aF[s][0], b1, mF, 0, 0, 0);
                sF = __builtin_amdgcn_mfma_f32_16x16x32_f16(aF[s][1], b1, sF, 0, 0, 0);
                sF = __builtin_amdgcn_mfma_f32_16x16x32_f16(aF[s][0], b2, sF, 0, 0, 0);
            }
            {
                const fh8 b1 = *reinterpret_cast<const fh8*>(rb);
                const fh8 b2 = *reinterpret_cast<const fh8*>(rb + SM_X2PART);
                mR = __builtin_amdgcn_mfma_f32_16x16x32_f16(aR[0], b1, mR, 0, 0, 0);
                sR = __builtin_amdgcn_mfma_f32_16x16x32_f16(aR[1], b1, sR, 0, 0, 0);
                sR = __builtin_amdgcn_mfma_f32_16x16x32_f16(aR[0], b2, sR, 0, 0, 0);
            }
            const int X = ox0 + 2 * pp + (kq >> 1), Y = oy0 + y;
            if (X < p.W && Y < p.H && !(p.dbg & 128)) {
                float* o = obase + (long long)Y * p.W + X;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t = __builtin_fmaf(sF[r], 0.00048828125f, mF[r]) * scF[r] + __builtin_fmaf(sR[r], 0.00048828125f, mR[r]) * scR[r];
                    o[(long long)r * HW] = s2d_lrelu(t, slope);
                }
            }
        }
    }
    if (!more) break;
    tile_id = next_id;
    tl = nxt;
    __syncthreads();   // every read of this tile's features and raw granules is done (the depth tile is rewritten next)
    z_store();
    __syncthreads();
  }   // tile loop
}

static int s2d_launch(S2DParams& p, const int* min_pool_sizes, int n_min, const int* max_pool_sizes,
                      int n_max, hipStream_t stream) {
    if (n_min < 0 || n_max < 0 || n_min + n_max < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (n_min + n_max > S2D_MAXPOOL) return KBN_ERR_UNSUPPORTED;
    if ((n_min && !min_pool_sizes) || (n_max && !max_pool_sizes)) return KBN_ERR_INVALID_ARGUMENT;
    int R = 0;
    p.Rmin = 0; p.Rmax = 0;
    for (int i = 0; i < n_min + n_max; ++i) {
        int k = i < n_min ? min_pool_sizes[i] : max_pool_sizes[i - n_min];
        if (k < 3 || (k & 1) == 0) return KBN_ERR_INVALID_ARGUMENT;  // the caller drops sizes <= 1
        if (k > 2 * S2D_MAXR + 1) return KBN_ERR_UNSUPPORTED;
        p.ksize[i] = k;
        if (k / 2 > R) R = k / 2;
        if (i < n_min) { if (k / 2 > p.Rmin) p.Rmin = k / 2; } else { if (k / 2 > p.Rmax) p.Rmax = k / 2; }
    }
    for (int i = n_min + n_max; i < S2D_MAXPOOL; ++i) p.ksize[i] = 1;
    p.nmin = n_min;
    p.npool = n_min + n_max;
    p.R = R;
    p.tilesX = ceil_div(p.W, S2D_TW);
    p.tilesY = ceil_div(p.H, S2D_TH);
    const size_t z_floats = ((size_t)2 * s2d_zh(R) * s2d_zw(R) + 3) & ~(size_t)3;
    const size_t v_floats = (size_t)p.npool * S2D_FH * s2d_vp(R);
    const size_t feat_floats = (size_t)S2D_MAXCH * S2D_FH * S2D_FWP;
    const size_t lds = sizeof(float) * (S2D_WFLOATS + z_floats + (v_floats > feat_floats ? v_floats : feat_floats));
    if (lds > 160 * 1024) return KBN_ERR_UNSUPPORTED;
    const long long blocks = (long long)p.tilesX * p.tilesY * p.N;
    if (blocks > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.nblocks = (int)blocks;
    p.dbg = knob(KNOB_S2D_DEBUG);
    auto matches = [&](int nm, std::initializer_list<int> ks) {
        if (p.nmin != nm || p.npool != (int)ks.size()) return false;
        int i = 0;
        for (int k : ks)
            if (p.ksize[i++] != k) return false;
        return true;
    };
    auto launch = [&](auto kern, DeviceOnce& once) -> int {
        if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
        int cus = device_cu_count();
        if (cus < 1) cus = 256;
        const long long per_cu = (2 * lds <= 160 * 1024) ? 2 : 1;
        const long long grid = blocks < per_cu * cus ? blocks : per_cu * cus;   // persistent workgroups
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(S2D_THREADS), lds, stream, p);
        return KBN_OK;
    };
    static DeviceOnce set_kitti, set_void, set_voidtrain, set_dyn, setm_kitti, setm_void, setm_voidtrain;
    // KBNet's own S2D shape on a compiled pool preset: the convolutions on the 16-bit matrix core (s2d_mfma_kernel); opt-in, KBN_S2D_DEBUG & 256 (round 6: parity green, level with the fp32 form in time -- profiles/r06/v82_s2d_mfma_form_ablation.txt)
    const bool mfma = !p.pyramid && p.nf == 8 && p.nconv == 3 && p.inC == 2 && (p.dbg & 256);
    auto launch_mfma = [&](auto kern, DeviceOnce& once, size_t lds_m) -> int {
        if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
        int cus = device_cu_count();
        if (cus < 1) cus = 256;
        const long long per_cu = (2 * lds_m <= 160 * 1024) ? 2 : 1;
        const long long grid = blocks < per_cu * cus ? blocks : per_cu * cus;   // persistent workgroups
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(S2D_THREADS), lds_m, stream, p);
        return KBN_OK;
    };
    int rc;
    if (matches(5, {5, 7, 9, 11, 13, 15, 17}))
        rc = mfma ? launch_mfma(s2d_mfma_kernel<KittiPools>, setm_kitti, S2DMfmaGeom<KittiPools>::LDS) : launch(s2d_kernel<KittiPools>, set_kitti);
    else if (matches(2, {15, 17, 23, 27, 29}))
        rc = mfma ? launch_mfma(s2d_mfma_kernel<VoidPools>, setm_void, S2DMfmaGeom<VoidPools>::LDS) : launch(s2d_kernel<VoidPools>, set_void);
    else if (matches(3, {15, 17, 19, 23, 27}))
        rc = mfma ? launch_mfma(s2d_mfma_kernel<VoidTrainPools>, setm_voidtrain, S2DMfmaGeom<VoidTrainPools>::LDS) : launch(s2d_kernel<VoidTrainPools>, set_voidtrain);
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
    if (!(negative_slope >= 0.f && negative_slope <= 1.f)) return KBN_ERR_UNSUPPORTED;   // LeakyReLU / ReLU / identity
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
