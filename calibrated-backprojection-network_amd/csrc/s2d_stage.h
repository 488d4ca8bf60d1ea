// s2d_stage.h -- SparseToDensePool.forward (reference src/networks.py:2168-2196) evaluated ON CHIP for the 19 x 35 full-resolution
// pixels one tile of the encoder's depth front reads (csrc/front.hip: kb1_depth_front_kernel<CFG>), with its convolutions on
// the 16-bit matrix core.  The S2D tensor (8 channels at full resolution: 13.7 MB per KITTI frame written by kbn_s2d_forward and
// read back by kbn_kb1_depth_front_forward) then exists only as the split granules of conv0_depth's input tile in LDS.
//
//   P1  sparse depth z of the (21 + 2R) x (40 + 2R) pixels around the tile -> LDS twice, like csrc/s2d.hip: zmin (0 -> 999,
//       +inf outside the image) and zmax (-inf outside); the validity values v of the 21 x 40 feature pixels stay in
//       registers; max(|z|, |v|) over the tile places the fp16 window of everything derived from them
//   P2  vertical min / max pass, register blocked (3 rows per item), every pool size from one outward sweep -> V
//   P3  horizontal pass, register blocked (4 pixels per item): the pooled values, BIT-EXACT compare / select arithmetic,
//       999-sentinel semantics (s2d.hip's P3) -> registers -> (barrier) -> split granules G [term][pixel][8 channels]
//   P4  the 1x1 chain (7|5 -> 8 -> 8 -> 8, LeakyReLU each) on v_mfma_f32_16x16x32_f16, IN PLACE on G, 32-pixel blocks, a wave's
//       (up to four) blocks layer by layer: D[filter | filter][pixel] -- rows 0-7 the filters at pixel 32 b + n (B k-groups 0, 1 =
//       its h1, h2), rows 8-15 the same filters at pixel 32 b + 16 + n (k-groups 2, 3).  TWO instructions per layer carry the
//       three products of the two-term split: A0 = [w1 | w1 2^-11] . [h1 ; h2] and A1 = [w2 | 0] . [h1 ; h2] (8 channels per
//       k-group; s2d_stage_pack_kernel in csrc/front.hip writes exactly this operand layout).  A lane ends up with 4 consecutive
//       channels of a pixel = half a granule: scale, LeakyReLU, split, ds_write_b64.
//   P5  the 3x3 conv over cat[features, z, v] (10 -> 8): 8 filters fill half of a 16-row MFMA, so a block computes pixel
//       PAIRS -- rows 0-7 the filters at pixel 2p, rows 8-15 the same filters at pixel 2p + 1 -- over the 3 x 4 window the two
//       share: K-step = window row, k-group = window column (weights of the column a pixel does not read are zero); the two
//       raw channels take a fourth K-step of their own (granules X2 [window row][pair] = (z, v) of the four columns, own
//       accumulators: their window is the tile maximum's, the features' the chain's bound).  Result: LeakyReLU, zero
//       outside the image (conv0_depth's padding), split -> IN [term][19 x 35 pixels][8 channels], what phase B of the depth
//       front reads.
// Windows: every on-chip tensor takes its 2^k from a BOUND (max |input| x the widest filter's L1 norm, layer by layer), as
// csrc/front.hip does for conv0; the bounds sit a few binades above the data and 2^16 of slack costs nothing.
#pragma once

#include "front_common.h"
#include "s2d_pools.h"

namespace kbn {

// blob of kbn_s2d_depth_front_pack_weight: SF_TAB floats, then fp16 [chain: 3 layers][2 operands][64 lanes][8] [conv: 4 K-steps][2 terms][64 lanes][8]
//   tab[0..2]  L1max of the chain layers (max over filters of sum |w|), tab[3] of the 3x3 conv (all 90 weights)
//   tab[8 + 8 i + f]  2^-e of filter f of chain layer i (i = 0..2), tab[32 + f] of the 3x3 conv
constexpr int SF_TAB = 64, SF_CHAIN_HALVES = 3 * 2 * 64 * 8, SF_CONV_HALVES = 4 * 2 * 64 * 8;

template <typename CFG>
struct S2DStage {
    static constexpr int R = CFG::RR, NP = CFG::NP;
    static constexpr int FH = FR_R0H + 2, FW = FR_R0W + 2, NQ = (FW + 3) / 4, FWP = NQ * 4;   // feature region 21 x 37 (pitch 40)
    static constexpr int ZH = FH + 2 * R, ZW = FWP + 2 * R;                                      // staged depth tile (every feature column has its window)
    static constexpr int VP = (ZW + 3) / 4 * 4, VPLANE = FH * VP;
    static constexpr int GS = 3, NGRP = FH / GS;                                                  // vertical pass: rows per item
    static constexpr int NG = (FH * FWP + 31) / 32 * 32, GPART = NG * 16;                         // granules of G (whole 32-pixel blocks)
    static constexpr int NPAIR = (FR_R0W + 1) / 2, X2PART = FH * NPAIR * 16;                      // 18 pairs per row
    static constexpr int ZBYTES = (2 * ZH * ZW * 4 + 15) / 16 * 16, VBYTES = NP * VPLANE * 4;
    static constexpr int OFF_Z = 0, OFF_V = ZBYTES, OFF_G = OFF_V, OFF_X2 = 0;
    static constexpr int BYTES = ZBYTES + VBYTES;                                                 // LDS of the stage (dead once IN is written)
    static_assert(FH % GS == 0 && VP % 4 == 0, "tile geometry");
    static_assert(2 * GPART <= VBYTES, "G overlays V");
    static_assert(2 * X2PART <= ZBYTES, "X2 overlays the depth tile");
    static_assert(ZW * NGRP <= FR_THREADS && 2 * FH * NQ + (NG - FH * FWP) <= FR_THREADS, "one round of items per pass");
    static constexpr int NBLK4 = (FH * FWP + 31) / 32;                                            // 32-pixel blocks of the chain: 27
    static constexpr int NPAIRS = FR_R0H * NPAIR, NBLK5 = (NPAIRS + 15) / 16;                     // pixel pairs of IN: 342 -> 22 blocks
};

struct S2DStageParams {
    const float* x;               // N x 2 x H x W: [sparse depth, validity]
    long long x_bstride;
    const float* tab;             // SF_TAB floats
    const _Float16* wchain;       // [3][2][64][8]
    const _Float16* wconv;        // [4][2][64][8]
    float slope;                  // S2D's LeakyReLU (0 <= slope <= 1)
    int dbg;                      // phase ablation for tools/depth_front_bench.py (KBN_S2D_DEBUG): 1 no vertical pass, 2 no horizontal pass, 4 no 1x1 chain, 8 no 3x3 conv, 16 no depth loads
};

__device__ __forceinline__ float s2d_stage_lrelu(float v, float slope) { return fmaxf(v, v * slope); }   // 0 <= slope <= 1

// Runs P1-P5 for the tile whose IN region starts at image pixel (Y0, X0) = (2 oy0 - 2, 2 ox0 - 2); on return IN [term][pixel][8]
// (FR_NIN entries per term, the slack zeroed) is complete at smem + off_in -- the caller's barrier publishes it -- and
// (pre_in, un_in) is the window IN was split with, bound_in the bound on max |IN| it was derived from.  `red`: 8 floats of scratch outside the stage's and IN's bytes.
template <typename CFG>
__device__ __forceinline__ void s2d_stage_run(const S2DStageParams& sp, unsigned char* smem, int off_in, float* red, int n, int Y0, int X0,
                                              int H, int W, float& pre_in_out, float& un_in_out, float& bound_in_out) {
    using S = S2DStage<CFG>;
    constexpr int R = S::R, NP = S::NP, ZW = S::ZW, ZH = S::ZH, VP = S::VP, VPLANE = S::VPLANE;
    constexpr int IN_PART = FR_NIN * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    float* zmin = reinterpret_cast<float*>(smem + S::OFF_Z);
    float* zmax = zmin + ZH * ZW;
    float* vbuf = reinterpret_cast<float*>(smem + S::OFF_V);
    const long long plane = (long long)H * W;
    const float* xz = sp.x + (long long)n * sp.x_bstride;
    const int YF = Y0 - 1, XF = X0 - 1;          // feature region origin
    const int YZ = YF - R, XZ = XF - R;          // depth tile origin
    const bool f_interior = YF >= 0 && YF + S::FH <= H && XF >= 0 && XF + S::FWP <= W;   // block-uniform: no zero padding to apply

    // ---- P1: depth tile (+halo) -> zmin / zmax; a horizontal-pass item = (row fr, columns 4 q .., pool half): threads 0-209 take
    // pools 0-3 (channels 0-3 of the granule) and the raw z / v values of the pixels, threads 210-419 pools 4-7
    constexpr int NITEM = S::FH * S::NQ;
    const int half = tid >= NITEM ? 1 : 0, tbase = tid - half * NITEM;
    const int fr = tbase / S::NQ, q = tbase - fr * S::NQ;
    const bool item = tid < NITEM, item2 = tid < 2 * NITEM;
    float vr[6];
    float tm = 0.f;
    {
        const int Y = YF + fr;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int X = XF + 4 * q + j;
            const bool ok = item && Y >= 0 && Y < H && X >= 0 && X < W;
            vr[j] = ok ? xz[plane + (long long)Y * W + X] : 0.f;
            tm = fmaxf(tm, fabsf(vr[j]));
        }
        constexpr int MAXE = (ZH * ZW + FR_THREADS - 1) / FR_THREADS;
        const bool z_inside = YZ >= 0 && YZ + ZH <= H && XZ >= 0 && XZ + ZW <= W;   // block-uniform
        float vz[MAXE];
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = u * FR_THREADS + tid;
            const int r = e / ZW, c = e - r * ZW;
            const int Y2 = YZ + r, X2 = XZ + c;
            const bool ok = e < ZH * ZW && (z_inside || (Y2 >= 0 && Y2 < H && X2 >= 0 && X2 < W));
            vz[u] = (ok && !(sp.dbg & 16)) ? xz[(long long)Y2 * W + X2] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = u * FR_THREADS + tid;
            if (e < ZH * ZW) {
                zmax[e] = vz[u];                                                                  // -inf outside the image
                zmin[e] = (vz[u] == 0.f) ? 999.f : ((vz[u] == -INFINITY) ? INFINITY : vz[u]);   // where(z == 0, 999, z)
                if (vz[u] != -INFINITY) tm = fmaxf(tm, fabsf(vz[u]));
            }
        }
        tm = __uint_as_float(wave_max_bits(tm));
        if (lane == 0) red[wave] = tm;
    }
    __syncthreads();
    tm = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
    // windows: inputs (pooled depths, z, v) from the tile maximum; layer outputs from bounds
    const float b0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(tm)));
    const float b1 = sp.tab[0] * b0, b2 = sp.tab[1] * b1, b3 = sp.tab[2] * b2;
    const float bIN = sp.tab[3] * fmaxf(b3, b0);
    float pre0, un0, pre1, un1, pre2, un2, pre3, un3, preI, unI;
    fr_scales(__float_as_uint(b0), pre0, un0);
    fr_scales(__float_as_uint(b1), pre1, un1);
    fr_scales(__float_as_uint(b2), pre2, un2);
    fr_scales(__float_as_uint(b3), pre3, un3);
    fr_scales(__float_as_uint(bIN), preI, unI);
    pre_in_out = preI; un_in_out = unI; bound_in_out = bIN;

    // ---- P2: vertical pass -> V[pool][feature row][z column]
    if (tid < ZW * S::NGRP && !(sp.dbg & 1)) {
        const int g = tid / ZW, c = tid - g * ZW;
        auto sweep = [&](auto is_min_c, const float* zsrc) {
            constexpr bool IS_MIN = decltype(is_min_c)::value != 0;
            constexpr int RS = IS_MIN ? CFG::RMIN : CFG::RMAX;
            if constexpr (RS > 0) {
                float m[S::GS + 2 * RS];
                const float* s = zsrc + (g * S::GS + (R - RS)) * ZW + c;
#pragma unroll
                for (int i = 0; i < S::GS + 2 * RS; ++i) m[i] = s[i * ZW];
#pragma unroll
                for (int j = 0; j < S::GS; ++j) {
                    float a = m[j + RS];
                    s2d_for<1, RS + 1>([&](auto dc) {
                        constexpr int d = decltype(dc)::value;
                        a = IS_MIN ? fminf(a, fminf(m[j + RS - d], m[j + RS + d])) : fmaxf(a, fmaxf(m[j + RS - d], m[j + RS + d]));
                        s2d_for<0, NP>([&](auto pic) {
                            constexpr int pi = decltype(pic)::value;
                            if constexpr ((pi < CFG::NMINP) == IS_MIN && CFG::radius(pi) == d)
                                vbuf[pi * VPLANE + (g * S::GS + j) * VP + c] = a;
                        });
                    });
                }
            }
        };
        sweep(IntC<1>{}, zmin);
        sweep(IntC<0>{}, zmax);
    }
    // raw z of this item's six columns (the depth tile dies at the next barrier but one)
    float zr[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float z = item ? zmax[(fr + R) * ZW + 4 * q + j + R] : 0.f;
        zr[j] = (z == -INFINITY) ? 0.f : z;
    }
    __syncthreads();

    // ---- P3: horizontal pass: the pooled values of 4 consecutive feature pixels (csrc/s2d.hip P3, same compare / select arithmetic)
    float pooled[4][4];   // [pixel][pool 4 half + i]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) pooled[j][pi] = 0.f;
    if (item2 && !(sp.dbg & 2)) {
        s2d_for<0, NP>([&](auto pic) {
            constexpr int pi = decltype(pic)::value;
            if ((pi >> 2) != half) return;
            constexpr int r = CFG::radius(pi);
            constexpr bool IS_MIN = pi < CFG::NMINP;
            constexpr int OFF = R - r;                    // z column of window element o of pixel j: 4q + j + OFF + o
            constexpr int A0 = OFF & ~3, SH = OFF & 3;    // 16-byte aligned start, shift inside the first word
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
            pooled[0][pi & 3] = o0; pooled[1][pi & 3] = o1; pooled[2][pi & 3] = o2; pooled[3][pi & 3] = o3;
        });
    }
    __syncthreads();   // every V and depth-tile read is done: G overlays V, X2 the depth tile

    // ---- P3b: split granules of the chain's input (G) and of the raw channels (X2)
    unsigned char* const G = smem + S::OFF_G;
    unsigned char* const X2 = smem + S::OFF_X2;
    if (item2) {
        const int Y = YF + fr;
        const bool rowin = Y >= 0 && Y < H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int X = XF + 4 * q + j;
            const bool in = f_interior || (rowin && X >= 0 && X < W);
            ff4 a = {pooled[j][0], pooled[j][1], pooled[j][2], pooled[j][3]};
            if (!in) a = (ff4){0.f, 0.f, 0.f, 0.f};   // (windows of pixels outside the image may hold the +-inf sentinels)
            fh4 a1, a2;
            fr_split4(a * pre0, a1, a2);
            const int e = (fr * S::FWP + 4 * q + j) * 16 + half * 8;
            *reinterpret_cast<fh4*>(G + e) = a1;
            *reinterpret_cast<fh4*>(G + S::GPART + e) = a2;
        }
        if (item && 2 * q + 1 < S::NPAIR) {   // pairs 2q, 2q + 1 of this row: (z, v) of columns 4q .. 4q + 3 and 4q + 2 .. 4q + 5
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const ff4 a = {zr[2 * h], vr[2 * h], zr[2 * h + 1], vr[2 * h + 1]}, b = {zr[2 * h + 2], vr[2 * h + 2], zr[2 * h + 3], vr[2 * h + 3]};
                fh4 a1, a2, b1, b2;
                fr_split4(a * pre0, a1, a2);
                fr_split4(b * pre0, b1, b2);
                const int e = (fr * S::NPAIR + 2 * q + h) * 16;
                *reinterpret_cast<fh8*>(X2 + e) = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                *reinterpret_cast<fh8*>(X2 + S::X2PART + e) = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    } else if (tid - 2 * NITEM < S::NG - S::FH * S::FWP) {   // the slack behind the last feature pixel: finite values for the last block
        const int e = (S::FH * S::FWP + tid - 2 * NITEM) * 16;
        const fh8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        *reinterpret_cast<fh8*>(G + e) = zero;
        *reinterpret_cast<fh8*>(G + S::GPART + e) = zero;
    }
    __syncthreads();

    // ---- P4: the 1x1 chain, in place on G.  A block = 32 pixels: columns n of the MFMA carry pixels 32 b + n (k-groups 0, 1 = its h1, h2)
    //      and 32 b + 16 + n (k-groups 2, 3); rows 0-7 are the filters at the first pixel, rows 8-15 the same filters at the second:
    //      every lane ends up with half a granule to write.  Two instructions per layer: [w1 | w1 2^-11] . [h1 ; h2] and [w2 | 0] . [h1 ; h2].
    //      A wave runs its (up to four) blocks layer by layer, so that four independent read -> MFMA -> split -> write chains overlap.
    {
        fh8 A[3][2];
        ff4 sc[3];
        const float us[3] = {un0 * pre1, un1 * pre2, un2 * pre3};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            A[i][0] = *reinterpret_cast<const fh8*>(sp.wchain + ((i * 2 + 0) * 64 + lane) * 8);
            A[i][1] = *reinterpret_cast<const fh8*>(sp.wchain + ((i * 2 + 1) * 64 + lane) * 8);
            sc[i] = *reinterpret_cast<const ff4*>(sp.tab + 8 + 8 * i + 4 * (kq & 1)) * us[i];
        }
        constexpr int NU = (S::NBLK4 + 7) / 8;
        unsigned char* gp[NU];
        bool in[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int px = 32 * (wave + 8 * u) + 16 * (kq >> 1) + l15;
            gp[u] = G + px * 16;
            const int pr = px / S::FWP, pc = px - pr * S::FWP;
            in[u] = f_interior || (YF + pr >= 0 && YF + pr < H && XF + pc >= 0 && XF + pc < W);
        }
        const int rd = (kq & 1) * S::GPART, wr = (kq & 1) * 8;
        const int nblk = (sp.dbg & 4) ? 0 : S::NBLK4;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (wave + 8 * u < nblk) {   // wave-uniform
                    const fh8 bv = *reinterpret_cast<const fh8*>(gp[u] + rd);
                    ff4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[i][0], bv, (ff4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[i][1], bv, d, 0, 0, 0);
                    d *= sc[i];
                    ff4 v = {s2d_stage_lrelu(d[0], sp.slope), s2d_stage_lrelu(d[1], sp.slope), s2d_stage_lrelu(d[2], sp.slope), s2d_stage_lrelu(d[3], sp.slope)};
                    if (i == 2 && !in[u]) v = (ff4){0.f, 0.f, 0.f, 0.f};   // zero padding of the 3x3 conv's input
                    fh4 h1, h2;
                    fr_split4(v, h1, h2);
                    *reinterpret_cast<fh4*>(gp[u] + wr) = h1;
                    *reinterpret_cast<fh4*>(gp[u] + S::GPART + wr) = h2;
                }
            }
            // the next layer's fh8 reads of these granules must stay behind the fh4 writes (another lane of the wave wrote the other
            // half: the LDS keeps a wave's accesses in order, the compiler is told here -- it may otherwise reorder accesses of
            // different vector types)
            asm volatile("" ::: "memory");
        }
    }
    __syncthreads();

    // ---- P5: 3x3 conv over [features | z, v] for pixel pairs -> IN
    {
        fh8 aF[3][2], aR[2];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) aF[s][t] = *reinterpret_cast<const fh8*>(sp.wconv + ((s * 2 + t) * 64 + lane) * 8);
#pragma unroll
        for (int t = 0; t < 2; ++t) aR[t] = *reinterpret_cast<const fh8*>(sp.wconv + ((3 * 2 + t) * 64 + lane) * 8);
        const ff4 inv = *reinterpret_cast<const ff4*>(sp.tab + 32 + 4 * (kq & 1));
        const ff4 scF = inv * (un3 * preI), scR = inv * (un0 * preI);
        unsigned char* const IN = smem + off_in;
        // a wave's (up to three) blocks as straight-line code: their fragment reads and MFMAs interleave; a wave without a third block
        // repeats the last one (same values to the same addresses)
        constexpr int NU5 = (S::NBLK5 + 7) / 8;
        if (!(sp.dbg & 8))
#pragma unroll
        for (int u = 0; u < NU5; ++u) {
            const int b = min(wave + 8 * u, S::NBLK5 - 1);
            const int qq = 16 * b + l15;
            const bool valid = qq < S::NPAIRS;
            const int qc = valid ? qq : S::NPAIRS - 1;
            const int y = qc / S::NPAIR, pp = qc - y * S::NPAIR;
            const unsigned char* fb = G + (y * S::FWP + 2 * pp + kq) * 16;
            const unsigned char* rb = X2 + ((y + (kq < 3 ? kq : 2)) * S::NPAIR + pp) * 16;
            ff4 mF = {0.f, 0.f, 0.f, 0.f}, sF = mF, mR = mF, sR = mF;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const fh8 b1 = *reinterpret_cast<const fh8*>(fb + s * S::FWP * 16);
                const fh8 b2 = *reinterpret_cast<const fh8*>(fb + S::GPART + s * S::FWP * 16);
                mF = __builtin_amdgcn_mfma_f32_16x16x32_f16(aF[s][0], b1, mF, 0, 0, 0);
                sF = __builtin_amdgcn_mfma_f32_16x16x32_f16(aF[s][1], b1, sF, 0, 0, 0);
                sF = __builtin_amdgcn_mfma_f32_16x16x32_f16(aF[s][0], b2, sF, 0, 0, 0);
            }
            {
                const fh8 b1 = *reinterpret_cast<const fh8*>(rb);
                const fh8 b2 = *reinterpret_cast<const fh8*>(rb + S::X2PART);
                mR = __builtin_amdgcn_mfma_f32_16x16x32_f16(aR[0], b1, mR, 0, 0, 0);
                sR = __builtin_amdgcn_mfma_f32_16x16x32_f16(aR[1], b1, sR, 0, 0, 0);
                sR = __builtin_amdgcn_mfma_f32_16x16x32_f16(aR[0], b2, sR, 0, 0, 0);
            }
            const int x = 2 * pp + (kq >> 1);                   // this lane's pixel of the pair, IN coordinates
            const int Y = Y0 + y, X = X0 + x;
            const bool in = Y >= 0 && Y < H && X >= 0 && X < W;
            ff4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = __builtin_fmaf(sF[r], 0.00048828125f, mF[r]) * scF[r] + __builtin_fmaf(sR[r], 0.00048828125f, mR[r]) * scR[r];
                v[r] = in ? s2d_stage_lrelu(t, sp.slope) : 0.f;
            }
            fh4 h1, h2;
            fr_split4(v, h1, h2);
            if (valid && x < FR_R0W) {
                const int e = (y * FR_R0W + x) * 16 + (kq & 1) * 8;
                *reinterpret_cast<fh4*>(IN + e) = h1;
                *reinterpret_cast<fh4*>(IN + IN_PART + e) = h2;
            }
        }
        if (tid < 2 * (FR_NIN - FR_NP0)) {   // the zeroed slack behind the last pixel (a column pair of phase B may start there)
            const fh8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<fh8*>(IN + (tid & 1) * IN_PART + (FR_NP0 + (tid >> 1)) * 16) = zero;
        }
    }
}

}  // namespace kbn
