// kb_pair_impl.h -- kernel template of the fused KB block (see kb_pair.hip); instantiated per n-block count in
// kb_pair_nb3.hip / kb_pair_nb4.hip so that the two sets build in parallel.
#pragma once

#include "conv_dma_impl.h"

namespace kbn {


struct KbPairParams {
    KbPairArgs a;
    int CpadF;                 // padded input channels of the conv_fused blob
    int CpadD;                 // padded input channels (depth + 3) of the conv_depth blob   (NBD > 0)
    int nTilesND;              // n-tiles of conv_depth: workgroups with nt < nTilesND also compute that tile
    int outH, outW, tilesX, tilesY, nTilesN, nblocks;
};

template <int NB, int MW, int TWB, int NBD = 0>
struct PairGeom {
    using G3 = DmaGeom<3, 2, MW, TWB>;
    using G1 = DmaGeom<1, 2, MW, TWB>;
    static constexpr int NT = NB * 16;
    static constexpr int NTD = NBD * 16;                     // conv_depth filters per workgroup (NBD <= NB: fits the stage)
    static constexpr int A1 = 4 * G3::PLANE;                 // phase-1 stage: 4 channel tiles ...
    static constexpr int B1 = 4 * 9 * NT + 4 * NT;           // ... + image slice + fused slice
    static constexpr int A2 = 8 * G1::PLANE;                 // phase-2 stage: 8 channel tiles + fused slice
    static constexpr int B2 = 8 * NT;
    static constexpr int BUF = (A1 + B1 > A2 + B2) ? (A1 + B1) : (A2 + B2);
    static constexpr int XYZ = 3 * G1::PLANE;
    static constexpr size_t LDS_BYTES = sizeof(float) * (2 * (size_t)BUF + XYZ);
};

template <int NB, int MW, int TWB, int NBD>
__global__ __launch_bounds__(256, 2) void kb_pair_kernel(const KbPairParams p) {
    using PG = PairGeom<NB, MW, TWB, NBD>;
    static_assert(NBD <= NB, "the conv_depth weight slice must fit the phase-1 stage");
    using G3 = typename PG::G3;
    using G1 = typename PG::G1;
    constexpr int NT = PG::NT, TH = G3::TH, TW = G3::TW;
    constexpr int NBDR = NBD > 0 ? NBD : 1;                  // array extent (no zero-length arrays)
    constexpr int NTD = PG::NTD;
    constexpr int PLANE3 = G3::PLANE, PITCH3 = G3::COLS, MAXJ3 = G3::MAXJ;
    constexpr int PLANE1 = G1::PLANE, PITCH1 = G1::COLS, MAXJ1 = G1::MAXJ;
    constexpr int BUF = PG::BUF;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xyz_s = smem + 2 * BUF;

    const KbPairArgs& a = p.a;
    const int tid = threadIdx.x;
    int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int nt = bid % p.nTilesN;
    bid /= p.nTilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int inH = a.height, inW = a.width, HWin = inH * inW;

    // ---- this lane's DMA granules (byte offset inside a channel plane, or -1) for the two geometries ----
    int goff3[MAXJ3], goff1[MAXJ1];
    {
        const int Y0 = oy0 * 2 - 1, XA = ox0 * 2 - 4;
        constexpr int cv4 = G3::COLS >> 2;
#pragma unroll
        for (int j = 0; j < MAXJ3; ++j) {
            const int f = j * 64 + lane;
            int g = -1;
            if (f < G3::NF4) {
                const int r = f / cv4, cv = f - r * cv4;
                const int Y = Y0 + r, X = XA + cv * 4;
                if (Y >= 0 && Y < inH && X >= 0 && X < inW) g = (Y * inW + X) * 4;
            }
            goff3[j] = g;
        }
    }
    {
        constexpr int cv4 = G1::COLS >> 2;
#pragma unroll
        for (int j = 0; j < MAXJ1; ++j) {
            const int f = j * 64 + lane;
            int g = -1;
            if (f < G1::NF4) {
                const int r = f / cv4, cv = f - r * cv4;
                const int Y = oy0 * 2 + r * 2, X = ox0 * 2 + cv * 4;
                if (Y < inH && X < inW) g = (Y * inW + X) * 4;
            }
            goff1[j] = g;
        }
    }

    // ---- per-lane fragment addressing ----
    int mbase3[MW], mbase1[MW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const int mb = wave * MW + mi;
        const int oy = mb / TWB;
        const int seg = mb - oy * TWB;
        mbase3[mi] = 2 * oy * PITCH3 + 2 * (seg * 16 + li) + 3 + lk * PLANE3;
        mbase1[mi] = oy * PITCH1 + 2 * (seg * 16 + li) + lk * PLANE1;
    }
    const int boff = (lk >> 1) * 2 * NT + li * 2 + (lk & 1);

    f32x4 accI[MW][NB], accF[MW][NB], accD[MW][NBDR];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            accI[mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accF[mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int nb = 0; nb < NBDR; ++nb) accD[mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_depth = NBD > 0 && nt < p.nTilesND;        // block-uniform

    const int Ci = a.channels_image, Cf = a.channels_fused;
    const float* wI = a.wp_image + (long long)nt * Ci * 9 * NT;
    const float* wF = a.wp_fused + (long long)nt * p.CpadF * NT;
    const float* img_w = a.image + (long long)n * a.image_bstride + (long long)wave * HWin;  // channel (c0 + wave)
    const float* fus_n = a.fused ? a.fused + (long long)n * a.fused_bstride : nullptr;

    // wave-uniform contiguous copy global -> LDS, n4 16-byte granules, spread over the four waves
    auto dma_copy = [&](const float* src, unsigned dst, int n4) {
        for (int e0 = 0; e0 < n4; e0 += 256) {
            const int eb = e0 + wave * 64;
            if (eb + lane < n4) lds_dma16_s(src + eb * 4, (unsigned)(lane * 16), dst + eb * 16);
        }
    };

    // ---- phase 0 (conv_depth, reference src/net_utils.py:1351): conv3x3 s2 on cat[depth, coordinates] ----
    const int Cd = a.channels_depth;
    const float* dep_n0 = a.depth + (long long)n * a.depth_bstride;
    const float* wD = NBD > 0 ? a.wp_depth + (long long)nt * p.CpadD * 9 * NTD : nullptr;
    const int boffD = (lk >> 1) * 2 * NTD + li * 2 + (lk & 1);
    auto stage0 = [&](float* As, int c0) {   // wave w: channel c0 + w = a depth plane, a coordinate plane or padding
        const int c = c0 + wave;                       // wave-uniform
        float* plane = As + wave * PLANE3;
        const float* cptr = nullptr;
        if (c < Cd) cptr = dep_n0 + (long long)c * HWin;
        else if (c < Cd + 3 && a.coords) cptr = a.coords + (long long)n * a.coords_bstride + (long long)(c - Cd) * HWin;
        if (cptr) {
            cptr = uniform_ptr(cptr);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(plane));
#pragma unroll
            for (int j = 0; j < MAXJ3; ++j)
                if (j * 64 < G3::NF4 && goff3[j] >= 0) lds_dma16_s(cptr, (unsigned)goff3[j], dst + j * 1024);
        } else if (c < Cd + 3) {                       // K^-1 [x y 1]^T row (c - Cd), zero outside the image
            const float* kv = a.kinv + (long long)n * 9 + (c - Cd) * 3;
            const float k0 = kv[0], k1 = kv[1], k2 = kv[2];
            const int Y0 = oy0 * 2 - 1, XA = ox0 * 2 - 4;
            for (int e = lane; e < G3::ROWS * G3::COLS; e += 64) {
                const int r = e / G3::COLS, cx = e - r * G3::COLS;
                const int Y = Y0 + r, X = XA + cx;
                const bool inb = Y >= 0 && Y < inH && X >= 0 && X < inW;
                plane[e] = inb ? fmaf(k1, (float)Y, k0 * (float)X) + k2 : 0.f;
            }
        } else {                                       // channel padding of the blob
            for (int e = lane * 4; e < PLANE3; e += 256) *reinterpret_cast<f32x4*>(plane + e) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const unsigned bs = __builtin_amdgcn_readfirstlane(lds_addr(As + PG::A1));
        dma_copy(wD + (long long)c0 * 9 * NTD, bs, 9 * NTD);
    };

    auto compute0 = [&](const float* As) {
        const float* Bs = As + PG::A1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const float* Ab = As + ky * PITCH3 + kx;
            const float* Bb = Bs + tap * 4 * NTD + boffD;
            float av[MW], bv[NBDR];
#pragma unroll
            for (int mi = 0; mi < MW; ++mi) av[mi] = Ab[mbase3[mi]];
#pragma unroll
            for (int nb = 0; nb < NBDR; ++nb) bv[nb] = Bb[nb * 32];
#pragma unroll
            for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                for (int nb = 0; nb < NBDR; ++nb)
                    accD[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bv[nb], accD[mi][nb], 0, 0, 0);
        }
    };

    auto stage1 = [&](float* As, int c0) {   // image channels c0..c0+3 (wave w: channel c0 + w) + both weight slices
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(As + wave * PLANE3));
#pragma unroll
        for (int j = 0; j < MAXJ3; ++j)
            if (j * 64 < G3::NF4 && goff3[j] >= 0) lds_dma16_s(img_w, (unsigned)goff3[j], dst + j * 1024);
        img_w += 4LL * HWin;
        const unsigned bs = __builtin_amdgcn_readfirstlane(lds_addr(As + PG::A1));
        dma_copy(wI + (long long)c0 * 9 * NT, bs, 9 * NT);
        dma_copy(wF + (long long)c0 * NT, bs + 36 * NT * 4, NT);
    };

    auto stage2 = [&](float* As, int k0) {   // conv_fused inputs Ci + k0 .. + 7: xyz (k0 == 0), fused, zero padding
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int q = wave + 4 * t;
            const int c = k0 + q;                      // wave-uniform
            float* plane = As + q * PLANE1;
            if (c < 3) {
                for (int e = lane * 4; e < PLANE1; e += 256)
                    *reinterpret_cast<f32x4*>(plane + e) = *reinterpret_cast<const f32x4*>(xyz_s + c * PLANE1 + e);
            } else if (c - 3 < Cf) {
                const float* cptr = uniform_ptr(fus_n + (long long)(c - 3) * HWin);
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(plane));
#pragma unroll
                for (int j = 0; j < MAXJ1; ++j)
                    if (j * 64 < G1::NF4 && goff1[j] >= 0) lds_dma16_s(cptr, (unsigned)goff1[j], dst + j * 1024);
            } else {                                   // channel padding of the blob: weights are 0, keep 0 * x finite
                for (int e = lane * 4; e < PLANE1; e += 256)
                    *reinterpret_cast<f32x4*>(plane + e) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        const unsigned bs = __builtin_amdgcn_readfirstlane(lds_addr(As + PG::A2));
        dma_copy(wF + (long long)(Ci + k0) * NT, bs, 2 * NT);
    };

    auto compute1 = [&](const float* As) {
        const float* Bs = As + PG::A1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const float* Ab = As + ky * PITCH3 + kx;
            const float* Bb = Bs + tap * 4 * NT + boff;
            float av[MW], bv[NB];
#pragma unroll
            for (int mi = 0; mi < MW; ++mi) av[mi] = Ab[mbase3[mi]];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = Bb[nb * 32];
#pragma unroll
            for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    accI[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bv[nb], accI[mi][nb], 0, 0, 0);
            if (tap == 4) {   // the pixels a 1x1 stride-2 conv reads
                const float* Bf = Bs + 36 * NT + boff;
                float fv[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) fv[nb] = Bf[nb * 32];
#pragma unroll
                for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        accF[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], fv[nb], accF[mi][nb], 0, 0, 0);
            }
        }
    };

    auto compute2 = [&](const float* As) {
        const float* Bs = As + PG::A2;
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            const float* Ab = As + c4 * 4 * PLANE1;
            const float* Bb = Bs + c4 * 4 * NT + boff;
            float av[MW], bv[NB];
#pragma unroll
            for (int mi = 0; mi < MW; ++mi) av[mi] = Ab[mbase1[mi]];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = Bb[nb * 32];
#pragma unroll
            for (int mi = 0; mi < MW; ++mi)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    accF[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bv[nb], accF[mi][nb], 0, 0, 0);
        }
    };

    // ---- clear the phase-1 tiles of both stages once (out-of-image granules are never written) ----
    {
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int e = tid * 4; e < PG::A1; e += 1024) {
            *reinterpret_cast<f32x4*>(smem + e) = zero;
            *reinterpret_cast<f32x4*>(smem + BUF + e) = zero;
        }
    }
    __syncthreads();
    if (do_depth) stage0(smem, 0);
    else stage1(smem, 0);

    // ---- xyz at the pixels conv_fused samples (reference src/net_utils.py:1354-1360), while the DMAs fly ----
    {
        const float* kinv = a.kinv ? a.kinv + (long long)n * 9 : nullptr;
        const float* dep_n = a.depth + (long long)n * a.depth_bstride;
        constexpr int ncols = G1::COLS / 2;
        for (int e = tid; e < G1::ROWS * ncols; e += 256) {
            const int r = e / ncols, cx = (e - r * ncols) * 2;
            const int Y = oy0 * 2 + r * 2, X = ox0 * 2 + cx;
            float cv[3] = {0.f, 0.f, 0.f};
            float z = 0.f;
            if (Y < inH && X < inW) {
                const int g = Y * inW + X;
                if (a.coords) {
                    const float* cb = a.coords + (long long)n * a.coords_bstride + g;
                    cv[0] = cb[0]; cv[1] = cb[HWin]; cv[2] = cb[2 * HWin];
                } else {
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        cv[j] = fmaf(kinv[j * 3 + 1], (float)Y, kinv[j * 3 + 0] * (float)X) + kinv[j * 3 + 2];
                }
                const float* db = dep_n + g;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // same four chains as conv_dma's staging code
                int c = 0;
                for (; c + 3 < a.channels_depth; c += 4) {
                    a0 = fmaf(a.proj[c], db[(long long)c * HWin], a0);
                    a1 = fmaf(a.proj[c + 1], db[(long long)(c + 1) * HWin], a1);
                    a2 = fmaf(a.proj[c + 2], db[(long long)(c + 2) * HWin], a2);
                    a3 = fmaf(a.proj[c + 3], db[(long long)(c + 3) * HWin], a3);
                }
                for (; c < a.channels_depth; ++c) a0 = fmaf(a.proj[c], db[(long long)c * HWin], a0);
                z = leaky_relu((a0 + a1) + (a2 + a3), a.slope);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) xyz_s[j * PLANE1 + r * PITCH1 + cx] = cv[j] * z;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    const int K2 = p.CpadF - Ci;
    if (do_depth) {
        for (int c0 = 0; c0 < p.CpadD; c0 += 4) {
            float* curA = smem + cur * BUF;
            float* nxtA = smem + (cur ^ 1) * BUF;
            if (c0 + 4 < p.CpadD) stage0(nxtA, c0 + 4);
            else stage1(nxtA, 0);
            compute0(curA);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
    }
    for (int c0 = 0; c0 < Ci; c0 += 4) {
        float* curA = smem + cur * BUF;
        float* nxtA = smem + (cur ^ 1) * BUF;
        if (c0 + 4 < Ci) stage1(nxtA, c0 + 4);
        else stage2(nxtA, 0);
        compute1(curA);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    for (int k0 = 0; k0 < K2; k0 += 8) {
        float* curA = smem + cur * BUF;
        float* nxtA = smem + (cur ^ 1) * BUF;
        if (k0 + 8 < K2) stage2(nxtA, k0 + 8);
        compute2(curA);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    const StoreDst dI{a.out_image, a.out_image_bstride, p.outH, p.outW, a.filters, TWB, 1, a.slope, a.absmax_image};
    store_tile_dst<NB, MW>(dI, accI, n, nt, oy0, ox0, wave, li, lk);
    const StoreDst dF{a.out_fused, a.out_fused_bstride, p.outH, p.outW, a.filters, TWB, 1, a.slope, a.absmax_fused};
    store_tile_dst<NB, MW>(dF, accF, n, nt, oy0, ox0, wave, li, lk);
    if constexpr (NBD > 0) {
        if (do_depth) {
            const StoreDst dD{a.out_depth, a.out_depth_bstride, p.outH, p.outW, a.filters_depth, TWB, 1, a.slope, a.absmax_depth};
            store_tile_dst<NBD, MW>(dD, accD, n, nt, oy0, ox0, wave, li, lk);
        }
    }
}

template <int NB, int MW, int TWB, int NBD>
int pair_variant(KbPairParams& p, hipStream_t stream) {
    using PG = PairGeom<NB, MW, TWB, NBD>;
    if constexpr (MW * (2 * NB + NBD) > 40) return KBN_ERR_UNSUPPORTED;   // accumulators would not stay in registers
    else {
    auto kern = kb_pair_kernel<NB, MW, TWB, NBD>;
    if (PG::LDS_BYTES > 160 * 1024) return KBN_ERR_UNSUPPORTED;
    static DeviceOnce once;
    if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
    p.tilesX = ceil_div(p.outW, PG::G3::TW);
    p.tilesY = ceil_div(p.outH, PG::G3::TH);
    const long long nb64 = (long long)p.tilesX * p.tilesY * p.a.n * p.nTilesN;
    if (nb64 > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.nblocks = (int)nb64;
    hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), PG::LDS_BYTES, stream, p);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
    }
}

// candidate tile shapes (MW m-blocks per wave, TWB 16-pixel segments per tile row)
constexpr int kPairCands = 6;
constexpr int kPairMW[kPairCands] = {4, 4, 2, 2, 1, 1};
constexpr int kPairTWB[kPairCands] = {2, 1, 2, 1, 2, 1};

template <int NB, int NBD>
int pair_dispatch(KbPairParams& p, int cand, hipStream_t st) {
    switch (cand) {
        case 0: return pair_variant<NB, 4, 2, NBD>(p, st);
        case 1: return pair_variant<NB, 4, 1, NBD>(p, st);
        case 2: return pair_variant<NB, 2, 2, NBD>(p, st);
        case 3: return pair_variant<NB, 2, 1, NBD>(p, st);
        case 4: return pair_variant<NB, 1, 2, NBD>(p, st);
        default: return pair_variant<NB, 1, 1, NBD>(p, st);
    }
}

// kb_pair_nb3.hip / kb_pair_nb4.hip: (NB, NBD) -> instantiated kernels; cand indexes kPairMW / kPairTWB
int kb_pair_dispatch_nb3(KbPairParams& p, int nbd, int cand, hipStream_t st);
int kb_pair_dispatch_nb4(KbPairParams& p, int nbd, int cand, hipStream_t st);

}  // namespace kbn
