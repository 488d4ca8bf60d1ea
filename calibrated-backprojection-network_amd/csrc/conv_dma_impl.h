// conv_dma_impl.h -- kernel template and per-tile-width launcher of the LDS-DMA conv (see conv_dma.hip).
// Instantiated once per tile width TWB in conv_dma_t1/t2/t4.hip so that the three sets build in parallel.
#pragma once

#include "conv_common.h"

namespace kbn {

// Tile geometry of a (KS, STRIDE, MW, TWB) variant -- all compile time, so that every LDS address of
// the K loop is a register plus an immediate (vector-ALU address arithmetic between MFMAs costs
// matrix-pipe time, tools/probe/issue_probe.hip).
template <int KS, int STRIDE, int MW, int TWB>
struct DmaGeom {
    static constexpr int TH = 4 * MW / TWB, TW = TWB * 16;
    static constexpr int ROWS = (KS == 3) ? (STRIDE == 2 ? 2 * TH + 1 : TH + 2) : TH;
    static constexpr int COLS = (KS == 3) ? (STRIDE == 2 ? 2 * TW + 4 : TW + 8) : STRIDE * TW;
    static constexpr int PLANE = ((ROWS * COLS + 15) / 32) * 32 + 16;  // 16 (mod 32): k / k+1 rows of a half-wave on disjoint banks
    static constexpr int NF4 = ROWS * (COLS / 4);                       // granules per channel tile
    static constexpr int MAXJ = (NF4 + 63) / 64;
};

template <int KS, int STRIDE, int CK, int NB, int MW, int TWB, bool SYN>
__global__ __launch_bounds__(256, KBN_WAVES_PER_SIMD) void conv_dma_kernel(const ConvParams p) {
    using G = DmaGeom<KS, STRIDE, MW, TWB>;
    constexpr int MAXJ = G::MAXJ, PLANE = G::PLANE, PITCH = G::COLS, TH = G::TH;
    constexpr int TAPS = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int YSTEP = (KS == 1) ? STRIDE : 1;  // input rows per staged row
    constexpr int NT = NB * 16;
    constexpr int NC4 = CK / 4;
    constexpr int B_FLOATS = CK * TAPS * NT;
    constexpr int XS = STRIDE;  // fragment column step

    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int a_floats = CK * PLANE;
    constexpr int buf_floats = a_floats + B_FLOATS;

    const int tid = threadIdx.x;
    int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int nt = bid % p.nTilesN;
    bid /= p.nTilesN;
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int TW = G::TW;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> scalar staging code
    const int li = lane & 15, lk = lane >> 4;

    // first staged input row / column (column aligned down to a multiple of 4)
    const int Y0 = oy0 * STRIDE - PAD;
    const int XA = ox0 * STRIDE - (KS == 3 ? 4 : 0);
    constexpr int cv4 = G::COLS >> 2;    // granules per staged row
    constexpr int nf4 = G::NF4;          // granules per channel tile

    // ---- this lane's granules: element offset inside a channel plane of the source, or -1 ----
    int goff[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int f = j * 64 + lane;
        int g = -1;
        if (f < nf4) {
            const int r = f / cv4, cv = f - r * cv4;
            const int Y = Y0 + r * YSTEP, X = XA + cv * 4;
            if (Y >= 0 && Y < p.inH && X >= 0 && X < p.inW) g = (Y * p.inW + X) * 4;  // byte offset in the plane
        }
        goff[j] = g;
    }
    int syn = -1;  // SYN: the launch has a computed (KB layer) source
    if constexpr (SYN) {
#pragma unroll
        for (int s = 0; s < KBN_MAX_SRC; ++s)
            if (s < p.nsrc && p.src[s].kind != KBN_SRC_TENSOR) syn = s;
    }

    // ---- per-lane fragment addressing ----------------------------------------------
    int mbase[MW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        const int mb = wave * MW + mi;
        const int oy = mb / TWB;
        const int seg = mb - oy * TWB;
        const int row = (KS == 3) ? STRIDE * oy : oy;
        mbase[mi] = row * PITCH + XS * (seg * 16 + li) + (KS == 3 ? 3 : 0) + lk * PLANE;
    }
    const int boff = (lk >> 1) * 2 * NT + li * 2 + (lk & 1);

    f32x4 acc[MW][NB];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* wp_nt = p.wp + (long long)nt * p.Cpad * TAPS * NT;

    // current source of the K loop (wave-uniform; chunks are staged in increasing c0 order)
    // Source tracking for the K loop (wave-uniform scalars; chunks are staged in increasing c0
    // order).  !SYN launches are "aligned": every source but the last ends on a chunk boundary
    // (conv_dma_launch checks), so one running pointer per wave is all the staging needs.
    int cs_idx = 0, cs_start = 0, cs_end = p.src[0].C, cs_hw = p.src[0].H * p.src[0].W;
    bool cs_tensor = p.src[0].kind == KBN_SRC_TENSOR;
    const float* cs_base = p.src[0].data + (long long)n * p.src[0].bstride;
    const int HWin = p.inH * p.inW;
    const float* wptr = cs_base + (long long)wave * HWin;  // channel (c0 + wave) of the current source
    int s_left = p.src[0].C;                               // channels of the current source from c0 on

    // ---- stage one chunk: A tile (CK channels) + B slice, all by LDS-DMA ----------------
    auto stage = [&](float* As, int c0) {
        if (!(p.dbg & 1)) {
            if constexpr (!SYN) {
                if (s_left <= 0 && cs_idx + 1 < p.nsrc) {
                    ++cs_idx;
                    wptr = p.src[cs_idx].data + (long long)n * p.src[cs_idx].bstride + (long long)wave * HWin;
                    s_left = p.src[cs_idx].C;
                }
#pragma unroll
                for (int t = 0; t < NC4; ++t) {  // wave w moves channels w, w+4, ... of the chunk
                    const int q = wave + 4 * t;
                    if (q < s_left) {
                        const float* cptr = wptr + (long long)(4 * t) * HWin;
                        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(As + q * PLANE));
#pragma unroll
                        for (int j = 0; j < MAXJ; ++j) {
                            if (j * 64 < nf4 && goff[j] >= 0)
                                lds_dma16_s(cptr, (p.dbg & 8) ? (unsigned)(lane * 16) : (unsigned)goff[j], dst + j * 1024);
                        }
                    }
                }
                wptr += (long long)CK * HWin;
                s_left -= CK;
            } else {
            // generic: chunks may straddle sources / hold computed channels
            while (c0 >= cs_end && cs_idx + 1 < p.nsrc) {  // advance to the source that holds c0
                ++cs_idx;
                cs_start = p.src[cs_idx].cstart;
                cs_end = cs_start + p.src[cs_idx].C;
                cs_tensor = p.src[cs_idx].kind == KBN_SRC_TENSOR;
                cs_hw = p.src[cs_idx].H * p.src[cs_idx].W;
                cs_base = p.src[cs_idx].data + (long long)n * p.src[cs_idx].bstride;
            }
            const bool inside = cs_tensor && c0 >= cs_start && c0 + CK <= cs_end;
#pragma unroll
            for (int t = 0; t < NC4; ++t) {
                const int q = wave + 4 * t;
                const float* cptr = nullptr;
                if (inside) {
                    cptr = cs_base + (long long)(c0 + q - cs_start) * cs_hw;
                } else {
                    const ChanRef cr = chan_lookup(p, n, c0 + q);  // wave-uniform
                    if (cr.kind == KBN_SRC_TENSOR) cptr = cr.ptr;
                }
                if (cptr) {
                    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(As + q * PLANE));
#pragma unroll
                    for (int j = 0; j < MAXJ; ++j) {
                        if (j * 64 < nf4 && goff[j] >= 0)
                            lds_dma16_s(cptr, (p.dbg & 8) ? (unsigned)(lane * 16) : (unsigned)goff[j], dst + j * 1024);
                    }
                }
            }
            }
            // computed channels (KB layer) and channel padding: plain stores, rare
            bool has_syn = false;
            if constexpr (SYN) has_syn = syn >= 0 && c0 + CK > p.src[syn].cstart && c0 < p.src[syn].cstart + 3;
            if (has_syn) {
                const SrcDev& sd = p.src[syn];
                const int HW = p.inH * p.inW;
                const float* kinv = sd.kinv ? sd.kinv + (long long)n * 9 : nullptr;
                // only the columns the fragments read: every one for 3x3, every STRIDE-th for 1x1
                constexpr int CSTEP = (KS == 1) ? STRIDE : 1;
                const int ncols = G::COLS / CSTEP;
                for (int e = tid; e < G::ROWS * ncols; e += 256) {
                    const int r = e / ncols, cx = (e - r * ncols) * CSTEP;
                    const int Y = Y0 + r * YSTEP, X = XA + cx;
                    float cv[3] = {0.f, 0.f, 0.f};
                    float z = 1.f;
                    const bool inb = (Y >= 0 && Y < p.inH && X >= 0 && X < p.inW);
                    if (inb) {
                        const int g = Y * p.inW + X;
                        if (sd.kind == KBN_SRC_XYZ && sd.coords) {
                            const float* cb = sd.coords + (long long)n * sd.coords_bstride + g;
                            cv[0] = cb[0]; cv[1] = cb[HW]; cv[2] = cb[2 * HW];
                        } else {
#pragma unroll
                            for (int j = 0; j < 3; ++j)
                                cv[j] = fmaf(kinv[j * 3 + 1], (float)Y, kinv[j * 3 + 0] * (float)X) + kinv[j * 3 + 2];
                        }
                        if (sd.kind == KBN_SRC_XYZ) {
                            const float* db = sd.data + (long long)n * sd.bstride + g;
                            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // 4 chains: loads issue ahead of the FMAs
                            int c = 0;
                            for (; c + 3 < sd.Cd; c += 4) {
                                a0 = fmaf(sd.proj[c], db[(long long)c * HW], a0);
                                a1 = fmaf(sd.proj[c + 1], db[(long long)(c + 1) * HW], a1);
                                a2 = fmaf(sd.proj[c + 2], db[(long long)(c + 2) * HW], a2);
                                a3 = fmaf(sd.proj[c + 3], db[(long long)(c + 3) * HW], a3);
                            }
                            for (; c < sd.Cd; ++c) a0 = fmaf(sd.proj[c], db[(long long)c * HW], a0);
                            const float a = (a0 + a1) + (a2 + a3);
                            z = p.act ? leaky_relu(a, p.slope) : a;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int q = sd.cstart + j - c0;
                        if (q >= 0 && q < CK) As[q * PLANE + r * PITCH + cx] = inb ? cv[j] * z : 0.f;
                    }
                }
            }
            if (c0 + CK > p.Ctot) {  // channel padding of the last chunk: clear stale planes
                const int q0 = p.Ctot - c0;
                for (int e = tid; e < (CK - q0) * PLANE; e += 256) As[q0 * PLANE + e] = 0.f;
            }
        }
        if (!(p.dbg & 2)) {
            constexpr int CNT4 = B_FLOATS / 4;
            const float4* s4 = reinterpret_cast<const float4*>(wp_nt + (long long)c0 * TAPS * NT);
            const unsigned bs = __builtin_amdgcn_readfirstlane(lds_addr(As + a_floats));
#pragma unroll
            for (int e0 = 0; e0 < CNT4; e0 += 256) {
                const int eb = e0 + wave * 64;
                if (eb + lane < CNT4) lds_dma16_s(reinterpret_cast<const float*>(s4 + eb), (unsigned)(lane * 16), bs + eb * 16);
            }
        }
    };

    auto compute = [&](const float* As, const float* Bs) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            const int toff = (KS == 1) ? 0 : (ky * PITCH + kx);
#pragma unroll
            for (int c4 = 0; c4 < NC4; ++c4) {
                const float* Ab = As + c4 * 4 * PLANE + toff;
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

    // ---- clear both A stages once (out-of-image granules are never written afterwards) ----
    {
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int e = tid * 4; e < a_floats; e += 1024) {
            *reinterpret_cast<f32x4*>(smem + e) = zero;
            *reinterpret_cast<f32x4*>(smem + buf_floats + e) = zero;
        }
    }
    __syncthreads();
    stage(smem, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int c0 = 0; c0 < p.Cpad; c0 += CK) {
        float* curA = smem + cur * buf_floats;
        float* nxtA = smem + (cur ^ 1) * buf_floats;
        if (c0 + CK < p.Cpad) stage(nxtA, c0 + CK);
        if (!(p.dbg & 4)) compute(curA, curA + a_floats);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    if (p.epi_lds) {   // launch-uniform
        const StoreDst d{p.out, p.out_bstride, p.outH, p.outW, p.OC, p.TWB, p.act, p.slope, p.out_absmax};
        store_tile_lds<NB, MW>(d, acc, n, nt, oy0, ox0, wave, li, lk, smem, tid);
    } else {
        store_tile<NB, MW>(p, acc, n, nt, oy0, ox0, wave, li, lk);
    }
}

// ----------------------------------------------------------------- host dispatch (one tile width)
template <int KS, int STRIDE, int CK, int NB, int MW, int TWB, bool SYN>
static int dma_variant(ConvParams& p, hipStream_t stream) {
    using G = DmaGeom<KS, STRIDE, MW, TWB>;
    if constexpr (TWB > 4 * MW) {
        return KBN_ERR_UNSUPPORTED;
    } else {
        auto kern = conv_dma_kernel<KS, STRIDE, CK, NB, MW, TWB, SYN>;
        constexpr size_t lds = 2 * sizeof(float) * ((size_t)CK * G::PLANE + (size_t)CK * KS * KS * NB * 16);
        if (lds > 160 * 1024) return KBN_ERR_UNSUPPORTED;
        if (lds < sizeof(float) * 16 * (64 * MW + 4)) p.epi_lds = 0;   // the LDS epilogue's scratch would not fit
        static DeviceOnce once;
        if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
        p.TWB = TWB; p.TH = G::TH;
        p.tilesX = ceil_div(p.outW, G::TW);
        p.tilesY = ceil_div(p.outH, G::TH);
        p.rowsS = G::ROWS; p.colsS = G::COLS; p.pitch = G::COLS; p.PH = 0; p.plane = G::PLANE;
        const long long nb64 = (long long)p.tilesX * p.tilesY * p.N * p.nTilesN;
        if (nb64 > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
        p.nblocks = (int)nb64;
        hipLaunchKernelGGL(kern, dim3(p.nblocks), dim3(256), lds, stream, p);
        KBN_CHECK_LAUNCH();
        return KBN_OK;
    }
}

template <int KS, int STRIDE, int CK, int NB, int TWB, bool SYN>
static int dma_mw(ConvParams& p, int MW, hipStream_t st) {
    if constexpr (NB >= 3) {
        switch (MW) {
            case 1: return dma_variant<KS, STRIDE, CK, NB, 1, TWB, SYN>(p, st);
            case 2: return dma_variant<KS, STRIDE, CK, NB, 2, TWB, SYN>(p, st);
            default: return dma_variant<KS, STRIDE, CK, NB, 4, TWB, SYN>(p, st);
        }
    } else {
        switch (MW) {
            case 1: return dma_variant<KS, STRIDE, CK, NB, 1, TWB, SYN>(p, st);
            case 2: return dma_variant<KS, STRIDE, CK, NB, 2, TWB, SYN>(p, st);
            case 4: return dma_variant<KS, STRIDE, CK, NB, 4, TWB, SYN>(p, st);
            default: return dma_variant<KS, STRIDE, CK, NB, 8, TWB, SYN>(p, st);
        }
    }
}

template <int KS, int STRIDE, int CK, int TWB, bool SYN>
static int dma_nb_syn(ConvParams& p, int NB, int MW, hipStream_t st) {
    switch (NB) {
        case 1: return dma_mw<KS, STRIDE, CK, 1, TWB, SYN>(p, MW, st);
        case 2: return dma_mw<KS, STRIDE, CK, 2, TWB, SYN>(p, MW, st);
        case 3: return dma_mw<KS, STRIDE, CK, 3, TWB, SYN>(p, MW, st);
        default: return dma_mw<KS, STRIDE, CK, 4, TWB, SYN>(p, MW, st);
    }
}

template <int KS, int STRIDE, int CK, int TWB>
static int dma_nb(ConvParams& p, int NB, int MW, hipStream_t st, bool syn) {
    if (syn) return dma_nb_syn<KS, STRIDE, CK, TWB, true>(p, NB, MW, st);
    return dma_nb_syn<KS, STRIDE, CK, TWB, false>(p, NB, MW, st);
}

template <int TWB>
int conv_dma_launch_twb(ConvParams& p, const ConvPlan& pl, int MW, int kernel_size, int stride, bool syn,
                        hipStream_t stream) {
    if (kernel_size == 3 && stride == 1)
        return pl.CK == 4 ? dma_nb<3, 1, 4, TWB>(p, pl.NB, MW, stream, syn) : dma_nb<3, 1, 8, TWB>(p, pl.NB, MW, stream, syn);
    if (kernel_size == 3 && stride == 2)
        return pl.CK == 4 ? dma_nb<3, 2, 4, TWB>(p, pl.NB, MW, stream, syn) : dma_nb<3, 2, 8, TWB>(p, pl.NB, MW, stream, syn);
    if (stride == 2) return dma_nb<1, 2, 16, TWB>(p, pl.NB, MW, stream, syn);
    return dma_nb<1, 1, 16, TWB>(p, pl.NB, MW, stream, syn);
}

}  // namespace kbn
