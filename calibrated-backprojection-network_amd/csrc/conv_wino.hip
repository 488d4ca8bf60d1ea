// conv_wino.hip -- 3x3 stride-1 convs with wide channel counts (the second conv of every decoder
// block, reference src/net_utils.py:1485-1487 via net_utils.Conv2d :120-141) evaluated with the
// Winograd F(2x2, 3x3) minimal-filtering algorithm in fp32:
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per 2x2 output tile, 4x4 input tile d_c
//
// i.e. 16 element-wise products per (tile, channel pair) instead of 36 MACs: 2.25x fewer MFMAs.
// The 16 "frequencies" xi = (i, j) are 16 independent GEMMs  M_xi[tile][oc] = V_xi[tile][c] U_xi[c][oc]
// that run on v_mfma_f32_16x16x4_f32 exactly like the direct kernels (conv_dma.hip).  All arithmetic
// stays fp32; on the benchmark frame the result is as close to an fp64 evaluation as the CPU
// oracle's own fp32 convs are (tools/winograd_error.py: 2.0e-5 vs 2.1e-5 max relative).
//
// Workgroup = 512 threads (8 waves, two per SIMD) = one region of RT x CT tiles (<= 64 tiles, i.e.
// up to 256 output pixels) x 64 output channels.  Wave w owns frequencies 2w and 2w+1 for the whole
// 64 x 64 (tile x channel) block: 2 x 4 x 4 accumulator tiles = 128 VGPRs.
//
// K loop over chunks of 8 input channels, one barrier per chunk, three streams in flight:
//   * LDS-DMA (global_load_lds_dwordx4, conv_common.h) brings the RAW input tile of chunk c+2 and the
//     pre-transformed weights U of chunk c+1 into LDS.  Wave w stages channel w of the chunk, the
//     same channel it transforms, so the raw tile needs no barrier of its own.
//   * the input transform B^T d B of chunk c+1: thread = (channel = wave, tile = lane); 16 ds_read,
//     32 adds, 16 ds_write into V[xi][c][tile] (tile index xor-swizzled by c&1 so that the MFMA A
//     fragments of k and k+1 fall on disjoint banks without padding);
//   * the MFMAs of chunk c from V and U (both double buffered).
// Epilogue: the accumulators of the 16 frequencies meet in LDS (one pass per 16 output channels),
// each thread applies A^T . A to two (tile, channel) pairs, fuses LeakyReLU and stores 2 x float2.
//
// Weights: U = G g G^T is computed once (fp64, rounded to fp32) by wino_pack_kernel and lives behind
// the direct-conv fragment-order weights in the caller's packed blob (kbn_conv2d_pack_weight), laid
// out [n-tile][chunk][xi][c/4][k>>1][n][k&1] so that a chunk is one contiguous 32 KiB DMA.
#include <stdlib.h>

#include "conv_common.h"

namespace kbn {

namespace {

constexpr int WCK = 8;                      // channels per chunk
constexpr int WNT = 64;                     // output channels per workgroup
constexpr int U_CHUNK = 16 * WCK * WNT;     // floats of U per chunk
constexpr int V_CHUNK = 16 * WCK * 64;      // floats of V per chunk
constexpr int M_NSTRIDE = 68;               // epilogue exchange buffer: [xi][16 n][64 tiles + 4]

struct WinoParams {
    const float* src[KBN_MAX_SRC];
    long long src_bstride[KBN_MAX_SRC];
    int srcC[KBN_MAX_SRC];
    int nsrc;
    const float* up;
    float* out;
    long long out_bstride;
    int N, OC, Cin, H, W;
    int RT, CT;                 // tiles per region (rows, columns)
    int regionsX, regionsY, nTilesN, nblocks;
    int rowsS, colsS, plane;    // raw staged tile: rows, columns (multiple of 4), floats per channel
    int act;
    float slope;
    int vec_ok;
    int dbg;  // ablation (KBN_DEBUG, tools/conv_bench.py): 1 no raw staging, 2 no U staging, 4 no MFMA, 16 no transform, 32 no epilogue
};

}  // namespace

__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int OC, int Cin,
                                 long long total) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int nch = Cin / WCK;
    const long long per_nt = (long long)nch * U_CHUNK;
    const int nt = (int)(e / per_nt);
    int r = (int)(e - nt * per_nt);
    const int chunk = r / U_CHUNK; r -= chunk * U_CHUNK;
    const int xi = r / (WCK * WNT); r -= xi * WCK * WNT;
    const int c4 = r / (4 * WNT); r -= c4 * 4 * WNT;
    const int khalf = r / (2 * WNT); r -= khalf * 2 * WNT;
    const int n = r >> 1, klow = r & 1;
    const int c = chunk * WCK + c4 * 4 + khalf * 2 + klow;
    const int oc = nt * WNT + n;
    double u = 0.0;
    if (oc < OC) {
        const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
        const float* g = w + ((long long)oc * Cin + c) * 9;
        const int i = xi >> 2, j = xi & 3;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) u += G[i][a] * (double)g[a * 3 + b] * G[j][b];
    }
    packed[e] = (float)u;
}

template <bool DBG>
__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int raw_floats = WCK * p.plane;          // one raw stage
    float* const rawS = smem;                      // [2][8][plane]
    float* const Us = smem + 2 * raw_floats;       // [2][U_CHUNK]
    float* const Vs = Us + 2 * U_CHUNK;            // [2][V_CHUNK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;

    int bid = xcd_remap(blockIdx.x, p.nblocks);
    const int nt = bid % p.nTilesN;
    bid /= p.nTilesN;
    const int rx = bid % p.regionsX;
    bid /= p.regionsX;
    const int ry = bid % p.regionsY;
    const int n = bid / p.regionsY;
    const int y0 = ry * 2 * p.RT, x0 = rx * 2 * p.CT;  // first output pixel of the region

    // ---- this lane's raw granules (channel-plane byte offsets, -1 = out of image) ----
    const int cv4 = p.colsS >> 2, nf4 = p.rowsS * cv4;
    int goff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = j * 64 + lane;
        int g = -1;
        if (f < nf4) {
            const int r = f / cv4, cv = f - r * cv4;
            const int Y = y0 - 1 + r, X = x0 - 4 + cv * 4;
            if (Y >= 0 && Y < p.H && X >= 0 && X < p.W) g = (Y * p.W + X) * 4;
        }
        goff[j] = g;
    }

    // ---- this lane's tile (input transform + epilogue) ----
    const int ntile = p.RT * p.CT;
    const int tl = lane < ntile ? lane : 0;
    const int ty = tl / p.CT, tx = tl - ty * p.CT;
    const int raw_off = wave * p.plane + 2 * ty * p.colsS + 2 * tx + 3;       // d[0][0] of (channel wave, tile)
    const int v_off = wave * 64 + (lane ^ ((wave & 1) << 4));                 // + xi * 512

    // ---- MFMA fragment addressing (wave owns xi = 2*wave, 2*wave+1) ----
    const int a_off = (2 * wave * WCK + lk) * 64 + li;                        // + x*512 + c4*256 + a_mb[mb]
    int a_mb[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) a_mb[mb] = (mb * 16) ^ ((lk & 1) << 4);    // the writer's xor swizzle
    const int b_off = 2 * wave * WCK * WNT + (lk >> 1) * 2 * WNT + li * 2 + (lk & 1);  // + x*512 + c4*256 + nb*32

    f32x4 acc[2][4][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[x][mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int HW = p.H * p.W;
    const int nch = p.Cin / WCK;
    const float* up_nt = p.up + (long long)nt * nch * U_CHUNK;

    // running source pointer: channel (c0 + wave) of the current source (every source holds a
    // multiple of 8 channels, so a chunk never straddles two)
    int cs_idx = 0, s_left = p.srcC[0];
    const float* wptr = p.src[0] + (long long)n * p.src_bstride[0] + (long long)wave * HW;

    auto stage_raw = [&](float* dstbuf) {
        if (s_left <= 0 && cs_idx + 1 < p.nsrc) {
            ++cs_idx;
            wptr = p.src[cs_idx] + (long long)n * p.src_bstride[cs_idx] + (long long)wave * HW;
            s_left = p.srcC[cs_idx];
        }
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(dstbuf + wave * p.plane));
        if (!(DBG && (p.dbg & 1))) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (j * 64 < nf4 && goff[j] >= 0) lds_dma16_s(wptr, (unsigned)goff[j], dst + j * 1024);
        }
        wptr += (long long)WCK * HW;
        s_left -= WCK;
    };
    auto stage_u = [&](float* dstbuf, int chunk) {
        if (DBG && (p.dbg & 2)) return;
        const float* s = up_nt + (long long)chunk * U_CHUNK + wave * 256;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(dstbuf + wave * 256));
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16_s(s + j * 2048, (unsigned)(lane * 16), dst + j * 8192);
    };
    auto transform = [&](const float* raw, float* V) {
        if (DBG && (p.dbg & 16)) return;
        const float* d = raw + raw_off;
        float t[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // rows of B^T d: (d0-d2, d1+d2, d2-d1, d1-d3) along the row index
            // column pass first on each raw row: t[i][j] = (d_i B)_j
            const float d0 = d[i * p.colsS], d1 = d[i * p.colsS + 1], d2 = d[i * p.colsS + 2], d3 = d[i * p.colsS + 3];
            t[i][0] = d0 - d2; t[i][1] = d1 + d2; t[i][2] = d2 - d1; t[i][3] = d1 - d3;
        }
        float* v = V + v_off;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[(0 * 4 + j) * 512] = t[0][j] - t[2][j];
            v[(1 * 4 + j) * 512] = t[1][j] + t[2][j];
            v[(2 * 4 + j) * 512] = t[2][j] - t[1][j];
            v[(3 * 4 + j) * 512] = t[1][j] - t[3][j];
        }
    };
    // MFMAs of one chunk in 4 groups (frequency x, k-step c4) of 16.  The fragments of group g+1 are
    // read from LDS before the MFMAs of group g are issued (pinned with sched_barrier: all 8 waves run
    // in step, so an exposed ds_read latency is paid by the whole CU); group 0 is read by `frags0`
    // right after the chunk's barrier, ahead of the transform.
    float fa[2][4], fb[2][4];
    auto load_frags = [&](const float* V, const float* U, int g, float (&a)[4], float (&b)[4]) {
        const float* Ab = V + a_off + (g >> 1) * 512 + (g & 1) * 256;
        const float* Bb = U + b_off + (g >> 1) * 512 + (g & 1) * 256;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) a[mb] = Ab[a_mb[mb]];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) b[nb] = Bb[nb * 32];
    };
    auto frags0 = [&](const float* V, const float* U) {
        if (DBG && (p.dbg & 4)) return;
        load_frags(V, U, 0, fa[0], fb[0]);
    };
    auto compute = [&](const float* V, const float* U) {
        if (DBG && (p.dbg & 4)) return;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) load_frags(V, U, g + 1, fa[(g + 1) & 1], fb[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc[g >> 1][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[g & 1][mb], fb[g & 1][nb], acc[g >> 1][mb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- clear both raw stages once (out-of-image granules are never written afterwards) ----
    {
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int e = tid * 4; e < 2 * raw_floats; e += 2048) *reinterpret_cast<f32x4*>(rawS + e) = zero;
    }
    __syncthreads();
    stage_raw(rawS);
    stage_u(Us, 0);
    if (nch > 1) stage_raw(rawS + raw_floats);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    transform(rawS, Vs);

    for (int c = 0; c + 1 < nch; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stage_u(Us + ((c + 1) & 1) * U_CHUNK, c + 1);
        if (c + 2 < nch) stage_raw(rawS + (c & 1) * raw_floats);
        // The two waves of a SIMD (w and w+4) run the two halves of the iteration in opposite order:
        // one transforms chunk c+1 (LDS + VALU) while the other keeps the matrix pipe busy with chunk c.
        frags0(Vs + (c & 1) * V_CHUNK, Us + (c & 1) * U_CHUNK);
        __builtin_amdgcn_sched_barrier(0);
        if (wave < 4) transform(rawS + ((c + 1) & 1) * raw_floats, Vs + ((c + 1) & 1) * V_CHUNK);
        __builtin_amdgcn_sched_barrier(0);
        compute(Vs + (c & 1) * V_CHUNK, Us + (c & 1) * U_CHUNK);
        __builtin_amdgcn_sched_barrier(0);
        if (wave >= 4) transform(rawS + ((c + 1) & 1) * raw_floats, Vs + ((c + 1) & 1) * V_CHUNK);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    frags0(Vs + ((nch - 1) & 1) * V_CHUNK, Us + ((nch - 1) & 1) * U_CHUNK);
    compute(Vs + ((nch - 1) & 1) * V_CHUNK, Us + ((nch - 1) & 1) * U_CHUNK);
    __syncthreads();

    if (DBG && (p.dbg & 32)) {
        if (acc[0][0][0][0] == 12345.f) p.out[0] = 0.f;  // keep the loop alive
        return;
    }
    // ---- output transform: 16 frequencies meet in LDS, one pass per 16 output channels ----
    float* const Ms = smem;  // [16 xi][16 n][M_NSTRIDE]
    float* outn = p.out + (long long)n * p.out_bstride;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                *reinterpret_cast<f32x4*>(Ms + ((2 * wave + x) * 16 + li) * M_NSTRIDE + mb * 16 + 4 * lk) = acc[x][mb][nb];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int nloc = (tid >> 6) + 8 * k;    // wave-uniform
            const int oc = nt * WNT + nb * 16 + nloc;
            const float* m = Ms + nloc * M_NSTRIDE + lane;
            float s[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m0 = m[(0 * 4 + j) * 16 * M_NSTRIDE], m1 = m[(1 * 4 + j) * 16 * M_NSTRIDE];
                const float m2 = m[(2 * 4 + j) * 16 * M_NSTRIDE], m3 = m[(3 * 4 + j) * 16 * M_NSTRIDE];
                s[0][j] = m0 + m1 + m2;
                s[1][j] = m1 - m2 - m3;
            }
            const int oy = y0 + 2 * ty, ox = x0 + 2 * tx;
            if (lane < ntile && oc < p.OC && ox < p.W) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    if (oy + a >= p.H) continue;
                    float v0 = s[a][0] + s[a][1] + s[a][2];
                    float v1 = s[a][1] - s[a][2] - s[a][3];
                    if (p.act) { v0 = leaky_relu(v0, p.slope); v1 = leaky_relu(v1, p.slope); }
                    float* o = outn + (long long)oc * HW + (long long)(oy + a) * p.W + ox;
                    if (p.vec_ok && ox + 1 < p.W) {
                        *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
                    } else {
                        o[0] = v0;
                        if (ox + 1 < p.W) o[1] = v1;
                    }
                }
            }
        }
        __syncthreads();
    }
}

long long wino_packed_floats(int oc, int cin, int ks, int stride) {
    const WinoPlan wp = wino_plan(oc, cin, ks, stride);
    return wp.ok ? (long long)wp.nTilesN * cin * 16 * WNT : 0;
}

int wino_pack(const float* weight, float* packed, int oc, int cin, hipStream_t stream) {
    const long long total = wino_packed_floats(oc, cin, 3, 1);
    if (total <= 0) return KBN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, weight, packed,
                       oc, cin, total);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

// Region shape: RT x CT tiles (CT even so that regions start on a 4-pixel boundary), chosen to
// minimise the number of 256-workgroup rounds (one workgroup per CU), then the workgroup count.
static void choose_region(int H, int W, int n, int nTilesN, int& RT, int& CT) {
    static const int cand[][2] = {{4, 16}, {8, 8}, {2, 32}, {6, 10}, {5, 12}, {3, 20}, {7, 8}, {10, 6}, {16, 4}};
    const int th = ceil_div(H, 2), tw = ceil_div(W, 2);
    long long best_rounds = 1LL << 60, best_wgs = 1LL << 60;
    const int force_rt = getenv("KBN_WINO_RT") ? atoi(getenv("KBN_WINO_RT")) : 0;
    for (const auto& c : cand) {
        if (force_rt && c[0] != force_rt) continue;
        const long long wgs = (long long)ceil_div(th, c[0]) * ceil_div(tw, c[1]) * n * nTilesN;
        const long long rounds = (wgs + 255) / 256;
        if (rounds < best_rounds || (rounds == best_rounds && wgs < best_wgs)) {
            best_rounds = rounds; best_wgs = wgs; RT = c[0]; CT = c[1];
        }
    }
}

int conv_wino_launch(const ConvParams& cp, hipStream_t stream) {
    const WinoPlan wp = wino_plan(cp.OC, cp.Ctot, 3, 1);
    if (!wp.ok || cp.resize || (cp.inW & 3)) return KBN_ERR_UNSUPPORTED;
    WinoParams p;
    for (int s = 0; s < KBN_MAX_SRC; ++s) { p.src[s] = nullptr; p.src_bstride[s] = 0; p.srcC[s] = 0; }
    for (int s = 0; s < cp.nsrc; ++s) {
        const SrcDev& d = cp.src[s];
        if (d.kind != KBN_SRC_TENSOR || (d.C % WCK) != 0) return KBN_ERR_UNSUPPORTED;
        if ((reinterpret_cast<uintptr_t>(d.data) & 15) || (d.bstride & 3)) return KBN_ERR_UNSUPPORTED;
        p.src[s] = d.data; p.src_bstride[s] = d.bstride; p.srcC[s] = d.C;
    }
    p.nsrc = cp.nsrc;
    const ConvPlan pl = make_plan(cp.OC, cp.Ctot, 3, 1);
    p.up = cp.wp + (long long)pl.nTilesN * pl.Cpad * 9 * pl.NT;  // U follows the direct-conv weights
    p.out = cp.out; p.out_bstride = cp.out_bstride;
    p.N = cp.N; p.OC = cp.OC; p.Cin = cp.Ctot; p.H = cp.inH; p.W = cp.inW;
    p.nTilesN = wp.nTilesN;
    choose_region(p.H, p.W, p.N, p.nTilesN, p.RT, p.CT);
    p.regionsX = ceil_div(ceil_div(p.W, 2), p.CT);
    p.regionsY = ceil_div(ceil_div(p.H, 2), p.RT);
    p.rowsS = 2 * p.RT + 2;
    p.colsS = round_up(2 * p.CT + 5, 4);
    p.plane = p.rowsS * p.colsS;  // multiple of 4
    if (p.plane > 512) return KBN_ERR_UNSUPPORTED;
    const long long nb64 = (long long)p.regionsX * p.regionsY * p.N * p.nTilesN;
    if (nb64 > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.nblocks = (int)nb64;
    p.act = cp.act; p.slope = cp.slope; p.dbg = cp.dbg;
    p.vec_ok = ((reinterpret_cast<uintptr_t>(cp.out) & 7) == 0) && ((cp.out_bstride & 1) == 0) && ((p.W & 1) == 0);
    size_t lds = sizeof(float) * ((size_t)2 * WCK * p.plane + 2 * U_CHUNK + 2 * V_CHUNK);
    const size_t lds_epi = sizeof(float) * (size_t)16 * 16 * M_NSTRIDE;
    if (lds < lds_epi) lds = lds_epi;
    if (lds > 160 * 1024) return KBN_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return KBN_ERR_LAUNCH;
        attr_set = true;
    }
    if (p.dbg)  // ablation build of the same kernel (tools/conv_bench.py --dbg)
        hipLaunchKernelGGL(conv_wino_kernel<true>, dim3(p.nblocks), dim3(512), lds, stream, p);
    else
        hipLaunchKernelGGL(conv_wino_kernel<false>, dim3(p.nblocks), dim3(512), lds, stream, p);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int wino_query(int n, int oc, int cin, int H, int W, int* RT, int* CT) {
    const WinoPlan wp = wino_plan(oc, cin, 3, 1);
    if (!wp.ok || (W & 3)) return 0;
    choose_region(H, W, n, wp.nTilesN, *RT, *CT);
    return ceil_div(ceil_div(W, 2), *CT) * ceil_div(ceil_div(H, 2), *RT) * n * wp.nTilesN;
}

}  // namespace kbn
