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
//     16 packed-fp32 adds (v_pk_add_f32), 16 ds_write into V[xi][c][tile] (tile index xor-swizzled by
//     c&1 so that the MFMA A fragments of k and k+1 fall on disjoint banks without padding);
//   * the MFMAs of chunk c from V and U (both double buffered).
// The chunk body is ONE basic block with a fixed interleave (static_for + sched_barrier(0)): every MFMA
// is followed by a slot holding a piece of the other two streams, because vector-ALU work that is not
// interleaved costs matrix-pipe time (tools/probe/issue_probe.hip); the last MFMA group of a chunk is
// issued after the NEXT chunk's barrier, where it covers the first fragment reads; DMA completion is
// counted (vmcnt(2) / vmcnt(6)), not drained; everything wave-uniform lives in SGPRs.
// Epilogue: the accumulators of the 16 frequencies meet in LDS (one pass per 16 output channels, bare
// s_barrier), each thread applies A^T . A to two (tile, channel) pairs, fuses LeakyReLU and stores
// 2 x float2.  Workgroups are persistent (one per CU -- the LDS footprint allows no more): the first
// DMAs of a workgroup's next tile are issued before the current tile's epilogue.
//
// Weights: U = G g G^T is computed once (fp64, rounded to fp32) by wino_pack_kernel and lives behind
// the direct-conv fragment-order weights in the caller's packed blob (kbn_conv2d_pack_weight), laid
// out [n-tile][chunk][xi][c/4][k>>1][n][k&1] so that a chunk is one contiguous 32 KiB DMA.
#include <stdlib.h>

#include <utility>

#include "conv_common.h"

namespace kbn {

namespace {

constexpr int WCK = 8;                      // channels per chunk
constexpr int WNT = 64;                     // output channels per workgroup
constexpr int U_CHUNK = 16 * WCK * WNT;     // floats of U per chunk
constexpr int V_CHUNK = 16 * WCK * 64;      // floats of V per chunk
constexpr int M_NSTRIDE = 68;               // epilogue exchange buffer: [xi][16 n][64 tiles + 4]

template <int I> struct IC { static constexpr int value = I; };
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float wino_max(float a, float b) {   // plain v_max_f32 (fmaxf adds a canonicalising v_max)
    float o;
    asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}

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

template <int DBG>  // compile-time ablation mask (0 in production; see WinoParams::dbg)
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

    // Persistent workgroups: one per CU (the LDS footprint allows no more), each walks the tile list
    // t = blockIdx.x, blockIdx.x + gridDim.x, ...  While tile t's outputs are being stored, the first DMAs
    // of tile t + gridDim.x are already in flight -- with one workgroup per CU nothing else would cover
    // the start-up latency of a tile.
    int nt = 0, n = 0, y0 = 0, x0 = 0;           // current STAGING tile (wave-uniform)
    unsigned gv0 = 0, gv1 = 0;                    // this lane's raw granules: byte offset inside a channel plane ...
    unsigned long long gm0 = 0, gm1 = 0;          // ... and the lanes whose granule is inside the image
    const int cv4 = p.colsS >> 2, nf4 = p.rowsS * cv4;
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

    const int HW = p.H * p.W;
    const int nch = p.Cin / WCK;

    // ---- wave-uniform state of the staging streams, kept in SGPRs (VALU instructions between MFMAs
    // are not free -- tools/probe/issue_probe.hip -- so nothing uniform may end up on the vector ALU) ----
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const unsigned rdst0 = lds0 + 4u * (unsigned)(wave * p.plane), rdst1 = rdst0 + 4u * (unsigned)raw_floats;
    const unsigned udst0 = lds0 + 4u * (unsigned)(2 * raw_floats + wave * 256), udst1 = udst0 + 4u * U_CHUNK;
    const float* usrc = nullptr;   // U of the chunk to stage next
    // Plane of channel (gch + wave) of the concatenated input (every source holds a multiple of 8
    // channels, so a chunk never straddles two); `advance_src` moves to the next chunk and stays on the
    // last one at the end (the surplus DMAs of the last iterations re-stage it into a dead buffer).
    const float* s1w = nullptr;
    const float* s2w = nullptr;
    const float* gsrc = nullptr;
    const int C0 = __builtin_amdgcn_readfirstlane(p.srcC[0]);
    const int C01 = __builtin_amdgcn_readfirstlane(p.srcC[0] + p.srcC[1]);
    const int Cin = __builtin_amdgcn_readfirstlane(p.Cin);
    int gch = 0;
    auto advance_src = [&]() {
        const int nx = gch + WCK;
        const bool adv = nx < Cin;
        const float* cand = nx == C0 ? s1w : (nx == C01 ? s2w : gsrc + (long long)WCK * HW);
        gsrc = uniform_ptr(adv ? cand : gsrc);   // (pinned: the compiler does not always prove the select chain uniform)
        gch = adv ? nx : gch;
    };
    const unsigned uv = (unsigned)(lane * 16);
    int uleft = 0;                                 // further advances before usrc reaches the last chunk

    // one DMA instruction of the raw tile (j = 0, 1) / of the U slice (j = 0..3) per call
    auto dma_raw = [&](unsigned dst, int j) {
        if ((DBG & 1) != 0) return;
        lds_dma16_sm(gsrc, j ? gv1 : gv0, dst + j * 1024, j ? gm1 : gm0);
    };
    auto dma_u = [&](unsigned dst, int j) {
        if ((DBG & 2) != 0) return;
        lds_dma16_s(usrc + j * 2048, uv, dst + j * 8192);
    };

    // Input transform of (channel = wave, tile = lane): d = raw 4x4 patch, t = d B, V = B^T t, on packed
    // fp32 (v_pk_add_f32: 16 VALU instructions instead of 32).  Register pairs hold two adjacent COLUMNS
    // of a row -- what one ds_read2_b32 delivers: P[r][h] = (d[r][2h], d[r][2h+1]), same for T.
    f32x2 P[4][2], T[4][2];
    auto xf_read = [&](const float* raw, int e) {   // e = 4 * row + column
        const float v = raw[(e >> 2) * p.colsS + (e & 3)];
        if (e & 1) P[e >> 2][(e >> 1) & 1].y = v; else P[e >> 2][(e >> 1) & 1].x = v;
    };
    auto xf_col = [&](int o) {                      // o = 2 * row + half: (t0,t1) = (d0-d2, d1+d2) or (t2,t3) = (d2-d1, d1-d3)
        const int r = o >> 1;
        if ((o & 1) == 0)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(T[r][0]) : "v"(P[r][0]), "v"(P[r][1]));
        else
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(T[r][1]) : "v"(P[r][0]), "v"(P[r][1]));
    };
    auto xf_row = [&](float* V, int o) {            // o = 2 * i + h: frequencies (i, 2h) and (i, 2h+1)
        const int i = o >> 1, h = o & 1;
        f32x2 v;
        if (i == 0) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v) : "v"(T[0][h]), "v"(T[2][h]));
        else if (i == 1) asm("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(T[1][h]), "v"(T[2][h]));
        else if (i == 2) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v) : "v"(T[2][h]), "v"(T[1][h]));
        else asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v) : "v"(T[1][h]), "v"(T[3][h]));
        const int xi = i * 4 + 2 * h;
        V[xi * 512] = v.x;
        V[(xi + 1) * 512] = v.y;
    };
    auto transform = [&](const float* raw, float* V) {   // un-pipelined form (prologue)
        if ((DBG & 16) != 0) return;
#pragma unroll
        for (int e = 0; e < 16; ++e) xf_read(raw + raw_off, e);
#pragma unroll
        for (int o = 0; o < 8; ++o) xf_col(o);
#pragma unroll
        for (int o = 0; o < 8; ++o) xf_row(V + v_off, o);
    };

    // MFMA fragments: group g = (frequency x = g >> 1, k-step c4 = g & 1), 4 A + 4 B values, 16 MFMAs
    float fa[2][4], fb[2][4];
    auto frag_read = [&](const float* V, const float* U, int g, int q) {
        if ((DBG & 4) != 0) return;
        if (q < 4) fa[g & 1][q] = V[a_off + (g >> 1) * 512 + (g & 1) * 256 + a_mb[q]];
        else fb[g & 1][q - 4] = U[b_off + (g >> 1) * 512 + (g & 1) * 256 + (q - 4) * 32];
    };
    // MFMA slot i of an iteration runs group (i/16 + 3) % 4: the LAST group of the previous chunk first --
    // its fragments are already in registers, so the matrix pipe has work while the first fragment reads
    // after the chunk's barrier are still in flight -- then groups 0..2 of this chunk.
    // `first` (the first chunk of a tile): the 16 slots of "group 3 of the previous chunk" stay empty and groups 0 and 2 --
    // the first products of their accumulators -- take a zero C operand, so that no accumulator is ever cleared
    // by hand (128 v_mov per tile that would sit on the matrix pipe's issue port).
    auto mfma = [&](int i, bool first = false) {
        if ((DBG & 4) != 0) return;
        const int g = ((i >> 4) + 3) & 3, mb = (i >> 2) & 3, nb = i & 3;
        if (first && i < 16) return;
        if (first && (g & 1) == 0)
            acc[g >> 1][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[g & 1][mb], fb[g & 1][nb], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else
            acc[g >> 1][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[g & 1][mb], fb[g & 1][nb], acc[g >> 1][mb][nb], 0, 0, 0);
    };

    // ---- per tile: decode, staging state, first DMAs ----
    auto setup_tile = [&](int t) {
        int bid = xcd_remap(t, p.nblocks);
        nt = bid % p.nTilesN;
        bid /= p.nTilesN;
        const int rx = bid % p.regionsX;
        bid /= p.regionsX;
        const int ry = bid % p.regionsY;
        n = bid / p.regionsY;
        y0 = ry * 2 * p.RT; x0 = rx * 2 * p.CT;      // first output pixel of the region
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
        gm0 = __ballot(goff[0] >= 0); gm1 = (64 < nf4) ? __ballot(goff[1] >= 0) : 0ull;
        gv0 = goff[0] < 0 ? 0u : (unsigned)goff[0]; gv1 = goff[1] < 0 ? 0u : (unsigned)goff[1];
        usrc = uniform_ptr(p.up + (long long)nt * nch * U_CHUNK + wave * 256);
        s1w = uniform_ptr(p.src[1] + (long long)n * p.src_bstride[1] + (long long)wave * HW);
        s2w = uniform_ptr(p.src[2] + (long long)n * p.src_bstride[2] + (long long)wave * HW);
        gsrc = uniform_ptr(p.src[0] + (long long)n * p.src_bstride[0] + (long long)wave * HW);
        gch = 0;
    };
    // Wave w clears ITS two raw planes (out-of-image granules are never written by the DMAs and must read
    // as zero padding; the previous tile may have had another mask), then issues raw(0), U(0), raw(1).
    // Only the issuing wave ever touches these planes, U stage 0 was last read a whole chunk ago.
    auto stage_first = [&]() {
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int e = lane * 4; e < p.plane; e += 256) {
            *reinterpret_cast<f32x4*>(rawS + wave * p.plane + e) = zero;
            *reinterpret_cast<f32x4*>(rawS + raw_floats + wave * p.plane + e) = zero;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_raw(rdst0, 0); dma_raw(rdst0, 1);                // raw(0)
        advance_src();
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_u(udst0, j);         // U(0)
        usrc += U_CHUNK;                                     // -> U(1)
        uleft = nch - 2;
        dma_raw(rdst1, 0); dma_raw(rdst1, 1);                // raw(1)
        advance_src();
    };

    int tile = blockIdx.x;
    setup_tile(tile);
    stage_first();
  for (;;) {   // ---- tile loop ----
    // output addressing of THIS tile (the staging state above moves on to the next tile before the epilogue)
    const int o_nt = nt;
    float* const outn = p.out + (long long)n * p.out_bstride;
    const int oy = y0 + 2 * ty, ox = x0 + 2 * tx;
    const bool in0 = lane < ntile && ox < p.W && oy < p.H, in1 = in0 && oy + 1 < p.H;
    const unsigned o0 = (unsigned)(oy * p.W + ox), o1 = o0 + (unsigned)p.W;
    if ((DBG & 4) != 0) {   // ablation without MFMAs: defined accumulators for the epilogue
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[x][mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // first DMAs landed (and the previous tile's stores are out)
    transform(rawS, Vs);

    // ---- K loop.  One basic block per chunk with a fixed instruction interleave: all 8 waves run in
    // step, and an in-order wave only keeps the matrix pipe fed if its LDS / VALU / DMA work sits
    // BETWEEN its MFMAs.  Slot i follows MFMA i of chunk c and carries a piece of: the fragment reads of
    // the next MFMA group, the DMA of U(c+1) and raw(c+2), the input transform of chunk c+1.
    // sched_barrier(0) pins the order; buffer parity is a template argument so that every LDS address
    // is a loop-invariant register plus an immediate.  DMAs in flight are counted, not drained:
    // at the top only U(c) must have landed (vmcnt(2): raw(c+1) may still fly), raw(c+1) is awaited
    // (vmcnt(6): U(c+1), raw(c+2) behind it) just before the transform reads it -- a full iteration
    // plus 24 MFMA slots after it was issued, which covers the loaded HBM latency.
    auto body = [&](auto par, auto first_c) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool FIRST = decltype(first_c)::value != 0;
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __syncthreads();
        const float* Vc = Vs + PAR * V_CHUNK;
        const float* Uc = Us + PAR * U_CHUNK;
        float* Vn = Vs + (PAR ^ 1) * V_CHUNK + v_off;
        const float* rawn = rawS + (PAR ^ 1) * raw_floats + raw_off;
        const unsigned rdst = PAR ? rdst1 : rdst0, udst = PAR ? udst0 : udst1;
#pragma unroll
        for (int q = 0; q < 8; ++q) frag_read(Vc, Uc, 0, q);
        __builtin_amdgcn_sched_barrier(0);
        static_for<64>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            mfma(i, FIRST);
            if constexpr (i < 4) dma_u(udst, i);
            if constexpr (i == 4 || i == 5) dma_raw(rdst, i - 4);
            if constexpr (i >= 16 && i < 24) frag_read(Vc, Uc, 1, i - 16);   // (its registers were group 3's until slot 15)
            if constexpr (i == 23) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            if constexpr (i >= 24 && i < 32) {
                if ((DBG & 16) == 0) { xf_read(rawn, 2 * (i - 24)); xf_read(rawn, 2 * (i - 24) + 1); }
            }
            if constexpr (i >= 32 && i < 40) frag_read(Vc, Uc, 2, i - 32);
            if constexpr (i >= 50 && i < 58) frag_read(Vc, Uc, 3, i - 50);   // consumed at the top of the next iteration
            if constexpr (i >= 34 && i < 42) {
                if ((DBG & 16) == 0) xf_col(i - 34);
            }
            if constexpr (i >= 42 && i < 50) {
                if ((DBG & 16) == 0) xf_row(Vn, i - 42);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if ((DBG & 8) == 0) {                        // (ablation 8: always re-stage the same, cache-hot chunk)
            usrc = uleft > 0 ? usrc + U_CHUNK : usrc;   // stays on the last chunk at the end, like gsrc
            --uleft;
            advance_src();
        }
    };
    // nch is even (wino_plan): every chunk goes through `body`; the last one stages and transforms a
    // surplus copy of itself into dead buffers, which keeps the loop free of special cases.
    body(IC<0>{}, IC<1>{});          // first chunk: zero-C products (nch >= 4, wino_plan)
    body(IC<1>{}, IC<0>{});
    for (int c = 2; c < nch; c += 2) {
        body(IC<0>{}, IC<0>{});
        body(IC<1>{}, IC<0>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<16>([&](auto ic) { mfma(decltype(ic)::value); });   // group 3 of the last chunk
    __syncthreads();

    // every wave is past its last LDS read of this tile: start the next tile's first DMAs (raw stages, U
    // stage 0), then run this tile's epilogue in the rest of the LDS (U stage 1 | V stages)
    const int next_tile = tile + (int)gridDim.x;
    const bool more = next_tile < p.nblocks;
    if (more) {
        setup_tile(next_tile);
        stage_first();
    }
    if ((DBG & 32) != 0) {  // ablation: no epilogue (every accumulator stays live)
        f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) sum += acc[x][mb][nb];
        if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.f) p.out[0] = 0.f;
    } else {
    // ---- output transform: 16 frequencies meet in LDS, one pass per 16 output channels ----
    // Bare s_barrier behind an lgkmcnt wait: __syncthreads() would also drain vmcnt, i.e. wait for the
    // previous pass's global stores (and the next tile's DMAs) at every pass.  Addressing is hoisted: a
    // lane's pixel offset is one 32-bit value, the channel plane a wave-uniform base.
    float* const Ms = Us + U_CHUNK;              // [16 xi][16 n][M_NSTRIDE], 68 KiB of the 96 KiB U1 | V0 | V1
    const float slope = p.act ? p.slope : 1.f;   // act off == slope 1
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                *reinterpret_cast<f32x4*>(Ms + ((2 * wave + x) * 16 + li) * M_NSTRIDE + mb * 16 + 4 * lk) = acc[x][mb][nb];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int nloc = wave + 8 * k;
            const int oc = o_nt * WNT + nb * 16 + nloc;    // wave-uniform
            if (oc >= p.OC) continue;
            const float* m = Ms + nloc * M_NSTRIDE + lane;
            float s[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m0 = m[(0 * 4 + j) * 16 * M_NSTRIDE], m1 = m[(1 * 4 + j) * 16 * M_NSTRIDE];
                const float m2 = m[(2 * 4 + j) * 16 * M_NSTRIDE], m3 = m[(3 * 4 + j) * 16 * M_NSTRIDE];
                s[0][j] = m0 + m1 + m2;
                s[1][j] = m1 - m2 - m3;
            }
            float* const oplane = outn + (long long)oc * HW;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float v0 = s[a][0] + s[a][1] + s[a][2];
                float v1 = s[a][1] - s[a][2] - s[a][3];
                v0 = wino_max(v0, v0 * slope);   // LeakyReLU for 0 <= slope <= 1 (the launcher checks; 1 = no activation)
                v1 = wino_max(v1, v1 * slope);
                if ((DBG & 64) != 0) {   // ablation: epilogue without its global stores
                    if (v0 + v1 == 12345.f) oplane[0] = 0.f;
                } else if (a ? in1 : in0) {
                    *reinterpret_cast<float2*>(oplane + (a ? o1 : o0)) = make_float2(v0, v1);  // W, ox even; 8-byte aligned (launcher)
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // Ms is rewritten by the next pass / the next tile's V
    }
    }
    if (!more) break;
    tile = next_tile;
  }   // tile loop
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
// (every shape's staged tile, (2 RT + 2) x round_up(2 CT + 5, 4) floats per channel, fits the 512-float plane)
static const int kRegions[][2] = {{4, 16}, {8, 8}, {2, 32}, {6, 10}, {5, 12}, {3, 20}, {7, 8}, {10, 6}};
constexpr int kNumRegions = 8;

static int choose_region(int H, int W, int n, int nTilesN) {   // index into kRegions
    const int th = ceil_div(H, 2), tw = ceil_div(W, 2);
    long long best_rounds = 1LL << 60, best_wgs = 1LL << 60;
    int best = 0;
    const int force_rt = knob(KNOB_WINO_RT);
    for (int i = 0; i < kNumRegions; ++i) {
        const int* c = kRegions[i];
        if (force_rt && c[0] != force_rt) continue;
        const long long wgs = (long long)ceil_div(th, c[0]) * ceil_div(tw, c[1]) * n * nTilesN;
        const long long rounds = (wgs + 255) / 256;
        if (rounds < best_rounds || (rounds == best_rounds && wgs < best_wgs)) {
            best_rounds = rounds; best_wgs = wgs; best = i;
        }
    }
    return best;
}

template <int DBG>
static int wino_variant(const WinoParams& p, size_t lds, hipStream_t stream) {
    static DeviceOnce once;
    if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(conv_wino_kernel<DBG>), 160 * 1024)) return rc;
    int n_cu = device_cu_count();   // persistent workgroups: one per CU of the current device
    if (n_cu < 1) n_cu = 256;
    const int grid = p.nblocks < n_cu ? p.nblocks : n_cu;
    hipLaunchKernelGGL(conv_wino_kernel<DBG>, dim3(grid), dim3(512), lds, stream, p);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

static int wino_dispatch(const WinoParams& p, size_t lds, hipStream_t stream);

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
    p.act = cp.act; p.slope = cp.slope; p.dbg = cp.dbg;
    p.vec_ok = ((reinterpret_cast<uintptr_t>(cp.out) & 7) == 0) && ((cp.out_bstride & 1) == 0) && ((p.W & 1) == 0);
    if (!p.vec_ok) return KBN_ERR_UNSUPPORTED;  // the epilogue stores float2
    if (p.act && !(p.slope >= 0.f && p.slope <= 1.f)) return KBN_ERR_UNSUPPORTED;   // the epilogue's max(v, slope v) form
    const int model = choose_region(p.H, p.W, p.N, p.nTilesN);
    auto launch = [&](int cand) -> int {
        WinoParams q = p;
        q.RT = kRegions[cand][0]; q.CT = kRegions[cand][1];
        q.regionsX = ceil_div(ceil_div(q.W, 2), q.CT);
        q.regionsY = ceil_div(ceil_div(q.H, 2), q.RT);
        q.rowsS = 2 * q.RT + 2;
        q.colsS = round_up(2 * q.CT + 5, 4);
        q.plane = q.rowsS * q.colsS;  // multiple of 4
        if (q.plane > 512) return KBN_ERR_UNSUPPORTED;
        const long long nb64 = (long long)q.regionsX * q.regionsY * q.N * q.nTilesN;
        if (nb64 > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
        q.nblocks = (int)nb64;
        size_t lds = sizeof(float) * ((size_t)2 * WCK * q.plane + 2 * U_CHUNK + 2 * V_CHUNK);
        const size_t lds_epi = sizeof(float) * ((size_t)2 * WCK * q.plane + U_CHUNK + (size_t)16 * 16 * M_NSTRIDE);
        if (lds < lds_epi) lds = lds_epi;
        if (lds > 160 * 1024) return KBN_ERR_UNSUPPORTED;
        return wino_dispatch(q, lds, stream);
    };
    int cand = model;
    if (!knob(KNOB_WINO_RT) && !p.dbg) {
        cand = tune_pick(TuneKey{2, p.N, p.OC, p.Cin, p.H, p.W, 0, 0, 0, 0}, kNumRegions, model, launch, stream);
    }
    return launch(cand);
}

static int wino_dispatch(const WinoParams& p, size_t lds, hipStream_t stream) {
    switch (p.dbg) {  // ablation builds of the same kernel (tools/conv_bench.py --dbg)
        case 1: return wino_variant<1>(p, lds, stream);
        case 2: return wino_variant<2>(p, lds, stream);
        case 3: return wino_variant<3>(p, lds, stream);
        case 4: return wino_variant<4>(p, lds, stream);
        case 8: return wino_variant<8>(p, lds, stream);
        case 16: return wino_variant<16>(p, lds, stream);
        case 19: return wino_variant<19>(p, lds, stream);
        case 23: return wino_variant<23>(p, lds, stream);
        case 32: return wino_variant<32>(p, lds, stream);
        case 55: return wino_variant<55>(p, lds, stream);
        case 64: return wino_variant<64>(p, lds, stream);
        default: return wino_variant<0>(p, lds, stream);
    }
}
int wino_query(int n, int oc, int cin, int H, int W, int* RT, int* CT) {
    const WinoPlan wp = wino_plan(oc, cin, 3, 1);
    if (!wp.ok || (W & 3)) return 0;
    int cand = choose_region(H, W, n, wp.nTilesN);
    tune_lookup(TuneKey{2, n, oc, cin, H, W, 0, 0, 0, 0}, &cand, kNumRegions);   // a tuned choice, if this shape has run already
    *RT = kRegions[cand][0]; *CT = kRegions[cand][1];
    return ceil_div(ceil_div(W, 2), *CT) * ceil_div(ceil_div(H, 2), *RT) * n * wp.nTilesN;
}

}  // namespace kbn
