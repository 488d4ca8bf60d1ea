// front.hip -- the full-resolution front of the encoder's image branch in ONE kernel, on the 16-bit matrix core:
//     conv0_image = act(conv3x3(image))                                  reference src/networks.py:364-365 (net_utils.Conv2d)
//     conv_image  = act(conv3x3 s2 (conv0_image))                        reference src/net_utils.py:1348   (KB block, level 0)
//     conv_fused  = act(conv1x1 s2 (cat[conv0_image, xyz]))              reference src/net_utils.py:1352-1369
// conv0_image -- 48 channels at full resolution, 82 MB per KITTI frame -- is consumed by these two convs only, so it is
// computed per tile (+ halo) and kept in LDS: it never reaches HBM (round 2 wrote and re-read 5.3 GB of it per 32-frame
// step: conv0_image at 4.1 TB/s + the fused fp32 KB kernel, 2.45 ms of a 12.8 ms step on the fp32 pipe).  All three convs take
// their fp32 products as three fp16 MFMAs over two-term splits of both operands, like csrc/conv_split.hip:
//     a 2^k = h1 + 2^-11 h2,   w 2^e = w1 + 2^-11 w2'   ->   a w 2^(e+k) = h1 w1 + 2^-11 (h1 w2' + h2 w1)
// with fp32 accumulators for the main and the small term.  The windows follow the data inside the kernel, tile by tile: k of
// the image from max |image| over the pixels the workgroup loads; k of the on-chip conv0 output from the BOUND
// max |conv0| <= (max_f sum |w0_f|) max |image| (the true maximum is not known before the values are written; the bound
// is within a few binades of it, and a window may sit 2^16 above the data before anything is lost --
// tests/test_split_math_cpu.py).  Outputs are fp32 NCHW with their absmax slots filled, like every other conv.
//
// Tile = 8 x 16 output pixels (half resolution) of one frame per workgroup of 8 waves; v_mfma_f32_16x16x32_f16 throughout.
//   A  the (2*8+3) x (2*16+3) image pixels the tile's conv0 outputs read: loaded, split, one 16-byte granule [8 channels]
//      per pixel and split term (channels >= c_in are zero).
//   B  conv0, 16 filters at a time: D[filter][pixel], A operand = weights, B operand = pixels; K = (tap, channel): k-group =
//      tap, so a B fragment is ONE ds_read_b128 of the granule at (pixel + tap offset).  The 16 x 16 result blocks hold 4
//      consecutive CHANNELS of a pixel per lane: LeakyReLU, zero outside the image (it is the padding of the next conv),
//      split with the bound's exponent, ds_write_b64 into X [term][k-group][17 x 33 pixels, columns de-interleaved][8 ch].
//   C  conv_image (nine taps, two per MFMA: K = [16 ch of tap t | 16 ch of tap t+1]) and conv_fused (centre tap) over
//      these 16 channels: D[pixel][filter], A operand = pixels of one output row per wave (stride-2 reads of X are
//      consecutive granules thanks to the de-interleave), B operand = weights from LDS (double buffered, LDS-DMA).
//   D  after the last chunk: scales, xyz channels of conv_fused in fp32 (kbn_kb_xyz_s2_forward computes them once per block),
//      LeakyReLU, 16-byte NCHW stores (a lane holds 4 consecutive pixels of a filter), absmax.
#include <initializer_list>
#include <type_traits>

#include "front_common.h"
#include "s2d_stage.h"

namespace kbn {

struct FrontParams {
    const float* image;
    long long image_bstride;
    const float* tab;             // scales / xyz weights (FR_TAB floats)
    const _Float16* w0;           // [chunk][2 k-steps][term][4 k-groups][16 filters][8] -- k-group = (tap row, column pair), 8 = 2 pixels x 4 channels
    const _Float16* wc;           // [chunk]{[5 k-steps][term][4 k-groups][FI filters][8] (tap pairs) | [term][2 k-groups][FI][8] (conv_fused)}
    const float* xyz;             // N x 3 x h x w or null
    long long xyz_bstride;
    float* out_image;
    long long out_image_bstride;
    float* out_fused;
    long long out_fused_bstride;
    unsigned* amax_out_image;
    unsigned* amax_out_fused;
    int N, Cin, H, W, h, w, tilesX, tilesY, ntiles;
    float slope0, slope1;
    int vec4;
    // NEXT (kbn_kb1_front_next_forward): the NEXT KB level's conv_fused in the same launch -- a 1x1 stride-2 conv over
    // cat[conv_image, xyz of that level, conv_fused] reads exactly the even pixels of this tile's two outputs (no halo)
    const float* tab2;            // [2^-e per filter FO][xyz weights FO x 3]
    const _Float16* wn;           // [k-step][term][4 k-groups][FO filters][8]: K = (conv_image channels, conv_fused channels)
    const float* xyz2;            // N x 3 x h2 x w2 (kbn_kb_xyz_s2_forward of the next level)
    long long xyz2_bstride;
    float* out2;                  // N x FO x h2 x w2
    long long out2_bstride;
    unsigned* amax_out2;
    int h2, w2, vec4_2;
    float slope2;
};

// LDS per workgroup: IN 10.5 KB + X 35.1 KB + one chunk of conv_image / conv_fused weights 33 KB = 78.6 KB: TWO workgroups
// per CU, so that one's image loads, barriers and stores hide under the other's MFMAs.
// ONE: the THROUGHPUT-ONLY one-term mode (KBN_FP16_ONE_TERM=1, BASELINE configs[2]'s 16-bit leg): every product is h1 w1 alone -- plain fp16
// operands, fp32 accumulation, a third of the MFMAs, no h2 granules written or read
template <int NC0, int NBI, bool NEXT = false, bool ONE = false>   // conv0 filters / 16, conv_image = conv_fused filters / 16; NEXT: + the next level's conv_fused
__global__ __launch_bounds__(FR_THREADS, NEXT ? 4 : 2) void kb1_front_kernel(const FrontParams p) {
    constexpr int FI = NBI * 16;
    constexpr int IN_PART = FR_NIN * 8, IN_BYTES = 2 * IN_PART;            // [term][pixel][4 channels] fp16
    constexpr int X_KG = FR_XP * 16, X_PART = 2 * X_KG, X_BYTES = 2 * X_PART;
    constexpr int WC_KQ = FI * 16, WC_PART = 4 * WC_KQ, WC_KS = 2 * WC_PART, WC_FUSED = 2 * 2 * WC_KQ, WC_CHUNK = 5 * WC_KS + WC_FUSED;
    constexpr int OFF_X = IN_BYTES, OFF_WC = OFF_X + X_BYTES;
    static_assert(OFF_WC + WC_CHUNK <= 80 * 1024, "two workgroups per CU");
    constexpr int NBLK = (FR_NB0 + 7) / 8;                                  // conv0 pixel blocks per wave: 5 (waves 4-7: 4)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0");   // fp16 results flush subnormals (see conv3x3_split_kernel)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    int bid = xcd_remap(blockIdx.x, p.ntiles);
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int oy0 = ty * FR_TH, ox0 = tx * FR_TW;
    const int H = p.H, W = p.W;
    const long long plane = (long long)H * W;

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(reinterpret_cast<const float*>(smem)));
    auto dma_wc = [&](int chunk) {   // whole-workgroup copy of one chunk's weights, 16 bytes per lane and round
        const float* src = reinterpret_cast<const float*>(p.wc + (long long)chunk * (WC_CHUNK / 2));
        constexpr int n4 = WC_CHUNK / 16;
#pragma unroll
        for (int e0 = 0; e0 < n4; e0 += FR_THREADS) {
            const int eb = e0 + wave * 64;
            if (eb + lane < n4) lds_dma16_s(src + eb * 4, (unsigned)(lane * 16), lds0 + (unsigned)(OFF_WC + eb * 16));
        }
    };
    dma_wc(0);

    // ---- A: image tile -> split entries [4 channels] per pixel (two pixels = one 16-byte K group of conv0).  The fp16
    // windows are the TILE's own: max |image| over the pixels this workgroup reads (one LDS reduction) places the image's, the
    // bound L1max0 * that maximum conv0's -- no pass over the image beforehand, and a tile never pays for a bright spot elsewhere.
    float pre_img, un_img, pre0, un0;
    {
        const float* img = p.image + (long long)n * p.image_bstride;
        ff4 raw[2];
        float tm = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pix = u * FR_THREADS + tid;
            const int r = pix / FR_R0W, c = pix - r * FR_R0W;
            const int Y = 2 * oy0 - 2 + r, X = 2 * ox0 - 2 + c;
            const bool ok = pix < FR_NP0 && Y >= 0 && Y < H && X >= 0 && X < W;   // entries past the tile stay zero (pair reads touch one)
            const float* src = img + (long long)(ok ? Y : 0) * W + (ok ? X : 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                raw[u][j] = (ok && j < p.Cin) ? src[(long long)(j < p.Cin ? j : 0) * plane] : 0.f;
                tm = fmaxf(tm, fabsf(raw[u][j]));
            }
        }
        tm = __uint_as_float(wave_max_bits(tm));
        float* red = reinterpret_cast<float*>(smem + OFF_X);   // X is idle until conv0 writes it (after the next barrier)
        if (lane == 0) red[wave] = tm;
        __syncthreads();
        tm = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
        const unsigned abits = __builtin_amdgcn_readfirstlane(__float_as_uint(tm));
        fr_scales(abits, pre_img, un_img);
        fr_scales(__float_as_uint(p.tab[0] * __uint_as_float(abits)), pre0, un0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pix = u * FR_THREADS + tid;
            if (pix < FR_NIN) {
                fh4 h1, h2;
                fr_split4(raw[u] * pre_img, h1, h2);
                *reinterpret_cast<fh4*>(smem + pix * 8) = h1;
                *reinterpret_cast<fh4*>(smem + IN_PART + pix * 8) = h2;
            }
        }
    }

    // ---- per-lane offsets, computed once per tile
    // conv0: k-group g = 4 ks + kq = (tap row g >> 1, column pair g & 1: columns 0-1 / 2-3, the fourth carries zero weights);
    // groups 6, 7 are all zero weights: any valid address
    int goff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int g = min(4 * ks + kq, 5);
        goff[ks] = ((g >> 1) * FR_R0W + 2 * (g & 1)) * 8;
    }
    // conv0 block j (0..35) of the 17 x 33 region: j < 34: row j >> 1, columns 16 (j & 1) + l15; 34, 35: the last column (32),
    // rows l15 / 16 + l15.  Wave w takes blocks w, w + 8, .. (five for waves 0-3, four for the others).
    int inoff[NBLK], xoff[NBLK];
    unsigned inside = 0, valid = 0;
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const int j = wave + 8 * i;
        const int r1 = j < 34 ? (j >> 1) : (j == 34 ? l15 : 16 + l15);
        const int c1 = j < 34 ? 16 * (j & 1) + l15 : 32;
        const bool ok = r1 < FR_R1H;
        const int r1c = ok ? r1 : FR_R1H - 1;
        const int Y = 2 * oy0 - 1 + r1c, X = 2 * ox0 - 1 + c1;
        inoff[i] = (r1c * FR_R0W + c1) * 8;
        const int xi = r1c * FR_R1W + ((c1 & 1) ? (FR_R1W + 1) / 2 + (c1 >> 1) : (c1 >> 1));   // columns de-interleaved
        xoff[i] = OFF_X + (kq >> 1) * X_KG + xi * 16 + (kq & 1) * 8;
        if (ok) valid |= 1u << i;
        if (ok && Y >= 0 && Y < H && X >= 0 && X < W) inside |= 1u << i;
    }
    // a tile whose halo region lies inside the image needs no zero padding of conv0's output (workgroup-uniform)
    const bool interior = 2 * oy0 - 1 >= 0 && 2 * oy0 - 1 + FR_R1H <= H && 2 * ox0 - 1 >= 0 && 2 * ox0 - 1 + FR_R1W <= W;
    // conv_image: k-step s covers taps 2 s, 2 s + 1 (k-groups 0-1 / 2-3), k-group parity = which 8 of the chunk's 16 channels
    const int yrow = wave;   // this wave's output row of the tile
    int aoff[5], aoff_f;
    {
        const int kg = kq & 1, tsel = kq >> 1;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int tap = min(2 * s + tsel, 8);
            const int ky = tap / 3, kx = tap % 3;
            const int col = kx == 0 ? l15 : (kx == 1 ? (FR_R1W + 1) / 2 + l15 : l15 + 1);
            aoff[s] = OFF_X + (kg * FR_XP + (2 * yrow + ky) * FR_R1W + col) * 16;
        }
        aoff_f = OFF_X + (kg * FR_XP + (2 * yrow + 1) * FR_R1W + (FR_R1W + 1) / 2 + l15) * 16;
    }
    const bool row_live = oy0 + yrow < p.h;   // wave-uniform
    const int nblk = wave + 8 * (NBLK - 1) < FR_NB0 ? NBLK : NBLK - 1;   // wave-uniform

    ff4 mI[NBI], sI[NBI], mF[NBI], sF[NBI];
#pragma unroll
    for (int nb = 0; nb < NBI; ++nb) {
        mI[nb] = (ff4){0.f, 0.f, 0.f, 0.f}; sI[nb] = mI[nb]; mF[nb] = mI[nb]; sF[nb] = mI[nb];
    }
    const float* inv0 = p.tab + 4;
    const float sc0 = un_img * pre0;   // conv0's result leaves the accumulators already in the next window's scale
    __syncthreads();   // IN complete

#pragma unroll 1
    for (int c = 0; c < NC0; ++c) {
        // ---- B: conv0, filters 16 c .. 16 c + 15, over the 561 pixels of the tile's halo region
        {
            fh8 a1[2], a2[2];   // weights straight from L2 / L1: 4 KB per chunk
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                a1[ks] = *reinterpret_cast<const fh8*>(p.w0 + ((c * 2 + ks) * 2 + 0) * 512 + lane * 8);
                a2[ks] = *reinterpret_cast<const fh8*>(p.w0 + ((c * 2 + ks) * 2 + 1) * 512 + lane * 8);
            }
            ff4 sc = *reinterpret_cast<const ff4*>(inv0 + 16 * c + 4 * kq);   // 2^-e of this lane's four filters
            sc *= sc0;
            const f32x2 sc01 = {sc[0], sc[1]}, sc23 = {sc[2], sc[3]};
            auto block = [&](int i, auto border_tag) {
                constexpr bool BORDER = decltype(border_tag)::value;
                const unsigned char* inb = smem + inoff[i];
                ff4 m = (ff4){0.f, 0.f, 0.f, 0.f}, s = m;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const fh4 b1l = *reinterpret_cast<const fh4*>(inb + goff[ks]);
                    const fh4 b1h = *reinterpret_cast<const fh4*>(inb + goff[ks] + 8);
                    const fh8 b1 = __builtin_shufflevector(b1l, b1h, 0, 1, 2, 3, 4, 5, 6, 7);
                    m = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ks], b1, m, 0, 0, 0);
                    if constexpr (!ONE) {
                        const fh4 b2l = *reinterpret_cast<const fh4*>(inb + IN_PART + goff[ks]);
                        const fh4 b2h = *reinterpret_cast<const fh4*>(inb + IN_PART + goff[ks] + 8);
                        const fh8 b2 = __builtin_shufflevector(b2l, b2h, 0, 1, 2, 3, 4, 5, 6, 7);
                        s = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[ks], b1, s, 0, 0, 0);
                        s = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ks], b2, s, 0, 0, 0);
                    }
                }
                // (main + 2^-11 small) 2^-e 2^(k0 - k), LeakyReLU as max(t, slope t) (0 <= slope <= 1), in packed fp32
                f32x2 t01 = (f32x2){s[0], s[1]} * 0.00048828125f + (f32x2){m[0], m[1]};
                f32x2 t23 = (f32x2){s[2], s[3]} * 0.00048828125f + (f32x2){m[2], m[3]};
                t01 *= sc01; t23 *= sc23;
                const f32x2 u01 = t01 * p.slope0, u23 = t23 * p.slope0;
                ff4 v = {fmaxf(t01[0], u01[0]), fmaxf(t01[1], u01[1]), fmaxf(t23[0], u23[0]), fmaxf(t23[1], u23[1])};
                if (BORDER && !((inside >> i) & 1)) v = (ff4){0.f, 0.f, 0.f, 0.f};   // outside the image: the zero padding of conv_image
                fh4 h1, h2;
                fr_split4(v, h1, h2);
                if (!BORDER || ((valid >> i) & 1)) {
                    *reinterpret_cast<fh4*>(smem + xoff[i]) = h1;
                    if constexpr (!ONE) *reinterpret_cast<fh4*>(smem + xoff[i] + X_PART) = h2;
                }
            };
            if (interior) {
#pragma unroll
                for (int i = 0; i < NBLK - 1; ++i) block(i, std::false_type{});     // blocks 0 .. 31: whole rows of 16 pixels
                if (NBLK - 1 < nblk) block(NBLK - 1, std::true_type{});              // 32 .. 35: the last row / the last column
            } else {
#pragma unroll
                for (int i = 0; i < NBLK; ++i)
                    if (i < nblk) block(i, std::true_type{});
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this chunk's conv_image / conv_fused weights (DMA issued a stage ago)
        __syncthreads();
        // ---- C: conv_image (taps in pairs) and conv_fused (centre tap) over these 16 channels
        if (row_live) {
            const unsigned char* wcb = smem + OFF_WC + kq * WC_KQ + l15 * 16;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const fh8 a1 = *reinterpret_cast<const fh8*>(smem + aoff[s]);
                fh8 a2 = a1;
                if constexpr (!ONE) a2 = *reinterpret_cast<const fh8*>(smem + X_PART + aoff[s]);
#pragma unroll
                for (int nb = 0; nb < NBI; ++nb) {
                    const fh8 b1 = *reinterpret_cast<const fh8*>(wcb + s * WC_KS + nb * 256);
                    mI[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, mI[nb], 0, 0, 0);
                    if constexpr (!ONE) {
                        const fh8 b2 = *reinterpret_cast<const fh8*>(wcb + s * WC_KS + WC_PART + nb * 256);
                        sI[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, sI[nb], 0, 0, 0);
                        sI[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, sI[nb], 0, 0, 0);
                    }
                }
            }
            {   // conv_fused: K = this chunk's 16 channels at the centre tap; k-groups 2, 3 of the MFMA are zeroed on the A side
                fh8 a1 = *reinterpret_cast<const fh8*>(smem + aoff_f);
                fh8 a2 = a1;
                if constexpr (!ONE) a2 = *reinterpret_cast<const fh8*>(smem + X_PART + aoff_f);
                if (kq >= 2) { a1 = (fh8)(_Float16)0.f; a2 = a1; }
                const unsigned char* wf = smem + OFF_WC + 5 * WC_KS + (kq & 1) * WC_KQ + l15 * 16;
#pragma unroll
                for (int nb = 0; nb < NBI; ++nb) {
                    const fh8 b1 = *reinterpret_cast<const fh8*>(wf + nb * 256);
                    mF[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, mF[nb], 0, 0, 0);
                    if constexpr (!ONE) {
                        const fh8 b2 = *reinterpret_cast<const fh8*>(wf + 2 * WC_KQ + nb * 256);
                        sF[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, sF[nb], 0, 0, 0);
                        sF[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, sF[nb], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();   // X and the weight buffer are free again
        if (c + 1 < NC0) dma_wc(c + 1);
    }

    // ---- D: a lane holds pixels x = 4 kq .. 4 kq + 3 of row yrow for filter 16 nb + l15
    const int Yo = oy0 + yrow, Xo = ox0 + 4 * kq;
    float amI = 0.f, amF = 0.f;
    f32x2 eI[NEXT ? NBI : 1], eF[NEXT ? NBI : 1];   // NEXT: this lane's two EVEN pixels (x = 4 kq, 4 kq + 2) of both outputs, zero outside the map
    if constexpr (NEXT) {
#pragma unroll
        for (int nb = 0; nb < NBI; ++nb) { eI[nb] = (f32x2){0.f, 0.f}; eF[nb] = eI[nb]; }
    }
    if (row_live && Xo < p.w) {
        const long long oplane = (long long)p.h * p.w;
        const long long pix = (long long)Yo * p.w + Xo;
        ff4 xz[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            xz[j] = (ff4){0.f, 0.f, 0.f, 0.f};
            if (p.xyz) {
                const float* xp = p.xyz + (long long)n * p.xyz_bstride + j * oplane + pix;
                if (p.vec4) {   // w % 4 == 0: the quad is inside the row (the caller checked xyz's alignment with the outputs')
                    xz[j] = *reinterpret_cast<const ff4*>(xp);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (Xo + r < p.w) xz[j][r] = xp[r];
                }
            }
        }
        const float* invI = p.tab + 4 + 64;
        const float* invF = p.tab + 4 + 128;
        const float* wx = p.tab + 4 + 192;
#pragma unroll
        for (int nb = 0; nb < NBI; ++nb) {
            const int f = nb * 16 + l15;
            const float scI = invI[f] * un0, scF = invF[f] * un0;
            const float w0x = wx[f * 3], w1x = wx[f * 3 + 1], w2x = wx[f * 3 + 2];
            ff4 vI, vF;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = __builtin_fmaf(sI[nb][r], 0.00048828125f, mI[nb][r]) * scI;
                vI[r] = a > 0.f ? a : a * p.slope1;
                float b = __builtin_fmaf(sF[nb][r], 0.00048828125f, mF[nb][r]) * scF;
                b = __builtin_fmaf(w2x, xz[2][r], __builtin_fmaf(w1x, xz[1][r], __builtin_fmaf(w0x, xz[0][r], b)));   // fixed order: the same bits in every instantiation
                vF[r] = b > 0.f ? b : b * p.slope1;
            }
            if constexpr (NEXT) {
                const bool two = Xo + 2 < p.w;
                eI[nb] = (f32x2){vI[0], two ? vI[2] : 0.f};
                eF[nb] = (f32x2){vF[0], two ? vF[2] : 0.f};
            }
            if constexpr (NEXT) {
                // the values wait in the accumulator registers: every store of this kernel is issued BEHIND stage E, whose loads would
                // otherwise queue behind these stores in the in-order vmcnt and wait for their write acknowledgements
                mI[nb] = vI; mF[nb] = vF;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (p.vec4 || Xo + r < p.w) { amI = fmaxf(amI, fabsf(vI[r])); amF = fmaxf(amF, fabsf(vF[r])); }
                continue;
            }
            float* oi = p.out_image + (long long)n * p.out_image_bstride + f * oplane + pix;
            float* of = p.out_fused + (long long)n * p.out_fused_bstride + f * oplane + pix;
            if (p.vec4) {
                *reinterpret_cast<ff4*>(oi) = vI;
                *reinterpret_cast<ff4*>(of) = vF;
#pragma unroll
                for (int r = 0; r < 4; ++r) { amI = fmaxf(amI, fabsf(vI[r])); amF = fmaxf(amF, fabsf(vF[r])); }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (Xo + r < p.w) {
                        oi[r] = vI[r]; of[r] = vF[r];
                        amI = fmaxf(amI, fabsf(vI[r])); amF = fmaxf(amF, fabsf(vF[r]));
                    }
            }
        }
    }
    auto store_own = [&]() {   // NEXT: phase D's stores, issued after stage E
        if (row_live && Xo < p.w) {
            const long long oplane = (long long)p.h * p.w;
            const long long pix = (long long)Yo * p.w + Xo;
#pragma unroll
            for (int nb = 0; nb < NBI; ++nb) {
                const int f = nb * 16 + l15;
                float* oi = p.out_image + (long long)n * p.out_image_bstride + f * oplane + pix;
                float* of = p.out_fused + (long long)n * p.out_fused_bstride + f * oplane + pix;
                if (p.vec4) {
                    *reinterpret_cast<ff4*>(oi) = mI[nb];
                    *reinterpret_cast<ff4*>(of) = mF[nb];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (Xo + r < p.w) { oi[r] = mI[nb][r]; of[r] = mF[nb][r]; }
                }
            }
        }
    };
    if (p.amax_out_image) absmax_commit(p.amax_out_image + n, amI);
    if (p.amax_out_fused) absmax_commit(p.amax_out_fused + n, amF);

    // ---- E (NEXT): conv_fused of the next KB level, reference src/net_utils.py:1352-1369 one level down: a 1x1 stride-2 conv over
    // cat[conv_image, xyz, conv_fused] of THIS level samples the pixels (2 y, 2 x) -- for this tile the 4 x 8 even pixels of the two
    // outputs the lanes still hold.  They become split granules G [term][k-group = channel / 8][32 pixels][8] in the (idle) X buffer,
    // window = the tile's own maximum; D[pixel][filter] over K = 2 FI channels, weights straight from L2 / L1 (36 KB, shared by every
    // tile); the three backprojection channels of the next level enter in fp32 like this level's in phase D.
    if constexpr (NEXT) {
        constexpr int FO = 2 * FI, NBK = FO / 16, KG2 = 2 * FI / 8, KS2 = KG2 / 4;
        static_assert(KG2 % 4 == 0, "whole k-steps");
        constexpr int G_PART = KG2 * 32 * 16;
        static_assert(2 * G_PART + 64 <= X_BYTES, "G overlays X");
        unsigned char* const G = smem + OFF_X;
        float* const red2 = reinterpret_cast<float*>(smem + OFF_X + 2 * G_PART);
        const bool even_row = (yrow & 1) == 0;   // wave-uniform
        float tm = 0.f;
        if (even_row) {
#pragma unroll
            for (int nb = 0; nb < NBI; ++nb)
                tm = fmaxf(fmaxf(tm, fmaxf(fabsf(eI[nb][0]), fabsf(eI[nb][1]))), fmaxf(fabsf(eF[nb][0]), fabsf(eF[nb][1])));
        }
        {
            const unsigned tb = wave_max_bits(tm);
            if (lane == 0) red2[wave] = __uint_as_float(tb);
        }
        // wave nbk < NBK owns the filter block 16 nbk .. 16 nbk + 15 for BOTH pixel blocks of the tile, so that a weight fragment is
        // fetched once per tile (the 36 KB of weights against 32 pixels are what this stage costs: every tile pulls them from L2 / L1);
        // the fragments are requested before the barriers
        static_assert(NBK <= 8, "one filter block per wave");
        const bool has_blk = wave < NBK;   // wave-uniform
        fh8 wb1[KS2], wb2[KS2];
        if (has_blk) {
            const int f = 16 * wave + l15;
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                wb1[ks] = *reinterpret_cast<const fh8*>(p.wn + ((long long)((ks * 2 + 0) * 4 + kq) * FO + f) * 8);
                if constexpr (!ONE) wb2[ks] = *reinterpret_cast<const fh8*>(p.wn + ((long long)((ks * 2 + 1) * 4 + kq) * FO + f) * 8);
            }
        }
        __syncthreads();
        tm = fmaxf(fmaxf(fmaxf(red2[0], red2[1]), fmaxf(red2[2], red2[3])), fmaxf(fmaxf(red2[4], red2[5]), fmaxf(red2[6], red2[7])));
        float pre2, un2;
        fr_scales(__builtin_amdgcn_readfirstlane(__float_as_uint(tm)), pre2, un2);
        if (even_row) {
            const int pbase = (yrow >> 1) * 8 + 2 * kq;
#pragma unroll
            for (int nb = 0; nb < NBI; ++nb) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {   // conv_image channels first, conv_fused channels behind them (the reference's cat order)
                    const f32x2 v = (t == 0 ? eI[nb] : eF[nb]) * pre2;
                    const fh2 c1 = __builtin_convertvector(v, fh2);
                    const f32x2 f1 = {(float)c1[0], (float)c1[1]};
                    const f32x2 r = (v - f1) * 2048.f;
                    const fh2 c2 = __builtin_convertvector(r, fh2);
                    const int kg = t * (FI / 8) + 2 * nb + (l15 >> 3);
                    _Float16* g1 = reinterpret_cast<_Float16*>(G + (kg * 32 + pbase) * 16) + (l15 & 7);
                    _Float16* g2 = reinterpret_cast<_Float16*>(G + G_PART + (kg * 32 + pbase) * 16) + (l15 & 7);
                    g1[0] = c1[0]; g1[8] = c1[1];
                    if constexpr (!ONE) { g2[0] = c2[0]; g2[8] = c2[1]; }
                }
            }
        }
        __syncthreads();
        const float* inv2 = p.tab2;
        const float* wx2 = p.tab2 + FO;
        const long long plane2 = (long long)p.h2 * p.w2;
        float am2 = 0.f;
        if (has_blk) {
            const int f = 16 * wave + l15;
            ff4 m[2], sm[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) { m[mb] = (ff4){0.f, 0.f, 0.f, 0.f}; sm[mb] = m[mb]; }
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const fh8 a1 = *reinterpret_cast<const fh8*>(G + ((4 * ks + kq) * 32 + 16 * mb + l15) * 16);
                    m[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, wb1[ks], m[mb], 0, 0, 0);
                    if constexpr (!ONE) {
                        const fh8 a2 = *reinterpret_cast<const fh8*>(G + G_PART + ((4 * ks + kq) * 32 + 16 * mb + l15) * 16);
                        sm[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, wb2[ks], sm[mb], 0, 0, 0);
                        sm[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, wb1[ks], sm[mb], 0, 0, 0);
                    }
                }
            }
            const float sc = inv2[f] * un2;
            const float w0x = wx2[f * 3], w1x = wx2[f * 3 + 1], w2x = wx2[f * 3 + 2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                // rows 4 kq + r of the block = pixels 16 mb + 4 kq + r of the tile's 4 x 8 even pixels: row 2 mb + (kq >> 1), columns 4 (kq & 1) + r
                const int Y2 = (oy0 >> 1) + 2 * mb + (kq >> 1), X2 = (ox0 >> 1) + 4 * (kq & 1);
                if (Y2 < p.h2 && X2 < p.w2) {
                    const long long pix2 = (long long)Y2 * p.w2 + X2;
                    const float* xp = p.xyz2 + (long long)n * p.xyz2_bstride + pix2;
                    float* op = p.out2 + (long long)n * p.out2_bstride + f * plane2 + pix2;
                    ff4 xz[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        if (p.vec4_2) {
                            xz[j] = *reinterpret_cast<const ff4*>(xp + j * plane2);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) xz[j][r] = X2 + r < p.w2 ? xp[j * plane2 + r] : 0.f;
                        }
                    }
                    ff4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = __builtin_fmaf(sm[mb][r], 0.00048828125f, m[mb][r]) * sc;
                        t = __builtin_fmaf(w2x, xz[2][r], __builtin_fmaf(w1x, xz[1][r], __builtin_fmaf(w0x, xz[0][r], t)));
                        v[r] = t > 0.f ? t : t * p.slope2;
                    }
                    if (p.vec4_2) {
                        *reinterpret_cast<ff4*>(op) = v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) am2 = fmaxf(am2, fabsf(v[r]));
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (X2 + r < p.w2) { op[r] = v[r]; am2 = fmaxf(am2, fabsf(v[r])); }
                    }
                }
            }
        }
        store_own();
        if (p.amax_out2) absmax_commit(p.amax_out2 + n, am2);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The depth branch of the same front, the same way:
//     conv0_depth = act(conv3x3(depth))                                    reference src/networks.py:366-367
//     conv_depth  = act(conv3x3 s2 (cat[conv0_depth, coordinates]))        reference src/net_utils.py:1351
//     xyz         = coordinates * act(proj_depth(conv0_depth))             reference src/net_utils.py:1354-1360, at the pixels
//                                                                          (2y, 2x) conv_fused's stride-2 1x1 conv reads
// conv0_depth (16 channels at full resolution) stays on the CU.  `depth` = the S2D output (<= 8 channels: one 16-byte K group
// per pixel and tap, nine groups in three K steps of conv0); conv_depth's 16 tensor channels go through the matrix core
// (taps in pairs), its three coordinate channels K^-1 [x y 1]^T -- affine in the pixel position -- enter in fp32 in the
// epilogue: for a pixel whose nine taps lie inside the image sum_tap w_tap c_j(tap) = c_j(centre) sum w + k_j0 sum w (kx - 1)
// + k_j1 sum w (ky - 1), three pre-summed weights per filter and coordinate; border pixels run the 27 masked terms.
struct DepthFrontParams {
    const float* depth;           // N x Cin x H x W (S2D output)
    long long depth_bstride;
    const float* kinv;            // N x 3 x 3
    const float* tab;             // DF_TAB floats, see depth_front_table_kernel
    const _Float16* w0;           // [3 k-steps][term][4 k-groups = taps][16 filters][8 channels]
    const _Float16* wc;           // [5 k-steps][term][4 k-groups][16 filters][8]: tap 2 s + (kq >> 1), channels 8 (kq & 1) + j
    float* out_depth;
    long long out_depth_bstride;
    float* xyz;                   // N x 3 x h x w
    long long xyz_bstride;
    unsigned* amax_out_depth;
    int N, Cin, H, W, h, w, tilesX, tilesY, ntiles;
    float slope0, slope1, slope_proj;
    int act_proj, vec4;
    S2DStageParams s2d;           // kb1_depth_front_kernel<pool preset>: S2D evaluated on chip from the raw [sparse depth, validity] planes
};
// Stage A of the depth front: NoS2D = `depth` is the S2D tensor in HBM (loaded and split); a pool preset (s2d_pools.h) = the tile
// of that tensor is computed on chip by s2d_stage_run from p.s2d.x (S2D -> conv0_depth -> KB1's depth branch
// as ONE launch; the 8-channel full-resolution S2D tensor never reaches HBM).
struct NoS2D {};
template <typename C>
struct DepthFrontLayout {
    static constexpr bool FUSED = true;
    static constexpr int STAGE = S2DStage<C>::BYTES;
};
template <>
struct DepthFrontLayout<NoS2D> {
    static constexpr bool FUSED = false;
    static constexpr int STAGE = 0;
};
// table (floats): [0] L1max of conv0_depth, [4..19] inv0, [20..35] invC, [36..51] proj, [52..) per filter f (16): S0[3], Sx[3],
// Sy[3] (9), then [196..) the raw coordinate weights wc[f][j][tap] (16 x 27)
constexpr int DF_TAB = 640, DF_SUM = 52, DF_RAW = DF_SUM + 16 * 9;

template <typename S2DCFG>
struct DepthFrontLds {
    using L = DepthFrontLayout<S2DCFG>;
    static constexpr int IN_PART = FR_NIN * 16, IN_BYTES = 2 * IN_PART;            // [term][pixel][8 channels] fp16
    static constexpr int X_KG = FR_XP * 16, X_PART = 2 * X_KG, X_BYTES = 2 * X_PART;
    static constexpr int WC_KQ = 16 * 16, WC_PART = 4 * WC_KQ, WC_KS = 2 * WC_PART, WC_BYTES = 5 * WC_KS;   // 10 KB
    // NoS2D: [IN][X][WC].  On-chip S2D: X and WC overlay the stage's bytes (dead once IN is complete), IN and the reduction scratch behind them
    static constexpr int OFF_X = L::FUSED ? 0 : IN_BYTES, OFF_WC = OFF_X + X_BYTES;
    static constexpr int OFF_IN = L::FUSED ? L::STAGE : 0, OFF_RED = L::FUSED ? OFF_IN + IN_BYTES : OFF_X;
    static constexpr int BYTES = L::FUSED ? OFF_RED + 64 : OFF_WC + WC_BYTES;
    static_assert(!L::FUSED || OFF_WC + WC_BYTES <= OFF_IN, "X and the conv_depth weights fit in front of IN");
    static_assert(BYTES <= 80 * 1024, "two workgroups per CU");
};

template <typename S2DCFG, bool ONE = false>   // ONE: h1 w1 alone (KBN_FP16_ONE_TERM, throughput only; the two-launch form only)
__global__ __launch_bounds__(FR_THREADS, 2) void kb1_depth_front_kernel(const DepthFrontParams p) {
    using LD = DepthFrontLds<S2DCFG>;
    constexpr bool FUSED = LD::L::FUSED;
    constexpr int IN_PART = LD::IN_PART, X_KG = LD::X_KG, X_PART = LD::X_PART;
    constexpr int WC_KQ = LD::WC_KQ, WC_PART = LD::WC_PART, WC_KS = LD::WC_KS, WC_BYTES = LD::WC_BYTES;
    constexpr int OFF_X = LD::OFF_X, OFF_WC = LD::OFF_WC, OFF_IN = LD::OFF_IN;
    constexpr int NBLK = (FR_NB0 + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0");   // fp16 results flush subnormals
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    int bid = xcd_remap(blockIdx.x, p.ntiles);
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int oy0 = ty * FR_TH, ox0 = tx * FR_TW;
    const int H = p.H, W = p.W;
    const long long plane = (long long)H * W;

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(reinterpret_cast<const float*>(smem)));
    auto stage_wc = [&]() {   // conv_depth's weights: 10 KB by LDS-DMA, awaited in front of the barrier that ends phase B
        constexpr int n4 = WC_BYTES / 16;
#pragma unroll
        for (int e0 = 0; e0 < n4; e0 += FR_THREADS) {
            const int eb = e0 + wave * 64;
            if (eb + lane < n4) lds_dma16_s(reinterpret_cast<const float*>(p.wc) + eb * 4, (unsigned)(lane * 16), lds0 + (unsigned)(OFF_WC + eb * 16));
        }
    };
    if constexpr (!FUSED) stage_wc();

    // ---- A: depth-feature tile -> split granules [8 channels] per pixel; windows from the tile's own maximum (see kb1_front_kernel)
    float pre_in, un_in, pre0, un0;
    if constexpr (FUSED) {
        // the S2D layer for this tile, on chip (s2d_stage.h): IN is written from the raw sparse depth / validity planes
        float bound_in;
        s2d_stage_run<S2DCFG>(p.s2d, smem, OFF_IN, reinterpret_cast<float*>(smem + LD::OFF_RED), n, 2 * oy0 - 2, 2 * ox0 - 2, H, W, pre_in, un_in, bound_in);
        fr_scales(__float_as_uint(p.tab[0] * bound_in), pre0, un0);
    } else {
        const float* src0 = p.depth + (long long)n * p.depth_bstride;
        float raw[2][8];
        float tm = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pix = u * FR_THREADS + tid;
            const int r = pix / FR_R0W, c = pix - r * FR_R0W;
            const int Y = 2 * oy0 - 2 + r, X = 2 * ox0 - 2 + c;
            const bool ok = pix < FR_NP0 && Y >= 0 && Y < H && X >= 0 && X < W;
            const float* src = src0 + (long long)(ok ? Y : 0) * W + (ok ? X : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                raw[u][j] = (ok && j < p.Cin) ? src[(long long)(j < p.Cin ? j : 0) * plane] : 0.f;
                tm = fmaxf(tm, fabsf(raw[u][j]));
            }
        }
        tm = __uint_as_float(wave_max_bits(tm));
        float* red = reinterpret_cast<float*>(smem + LD::OFF_RED);
        if (lane == 0) red[wave] = tm;
        __syncthreads();
        tm = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
        const unsigned abits = __builtin_amdgcn_readfirstlane(__float_as_uint(tm));
        fr_scales(abits, pre_in, un_in);
        fr_scales(__float_as_uint(p.tab[0] * __uint_as_float(abits)), pre0, un0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pix = u * FR_THREADS + tid;
            if (pix < FR_NIN) {
                fh4 a1, a2, b1, b2;
                fr_split4((ff4){raw[u][0], raw[u][1], raw[u][2], raw[u][3]} * pre_in, a1, a2);
                fr_split4((ff4){raw[u][4], raw[u][5], raw[u][6], raw[u][7]} * pre_in, b1, b2);
                *reinterpret_cast<fh8*>(smem + OFF_IN + pix * 16) = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                *reinterpret_cast<fh8*>(smem + OFF_IN + IN_PART + pix * 16) = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    }

    // ---- per-lane offsets
    int tapoff[3];   // conv0: k-group kq of k-step ks is tap 4 ks + kq (taps past the ninth carry zero weights: any valid address)
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        const int tap = min(4 * ks + kq, 8);
        tapoff[ks] = ((tap / 3) * FR_R0W + tap % 3) * 16;
    }
    int inoff[NBLK], xoff[NBLK];
    unsigned inside = 0, valid = 0;
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const int j = wave + 8 * i;
        const int r1 = j < 34 ? (j >> 1) : (j == 34 ? l15 : 16 + l15);
        const int c1 = j < 34 ? 16 * (j & 1) + l15 : 32;
        const bool ok = r1 < FR_R1H;
        const int r1c = ok ? r1 : FR_R1H - 1;
        const int Y = 2 * oy0 - 1 + r1c, X = 2 * ox0 - 1 + c1;
        inoff[i] = OFF_IN + (r1c * FR_R0W + c1) * 16;
        const int xi = r1c * FR_R1W + ((c1 & 1) ? (FR_R1W + 1) / 2 + (c1 >> 1) : (c1 >> 1));
        xoff[i] = OFF_X + (kq >> 1) * X_KG + xi * 16 + (kq & 1) * 8;
        if (ok) valid |= 1u << i;
        if (ok && Y >= 0 && Y < H && X >= 0 && X < W) inside |= 1u << i;
    }
    const bool interior = 2 * oy0 - 1 >= 0 && 2 * oy0 - 1 + FR_R1H <= H && 2 * ox0 - 1 >= 0 && 2 * ox0 - 1 + FR_R1W <= W;
    const int yrow = wave;
    int aoff[5];
    {
        const int kg = kq & 1, tsel = kq >> 1;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int tap = min(2 * s + tsel, 8);
            const int ky = tap / 3, kx = tap % 3;
            const int col = kx == 0 ? l15 : (kx == 1 ? (FR_R1W + 1) / 2 + l15 : l15 + 1);
            aoff[s] = OFF_X + (kg * FR_XP + (2 * yrow + ky) * FR_R1W + col) * 16;
        }
    }
    const int zoff = OFF_X + ((kq >> 1) * FR_XP + (2 * yrow + 1) * FR_R1W + (FR_R1W + 1) / 2 + l15) * 16 + (kq & 1) * 8;   // centre tap, this lane's 4 channels
    const bool row_live = oy0 + yrow < p.h;
    const int nblk = wave + 8 * (NBLK - 1) < FR_NB0 ? NBLK : NBLK - 1;
    __syncthreads();   // IN complete
    if constexpr (FUSED) stage_wc();   // its bytes (and X's) were the S2D stage's until this barrier

    // ---- B: conv0_depth over the 561 pixels of the halo region
    {
        fh8 a1[3], a2[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            a1[ks] = *reinterpret_cast<const fh8*>(p.w0 + (ks * 2 + 0) * 512 + lane * 8);
            a2[ks] = *reinterpret_cast<const fh8*>(p.w0 + (ks * 2 + 1) * 512 + lane * 8);
        }
        ff4 sc = *reinterpret_cast<const ff4*>(p.tab + 4 + 4 * kq);
        sc *= un_in * pre0;
        const f32x2 sc01 = {sc[0], sc[1]}, sc23 = {sc[2], sc[3]};
        auto block = [&](int i, auto border_tag) {
            constexpr bool BORDER = decltype(border_tag)::value;
            const unsigned char* inb = smem + inoff[i];
            ff4 m = (ff4){0.f, 0.f, 0.f, 0.f}, s = m;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const fh8 b1 = *reinterpret_cast<const fh8*>(inb + tapoff[ks]);
                m = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ks], b1, m, 0, 0, 0);
                if constexpr (!ONE) {
                    const fh8 b2 = *reinterpret_cast<const fh8*>(inb + IN_PART + tapoff[ks]);
                    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[ks], b1, s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ks], b2, s, 0, 0, 0);
                }
            }
            f32x2 t01 = (f32x2){s[0], s[1]} * 0.00048828125f + (f32x2){m[0], m[1]};
            f32x2 t23 = (f32x2){s[2], s[3]} * 0.00048828125f + (f32x2){m[2], m[3]};
            t01 *= sc01; t23 *= sc23;
            const f32x2 u01 = t01 * p.slope0, u23 = t23 * p.slope0;
            ff4 v = {fmaxf(t01[0], u01[0]), fmaxf(t01[1], u01[1]), fmaxf(t23[0], u23[0]), fmaxf(t23[1], u23[1])};
            if (BORDER && !((inside >> i) & 1)) v = (ff4){0.f, 0.f, 0.f, 0.f};
            fh4 h1, h2;
            fr_split4(v, h1, h2);
            if (!BORDER || ((valid >> i) & 1)) {
                *reinterpret_cast<fh4*>(smem + xoff[i]) = h1;
                if constexpr (!ONE) *reinterpret_cast<fh4*>(smem + xoff[i] + X_PART) = h2;
            }
        };
        if (FUSED && (p.s2d.dbg & 32)) {
        } else if (interior) {
#pragma unroll
            for (int i = 0; i < NBLK - 1; ++i) block(i, std::false_type{});
            if (NBLK - 1 < nblk) block(NBLK - 1, std::true_type{});
        } else {
#pragma unroll
            for (int i = 0; i < NBLK; ++i)
                if (i < nblk) block(i, std::true_type{});
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // conv_depth's weights
    __syncthreads();

    if (!row_live || (FUSED && (p.s2d.dbg & 64))) return;   // wave-uniform; no barrier follows
    // ---- C: conv_depth's tensor channels (taps in pairs)
    ff4 mD = (ff4){0.f, 0.f, 0.f, 0.f}, sD = mD;
    {
        const unsigned char* wcb = smem + OFF_WC + kq * WC_KQ + l15 * 16;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const fh8 a1 = *reinterpret_cast<const fh8*>(smem + aoff[s]);
            const fh8 b1 = *reinterpret_cast<const fh8*>(wcb + s * WC_KS);
            mD = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, mD, 0, 0, 0);
            if constexpr (!ONE) {
                const fh8 a2 = *reinterpret_cast<const fh8*>(smem + X_PART + aoff[s]);
                const fh8 b2 = *reinterpret_cast<const fh8*>(wcb + s * WC_KS + WC_PART);
                sD = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, sD, 0, 0, 0);
                sD = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, sD, 0, 0, 0);
            }
        }
    }
    const float* ki = p.kinv + (long long)n * 9;
    const float k00 = ki[0], k01 = ki[1], k02 = ki[2], k10 = ki[3], k11 = ki[4], k12 = ki[5], k20 = ki[6], k21 = ki[7], k22 = ki[8];
    const int Yo = oy0 + yrow;
    const long long oplane = (long long)p.h * p.w;
    // ---- z = act(proj . conv0_depth) at (2 Yo, 2 x): lane (x = l15, kq) sums its 4 channels, the 4 lanes of a pixel share the sum;
    // lane kq = j < 3 then stores xyz channel j of pixel x
    {
        const fh4 h1 = *reinterpret_cast<const fh4*>(smem + zoff);
        fh4 h2 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};   // ONE: no h2 granules exist
        if constexpr (!ONE) h2 = *reinterpret_cast<const fh4*>(smem + zoff + X_PART);
        const ff4 pw = *reinterpret_cast<const ff4*>(p.tab + 36 + 4 * kq);
        float z = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) z = __builtin_fmaf(__builtin_fmaf((float)h2[r], 0.00048828125f, (float)h1[r]), pw[r], z);
        z += __shfl_xor(z, 16);
        z += __shfl_xor(z, 32);
        z *= un0;
        if (p.act_proj) z = z > 0.f ? z : z * p.slope_proj;
        const int Xz = ox0 + l15;
        if (kq < 3 && Xz < p.w) {
            const float X = (float)(2 * Xz), Y = (float)(2 * Yo);
            const float c = kq == 0 ? (__builtin_fmaf(k01, Y, k00 * X) + k02) : (kq == 1 ? (__builtin_fmaf(k11, Y, k10 * X) + k12) : (__builtin_fmaf(k21, Y, k20 * X) + k22));
            p.xyz[(long long)n * p.xyz_bstride + kq * oplane + (long long)Yo * p.w + Xz] = c * z;
        }
    }
    // ---- D: a lane holds pixels x = 4 kq .. 4 kq + 3 of row Yo for filter l15; coordinate channels in fp32
    const int Xo = ox0 + 4 * kq;
    float am = 0.f;
    if (Xo < p.w) {
        const int f = l15;
        const float scD = p.tab[20 + f] * un0;
        const float* sm = p.tab + DF_SUM + f * 9;   // S0[3], Sx[3], Sy[3]
        const float kj0[3] = {k00, k10, k20}, kj1[3] = {k01, k11, k21}, kj2[3] = {k02, k12, k22};
        float cst = 0.f;                             // sum_j k_j0 Sx_j + k_j1 Sy_j
#pragma unroll
        for (int j = 0; j < 3; ++j) cst += kj0[j] * sm[3 + j] + kj1[j] * sm[6 + j];
        ff4 v;
        const bool rows_in = 2 * Yo - 1 >= 0 && 2 * Yo + 1 < H;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int Xc = 2 * (Xo + r), Yc = 2 * Yo;
            float cterm;
            if (rows_in && Xc - 1 >= 0 && Xc + 1 < W) {
                cterm = cst;
#pragma unroll
                for (int j = 0; j < 3; ++j) cterm += (__builtin_fmaf(kj1[j], (float)Yc, kj0[j] * (float)Xc) + kj2[j]) * sm[j];
            } else {   // a tap outside the image reads the zero padding: the 27 terms one by one
                const float* wr = p.tab + DF_RAW + f * 27;
                cterm = 0.f;
                for (int j = 0; j < 3; ++j)
                    for (int t = 0; t < 9; ++t) {
                        const int Yt = Yc + t / 3 - 1, Xt = Xc + t % 3 - 1;
                        if (Yt >= 0 && Yt < H && Xt >= 0 && Xt < W) cterm += (__builtin_fmaf(kj1[j], (float)Yt, kj0[j] * (float)Xt) + kj2[j]) * wr[j * 9 + t];
                    }
            }
            const float a = __builtin_fmaf(sD[r], 0.00048828125f, mD[r]) * scD + cterm;
            v[r] = a > 0.f ? a : a * p.slope1;
        }
        float* o = p.out_depth + (long long)n * p.out_depth_bstride + f * oplane + (long long)Yo * p.w + Xo;
        if (p.vec4) {
            *reinterpret_cast<ff4*>(o) = v;
            am = sp_amax4f(am, v);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (Xo + r < p.w) { o[r] = v[r]; am = fmaxf(am, fabsf(v[r])); }
        }
    }
    if (p.amax_out_depth) absmax_commit(p.amax_out_depth + n, am);
}

// table of the depth front: scales, the bound factor, proj, pre-summed and raw coordinate weights
__global__ void depth_front_table_kernel(const float* __restrict__ w0, const float* __restrict__ wc, const float* __restrict__ proj,
                                         float* __restrict__ tab, int Cin) {
    __shared__ float red[256], red2[256];
    const int which = blockIdx.y, f = blockIdx.x;   // which 0: conv0_depth (Cin x 9 per filter), 1: conv_depth (19 x 9)
    const int per = which == 0 ? Cin * 9 : 19 * 9;
    const float* w = which == 0 ? w0 : wc;
    float m = 0.f, l1 = 0.f;
    for (int i = threadIdx.x; i < per; i += 256) {
        const float a = fabsf(w[(long long)f * per + i]);
        m = fmaxf(m, a);
        l1 += a;
    }
    red[threadIdx.x] = m;
    red2[threadIdx.x] = l1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
            red2[threadIdx.x] += red2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int ex = FR_WEXP;
        if (red[0] > 0.f && red[0] < 3.0e38f) (void)frexpf(red[0], &ex);
        int e = FR_WEXP - ex;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        tab[4 + 16 * which + f] = ldexpf(1.f, -e);
        if (which == 0) {
            atomicMax(reinterpret_cast<unsigned*>(tab), __float_as_uint(red2[0]));
            tab[36 + f] = proj[f];
        } else {
            for (int j = 0; j < 3; ++j) {
                float s0 = 0.f, sx = 0.f, sy = 0.f;
                for (int t = 0; t < 9; ++t) {
                    const float wv = wc[((long long)f * 19 + 16 + j) * 9 + t];
                    tab[DF_RAW + f * 27 + j * 9 + t] = wv;
                    s0 += wv; sx += wv * (float)(t % 3 - 1); sy += wv * (float)(t / 3 - 1);
                }
                tab[DF_SUM + f * 9 + j] = s0; tab[DF_SUM + f * 9 + 3 + j] = sx; tab[DF_SUM + f * 9 + 6 + j] = sy;
            }
        }
    }
}

__global__ void depth_front_pack_kernel(const float* __restrict__ w0, const float* __restrict__ wc, const float* __restrict__ tab,
                                        _Float16* __restrict__ out0, _Float16* __restrict__ outc, int Cin) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 3 * 2 * 64 * 8) {   // conv0_depth: [ks 3][term][kq 4][m 16][j 8], tap = 4 ks + kq
        int r = e;
        const int j = r & 7; r >>= 3;
        const int m = r & 15; r >>= 4;
        const int kq = r & 3; r >>= 2;
        const int term = r & 1; r >>= 1;
        const int tap = 4 * r + kq;
        out0[e] = (tap < 9 && j < Cin) ? fr_term(w0[((long long)m * Cin + j) * 9 + tap] / tab[4 + m], term) : (_Float16)0.f;
    }
    if (e < 5 * 2 * 64 * 8) {   // conv_depth: [s 5][term][kq 4][n 16][j 8], tap = 2 s + (kq >> 1), channel 8 (kq & 1) + j
        int r = e;
        const int j = r & 7; r >>= 3;
        const int nf = r & 15; r >>= 4;
        const int kq = r & 3; r >>= 2;
        const int term = r & 1; r >>= 1;
        const int tap = 2 * r + (kq >> 1), ch = 8 * (kq & 1) + j;
        outc[e] = tap < 9 ? fr_term(wc[((long long)nf * 19 + ch) * 9 + tap] / tab[20 + nf], term) : (_Float16)0.f;
    }
}

// blob of the on-chip S2D stage (s2d_stage.h): tables, then the A operands of its MFMAs lane by lane.  ONE workgroup.
//   chain layer i (cin = n_pools for i = 0, else 8 input channels), two operands: lane (m = lane & 15, kq = lane >> 4), rows 0-7 = filter m at the
//       block's first pixel (k-groups 0, 1 = its h1, h2), rows 8-15 = filter m - 8 at the second pixel (k-groups 2, 3):
//       operand 0: [w1 = fp16(w 2^e) | fp16(w1 2^-11)] (the second pairs with the scaled residual h2), operand 1: [w2 = fp16(w 2^e - w1) | 0]
//   3x3 conv, K-steps s = 0..2 (window row): rows 0-7 = filter m at the pair's first pixel (window column kq = kx, kq 3: zero), rows 8-15 =
//       filter m - 8 at its second pixel (kx = kq - 1, kq 0: zero), k-group entry j = feature channel j; terms (w1, (w 2^e - w1) 2^11)
//       K-step 3 (raw channels): k-group kq = window row (kq 3: zero), entry j = (window column j >> 1, channel 8 + (j & 1))
__global__ __launch_bounds__(256) void s2d_stage_pack_kernel(const float* __restrict__ wp0, const float* __restrict__ wp1,
                                                             const float* __restrict__ wp2, const float* __restrict__ wc, float* __restrict__ tab,
                                                             _Float16* __restrict__ wchain, _Float16* __restrict__ wconv, int n_pools) {
    __shared__ float inv[4][8], l1[4][8];
    const int t = threadIdx.x;
    for (int e = t; e < SF_TAB; e += 256) tab[e] = 0.f;
    if (t < 32) {
        const int layer = t >> 3, f = t & 7;
        const int per = layer == 0 ? n_pools : (layer == 3 ? 90 : 8);
        const float* w = (layer == 0 ? wp0 : layer == 1 ? wp1 : layer == 2 ? wp2 : wc) + (long long)f * per;
        float m = 0.f, s1 = 0.f;
        for (int i = 0; i < per; ++i) { m = fmaxf(m, fabsf(w[i])); s1 += fabsf(w[i]); }
        int ex = FR_WEXP;
        if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &ex);
        int e = FR_WEXP - ex;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        inv[layer][f] = ldexpf(1.f, -e);
        l1[layer][f] = s1;
    }
    __syncthreads();
    if (t < 32) tab[8 + t] = inv[t >> 3][t & 7];
    if (t < 4) {
        float m = 0.f;
        for (int f = 0; f < 8; ++f) m = fmaxf(m, l1[t][f]);
        tab[t] = m;
    }
    for (int e = t; e < SF_CHAIN_HALVES; e += 256) {
        const int j = e & 7, lane = (e >> 3) & 63, op = (e >> 9) & 1, layer = e >> 10;
        const int m = lane & 15, kq = lane >> 4;
        const int f = m & 7, second = m >> 3;       // rows 8-15: the same filters at the block's second pixel (k-groups 2, 3)
        const int cin = layer == 0 ? n_pools : 8;
        float v = 0.f;
        if (j < cin && (kq >> 1) == second) {
            const float* w = layer == 0 ? wp0 : (layer == 1 ? wp1 : wp2);
            const float ws = w[f * cin + j] / inv[layer][f];
            const _Float16 w1 = (_Float16)ws;
            if (op == 0) v = (kq & 1) ? (float)w1 * 0.00048828125f : (float)w1;   // [w1 | w1 2^-11] . [h1 ; h2]
            else v = (kq & 1) ? 0.f : ws - (float)w1;                                // [w2 | 0] . [h1 ; h2]
        }
        wchain[e] = (_Float16)v;
    }
    for (int e = t; e < SF_CONV_HALVES; e += 256) {
        const int j = e & 7, lane = (e >> 3) & 63, term = (e >> 9) & 1, s = e >> 10;
        const int m = lane & 15, kq = lane >> 4;
        const int f = m & 7, second = m >> 3;
        float ws = 0.f;
        bool live = false;
        if (s < 3) {              // features: window row s, window column kq
            const int kx = kq - second;
            if (kx >= 0 && kx < 3) { live = true; ws = wc[((f * 10 + j) * 3 + s) * 3 + kx]; }
        } else if (kq < 3) {      // raw channels: window row kq, window column j >> 1
            const int kx = (j >> 1) - second;
            if (kx >= 0 && kx < 3) { live = true; ws = wc[((f * 10 + 8 + (j & 1)) * 3 + kq) * 3 + kx]; }
        }
        wconv[e] = live ? fr_term(ws / inv[3][f], term) : (_Float16)0.f;
    }
}

// ---- weight packing ---------------------------------------------------------------------------------------------
// table: per-filter 2^-e (largest |w 2^e| in [2^12, 2^13)), the bound factor L1max0 = max_f sum |w0_f|, the fp32 xyz weights
__global__ void front_table_kernel(const float* __restrict__ w0, const float* __restrict__ wi, const float* __restrict__ wf,
                                   float* __restrict__ tab, int Cin, int F0, int FI) {
    __shared__ float red[256], red2[256];
    const int which = blockIdx.y, f = blockIdx.x;   // which 0: conv0, 1: conv_image, 2: conv_fused
    const int nf = which == 0 ? F0 : FI;
    const int per = which == 0 ? Cin * 9 : (which == 1 ? F0 * 9 : F0 + 3);
    const float* w = which == 0 ? w0 : (which == 1 ? wi : wf);
    float m = 0.f, l1 = 0.f;
    if (f < nf)
        for (int i = threadIdx.x; i < per; i += 256) {
            const float a = fabsf(w[(long long)f * per + i]);
            m = fmaxf(m, a);
            l1 += a;
        }
    red[threadIdx.x] = m;
    red2[threadIdx.x] = l1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
            red2[threadIdx.x] += red2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int ex = FR_WEXP;
        if (red[0] > 0.f && red[0] < 3.0e38f) (void)frexpf(red[0], &ex);
        int e = FR_WEXP - ex;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        tab[4 + 64 * which + f] = ldexpf(1.f, -e);
        if (which == 0 && f < F0) atomicMax(reinterpret_cast<unsigned*>(tab), __float_as_uint(red2[0]));   // L1max0 (non-negative floats order like their bits)
        if (which == 2 && f < FI)
            for (int j = 0; j < 3; ++j) tab[4 + 192 + f * 3 + j] = wf[(long long)f * (F0 + 3) + F0 + j];
    }
}


// conv0 panel: [chunk][k-step 2][term][k-group 4][filter 16][8]; k-group g = 4 ks + kq = (tap row g >> 1, column pair g & 1),
// slot j = (column 2 (g & 1) + (j >> 2), channel j & 3); groups >= 6, column 3, channels >= Cin: zero
__global__ void front_pack0_kernel(const float* __restrict__ w0, const float* __restrict__ tab, _Float16* __restrict__ out, int Cin, int F0,
                                   int total) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    int r = e;
    const int j = r & 7; r >>= 3;
    const int m = r & 15; r >>= 4;
    const int kq = r & 3; r >>= 2;
    const int term = r & 1; r >>= 1;
    const int ks = r & 1, c = r >> 1;
    const int g = 4 * ks + kq, f = 16 * c + m;
    const int ky = g >> 1, kx = 2 * (g & 1) + (j >> 2), ch = j & 3;
    _Float16 h = (_Float16)0.f;
    if (g < 6 && kx < 3 && ch < Cin && f < F0) h = fr_term(w0[((long long)f * Cin + ch) * 9 + ky * 3 + kx] / tab[4 + f], term);
    out[e] = h;
}

// conv_image + conv_fused panel, per chunk: [k-step 5][term][k-group 4][filter FI][8]: tap 2 s + (kq >> 1) (< 9, else zero),
// channel 16 chunk + 8 (kq & 1) + j of conv_image; then [term][k-group 2][filter FI][8]: conv_fused's weights of channels
// 16 chunk + 8 kq + j
__global__ void front_packc_kernel(const float* __restrict__ wi, const float* __restrict__ wf, const float* __restrict__ tab,
                                   _Float16* __restrict__ out, int F0, int FI, int total) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int per_chunk = (5 * 2 * 4 + 2 * 2) * FI * 8;
    const int c = e / per_chunk;
    int r = e - c * per_chunk;
    _Float16 h = (_Float16)0.f;
    if (r < 5 * 2 * 4 * FI * 8) {
        const int j = r & 7; r >>= 3;
        const int f = r % FI; r /= FI;
        const int kq = r & 3; r >>= 2;
        const int term = r & 1; r >>= 1;
        const int s = r;
        const int ch = 16 * c + 8 * (kq & 1) + j, tap = 2 * s + (kq >> 1);
        if (ch < F0 && tap < 9) h = fr_term(wi[((long long)f * F0 + ch) * 9 + tap] / tab[4 + 64 + f], term);
    } else {
        r -= 5 * 2 * 4 * FI * 8;
        const int j = r & 7; r >>= 3;
        const int f = r % FI; r /= FI;
        const int kg = r & 1; r >>= 1;
        const int term = r;
        const int ch = 16 * c + 8 * kg + j;
        if (ch < F0) h = fr_term(wf[(long long)f * (F0 + 3) + ch] / tab[4 + 128 + f], term);
    }
    out[e] = h;
}

// next-level conv_fused panel (kb1_front_kernel<.., NEXT>): tab2 = [2^-e per filter FO][xyz weights FO x 3], then
// [k-step][term][k-group 4][filter FO][8]: K index 32 ks + 8 kq + j = channel c of cat[conv_image (CI), conv_fused (CF)];
// the reference's weight columns are [image CI | xyz 3 | fused CF] (src/net_utils.py:1362-1368)
__global__ void front_next_pack_kernel(const float* __restrict__ wf, float* __restrict__ tab2, _Float16* __restrict__ out, int CI, int CF, int FO,
                                       int total) {
    __shared__ float inv[128];
    const int per = CI + 3 + CF;
    for (int f = threadIdx.x; f < FO; f += blockDim.x) {
        float m = 0.f;
        for (int c = 0; c < per; ++c)
            if (c < CI || c >= CI + 3) m = fmaxf(m, fabsf(wf[(long long)f * per + c]));
        int ex = FR_WEXP;
        if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &ex);
        int e = FR_WEXP - ex;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        inv[f] = ldexpf(1.f, -e);
        tab2[f] = inv[f];
        for (int j = 0; j < 3; ++j) tab2[FO + f * 3 + j] = wf[(long long)f * per + CI + j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        int r = e;
        const int j = r & 7; r >>= 3;
        const int f = r % FO; r /= FO;
        const int kq = r & 3; r >>= 2;
        const int term = r & 1; r >>= 1;
        const int c = 32 * r + 8 * kq + j;
        _Float16 h = (_Float16)0.f;
        if (c < CI + CF) h = fr_term(wf[(long long)f * per + (c < CI ? c : c + 3)] / inv[f], term);
        out[e] = h;
    }
}

}  // namespace kbn

extern "C" {

static bool front_shape_ok(int c_in, int f0, int fi) { return c_in >= 1 && c_in <= 4 && f0 == 48 && fi == 48; }
static size_t front_w0_halves(int f0) { return (size_t)(f0 / 16) * 2 * 2 * 64 * 8; }
static size_t front_wc_halves(int f0, int fi) { return (size_t)(f0 / 16) * (5 * 2 * 4 + 2 * 2) * fi * 8; }

size_t kbn_kb1_front_packed_weight_bytes(int image_channels, int conv0_filters, int kb_filters) {
    if (!front_shape_ok(image_channels, conv0_filters, kb_filters)) return 0;
    return (size_t)kbn::FR_TAB * 4 + 2 * (front_w0_halves(conv0_filters) + front_wc_halves(conv0_filters, kb_filters));
}

int kbn_kb1_front_pack_weight(const float* w_conv0, const float* w_conv_image, const float* w_conv_fused, void* packed,
                              int image_channels, int conv0_filters, int kb_filters, kbn_stream_t stream) {
    using namespace kbn;
    if (!w_conv0 || !w_conv_image || !w_conv_fused || !packed) return KBN_ERR_INVALID_ARGUMENT;
    if (!front_shape_ok(image_channels, conv0_filters, kb_filters)) return KBN_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    float* tab = static_cast<float*>(packed);
    if (hipMemsetAsync(tab, 0, FR_TAB * 4, st) != hipSuccess) return KBN_ERR_LAUNCH;
    hipLaunchKernelGGL(front_table_kernel, dim3(64, 3), dim3(256), 0, st, w_conv0, w_conv_image, w_conv_fused, tab, image_channels,
                       conv0_filters, kb_filters);
    _Float16* p0 = reinterpret_cast<_Float16*>(tab + FR_TAB);
    const int t0 = (int)front_w0_halves(conv0_filters);
    hipLaunchKernelGGL(front_pack0_kernel, dim3((t0 + 255) / 256), dim3(256), 0, st, w_conv0, tab, p0, image_channels, conv0_filters, t0);
    const int tc = (int)front_wc_halves(conv0_filters, kb_filters);
    hipLaunchKernelGGL(front_packc_kernel, dim3((tc + 255) / 256), dim3(256), 0, st, w_conv_image, w_conv_fused, tab, p0 + t0, conv0_filters,
                       kb_filters, tc);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_kb1_front_query(int image_channels, int conv0_filters, int kb_filters, int height, int width, float conv0_negative_slope) {
    using namespace kbn;
    if (height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (!front_shape_ok(image_channels, conv0_filters, kb_filters) || knob(KNOB_NO_SPLIT)) return KBN_ERR_UNSUPPORTED;
    if ((long long)height * width > 0x1fffffffLL) return KBN_ERR_UNSUPPORTED;
    if (!(conv0_negative_slope >= 0.f && conv0_negative_slope <= 1.f)) return KBN_ERR_UNSUPPORTED;   // LeakyReLU as max(t, slope t)
    return KBN_OK;
}

// ---- the next level's conv_fused inside the same launch (kb1_front_kernel<3, 3, true>) ----
static bool front_next_shape_ok(int ci, int cf, int fo) { return ci == 48 && cf == 48 && fo == 96; }
static size_t front_next_tab_floats(int fo) { return (size_t)fo * 4; }
static size_t front_next_halves(int ci, int cf, int fo) { return (size_t)((ci + cf) / 32) * 2 * 4 * fo * 8; }

size_t kbn_kb1_front_next_packed_weight_bytes(int image_channels, int fused_channels, int filters) {
    if (!front_next_shape_ok(image_channels, fused_channels, filters)) return 0;
    return front_next_tab_floats(filters) * 4 + 2 * front_next_halves(image_channels, fused_channels, filters);
}

int kbn_kb1_front_next_pack_weight(const float* w_conv_fused, void* packed, int image_channels, int fused_channels, int filters,
                                   kbn_stream_t stream) {
    using namespace kbn;
    if (!w_conv_fused || !packed) return KBN_ERR_INVALID_ARGUMENT;
    if (!front_next_shape_ok(image_channels, fused_channels, filters)) return KBN_ERR_UNSUPPORTED;
    float* tab2 = static_cast<float*>(packed);
    hipLaunchKernelGGL(front_next_pack_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w_conv_fused, tab2,
                       reinterpret_cast<_Float16*>(tab2 + front_next_tab_floats(filters)), image_channels, fused_channels, filters,
                       (int)front_next_halves(image_channels, fused_channels, filters));
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_kb1_front_next_query(int image_channels, int conv0_filters, int kb_filters, int next_filters, int height, int width,
                             float conv0_negative_slope) {
    using namespace kbn;
    if (int rc = kbn_kb1_front_query(image_channels, conv0_filters, kb_filters, height, width, conv0_negative_slope)) return rc;
    if (!front_next_shape_ok(kb_filters, kb_filters, next_filters) || knob(KNOB_NO_FRONT_NEXT)) return KBN_ERR_UNSUPPORTED;
    return KBN_OK;
}

static int kb1_front_launch(const float* image, long long image_batch_stride, const void* packed_weight,
                            const float* xyz, long long xyz_batch_stride, float* out_image, long long out_image_batch_stride,
                            float* out_fused, long long out_fused_batch_stride, int n, int image_channels, int conv0_filters,
                            int kb_filters, int height, int width, float conv0_negative_slope, float kb_negative_slope,
                            unsigned* out_image_absmax, unsigned* out_fused_absmax, const void* packed_next, const float* xyz_next,
                            long long xyz_next_batch_stride, float* out_next, long long out_next_batch_stride, int next_filters,
                            float next_negative_slope, unsigned* out_next_absmax, kbn_stream_t stream) {
    using namespace kbn;
    if (!image || !packed_weight || !out_image || !out_fused || n < 1 || height < 1 || width < 1)
        return KBN_ERR_INVALID_ARGUMENT;
    if (int rc = kbn_kb1_front_query(image_channels, conv0_filters, kb_filters, height, width, conv0_negative_slope)) return rc;
    if (packed_next) {
        if (!xyz_next || !out_next) return KBN_ERR_INVALID_ARGUMENT;
        if (int rc = kbn_kb1_front_next_query(image_channels, conv0_filters, kb_filters, next_filters, height, width, conv0_negative_slope)) return rc;
    }
    FrontParams p{};
    p.image = image; p.image_bstride = image_batch_stride;
    p.tab = static_cast<const float*>(packed_weight);
    p.w0 = reinterpret_cast<const _Float16*>(p.tab + FR_TAB);
    p.wc = p.w0 + front_w0_halves(conv0_filters);
    p.xyz = xyz; p.xyz_bstride = xyz_batch_stride;
    p.out_image = out_image; p.out_image_bstride = out_image_batch_stride;
    p.out_fused = out_fused; p.out_fused_bstride = out_fused_batch_stride;
    p.amax_out_image = out_image_absmax; p.amax_out_fused = out_fused_absmax;
    p.N = n; p.Cin = image_channels; p.H = height; p.W = width;
    p.h = ceil_div(height, 2); p.w = ceil_div(width, 2);
    p.tilesX = ceil_div(p.w, FR_TW); p.tilesY = ceil_div(p.h, FR_TH);
    const long long tiles = (long long)p.tilesX * p.tilesY * n;
    if (tiles > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.ntiles = (int)tiles;
    p.slope0 = conv0_negative_slope; p.slope1 = kb_negative_slope;
    p.vec4 = !((p.w & 3) || (reinterpret_cast<uintptr_t>(out_image) & 15) || (reinterpret_cast<uintptr_t>(out_fused) & 15) ||
               (out_image_batch_stride & 3) || (out_fused_batch_stride & 3) || (reinterpret_cast<uintptr_t>(xyz) & 15) ||
               (xyz_batch_stride & 3)) ? 1 : 0;
    constexpr size_t lds = 2 * FR_NIN * 8 + 2 * 2 * FR_XP * 16 + (5 * 2 * 4 + 2 * 2) * 48 * 16;
    if (packed_next) {
        p.tab2 = static_cast<const float*>(packed_next);
        p.wn = reinterpret_cast<const _Float16*>(p.tab2 + front_next_tab_floats(next_filters));
        p.xyz2 = xyz_next; p.xyz2_bstride = xyz_next_batch_stride;
        p.out2 = out_next; p.out2_bstride = out_next_batch_stride;
        p.amax_out2 = out_next_absmax;
        p.h2 = ceil_div(p.h, 2); p.w2 = ceil_div(p.w, 2);
        p.slope2 = next_negative_slope;
        p.vec4_2 = !((p.w2 & 3) || (reinterpret_cast<uintptr_t>(out_next) & 15) || (out_next_batch_stride & 3) ||
                     (reinterpret_cast<uintptr_t>(xyz_next) & 15) || (xyz_next_batch_stride & 3)) ? 1 : 0;
    }
    const bool one_term = knob(KNOB_FP16_ONE_TERM) != 0;   // THROUGHPUT-ONLY: h1 w1 alone
    static DeviceOnce once[4];
    auto go = [&](auto kern, DeviceOnce& o) -> int {
        if (int rc = set_max_dynamic_lds(o, reinterpret_cast<const void*>(kern), 80 * 1024)) return rc;
        hipLaunchKernelGGL(kern, dim3(p.ntiles), dim3(FR_THREADS), lds, (hipStream_t)stream, p);
        return KBN_OK;
    };
    {
        if (int rc = packed_next ? (one_term ? go(kb1_front_kernel<3, 3, true, true>, once[3]) : go(kb1_front_kernel<3, 3, true, false>, once[2]))
                                 : (one_term ? go(kb1_front_kernel<3, 3, false, true>, once[1]) : go(kb1_front_kernel<3, 3, false, false>, once[0])))
            return rc;
    }
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_kb1_front_forward(const float* image, long long image_batch_stride, const void* packed_weight,
                          const float* xyz, long long xyz_batch_stride, float* out_image, long long out_image_batch_stride,
                          float* out_fused, long long out_fused_batch_stride, int n, int image_channels, int conv0_filters,
                          int kb_filters, int height, int width, float conv0_negative_slope, float kb_negative_slope,
                          unsigned* out_image_absmax, unsigned* out_fused_absmax, kbn_stream_t stream) {
    return kb1_front_launch(image, image_batch_stride, packed_weight, xyz, xyz_batch_stride, out_image, out_image_batch_stride, out_fused,
                            out_fused_batch_stride, n, image_channels, conv0_filters, kb_filters, height, width, conv0_negative_slope,
                            kb_negative_slope, out_image_absmax, out_fused_absmax, nullptr, nullptr, 0, nullptr, 0, 0, 0.f, nullptr, stream);
}

int kbn_kb1_front_next_forward(const float* image, long long image_batch_stride, const void* packed_weight,
                               const float* xyz, long long xyz_batch_stride, float* out_image, long long out_image_batch_stride,
                               float* out_fused, long long out_fused_batch_stride, int n, int image_channels, int conv0_filters,
                               int kb_filters, int height, int width, float conv0_negative_slope, float kb_negative_slope,
                               unsigned* out_image_absmax, unsigned* out_fused_absmax, const void* packed_next, const float* xyz_next,
                               long long xyz_next_batch_stride, float* out_next_fused, long long out_next_fused_batch_stride, int next_filters,
                               float next_negative_slope, unsigned* out_next_fused_absmax, kbn_stream_t stream) {
    if (!packed_next) return KBN_ERR_INVALID_ARGUMENT;
    return kb1_front_launch(image, image_batch_stride, packed_weight, xyz, xyz_batch_stride, out_image, out_image_batch_stride, out_fused,
                            out_fused_batch_stride, n, image_channels, conv0_filters, kb_filters, height, width, conv0_negative_slope,
                            kb_negative_slope, out_image_absmax, out_fused_absmax, packed_next, xyz_next, xyz_next_batch_stride,
                            out_next_fused, out_next_fused_batch_stride, next_filters, next_negative_slope, out_next_fused_absmax, stream);
}

static bool depth_front_shape_ok(int c_in, int f0, int fd) { return c_in >= 1 && c_in <= 8 && f0 == 16 && fd == 16; }

size_t kbn_kb1_depth_front_packed_weight_bytes(int depth_channels, int conv0_filters, int kb_filters) {
    if (!depth_front_shape_ok(depth_channels, conv0_filters, kb_filters)) return 0;
    return (size_t)kbn::DF_TAB * 4 + 2 * (3 * 2 * 64 * 8 + 5 * 2 * 64 * 8);
}

int kbn_kb1_depth_front_pack_weight(const float* w_conv0, const float* w_conv_depth, const float* w_proj, void* packed,
                                    int depth_channels, int conv0_filters, int kb_filters, kbn_stream_t stream) {
    using namespace kbn;
    if (!w_conv0 || !w_conv_depth || !w_proj || !packed) return KBN_ERR_INVALID_ARGUMENT;
    if (!depth_front_shape_ok(depth_channels, conv0_filters, kb_filters)) return KBN_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    float* tab = static_cast<float*>(packed);
    if (hipMemsetAsync(tab, 0, DF_TAB * 4, st) != hipSuccess) return KBN_ERR_LAUNCH;
    hipLaunchKernelGGL(depth_front_table_kernel, dim3(16, 2), dim3(256), 0, st, w_conv0, w_conv_depth, w_proj, tab, depth_channels);
    _Float16* p0 = reinterpret_cast<_Float16*>(tab + DF_TAB);
    hipLaunchKernelGGL(depth_front_pack_kernel, dim3((5 * 2 * 64 * 8 + 255) / 256), dim3(256), 0, st, w_conv0, w_conv_depth, tab, p0,
                       p0 + 3 * 2 * 64 * 8, depth_channels);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_kb1_depth_front_query(int depth_channels, int conv0_filters, int kb_filters, int height, int width, float conv0_negative_slope) {
    using namespace kbn;
    if (height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (!depth_front_shape_ok(depth_channels, conv0_filters, kb_filters) || knob(KNOB_NO_SPLIT)) return KBN_ERR_UNSUPPORTED;
    if ((long long)height * width > 0x1fffffffLL) return KBN_ERR_UNSUPPORTED;
    if (!(conv0_negative_slope >= 0.f && conv0_negative_slope <= 1.f)) return KBN_ERR_UNSUPPORTED;
    return KBN_OK;
}

}  // extern "C"

// fills the part of the parameters both forms of the depth front share; KBN_OK or a status
static int depth_front_params(kbn::DepthFrontParams& p, const float* kinv, const void* packed_weight, float* out_depth,
                              long long out_depth_batch_stride, float* xyz, long long xyz_batch_stride, int n, int depth_channels, int height,
                              int width, float conv0_negative_slope, float kb_negative_slope, int proj_activation, float proj_negative_slope,
                              unsigned* out_depth_absmax) {
    using namespace kbn;
    p.kinv = kinv;
    p.tab = static_cast<const float*>(packed_weight);
    p.w0 = reinterpret_cast<const _Float16*>(p.tab + DF_TAB);
    p.wc = p.w0 + 3 * 2 * 64 * 8;
    p.out_depth = out_depth; p.out_depth_bstride = out_depth_batch_stride;
    p.xyz = xyz; p.xyz_bstride = xyz_batch_stride;
    p.amax_out_depth = out_depth_absmax;
    p.N = n; p.Cin = depth_channels; p.H = height; p.W = width;
    p.h = ceil_div(height, 2); p.w = ceil_div(width, 2);
    p.tilesX = ceil_div(p.w, FR_TW); p.tilesY = ceil_div(p.h, FR_TH);
    const long long tiles = (long long)p.tilesX * p.tilesY * n;
    if (tiles > 0x7fffffffLL) return KBN_ERR_UNSUPPORTED;
    p.ntiles = (int)tiles;
    p.slope0 = conv0_negative_slope; p.slope1 = kb_negative_slope;
    p.act_proj = proj_activation ? 1 : 0; p.slope_proj = proj_negative_slope;
    p.vec4 = !((p.w & 3) || (reinterpret_cast<uintptr_t>(out_depth) & 15) || (out_depth_batch_stride & 3)) ? 1 : 0;
    return KBN_OK;
}

template <typename S2DCFG>
static int depth_front_launch(const kbn::DepthFrontParams& p, hipStream_t stream) {
    using namespace kbn;
    static DeviceOnce once;   // one per instantiation
    auto kern = kb1_depth_front_kernel<S2DCFG>;
    if (int rc = set_max_dynamic_lds(once, reinterpret_cast<const void*>(kern), 80 * 1024)) return rc;
    hipLaunchKernelGGL(kern, dim3(p.ntiles), dim3(FR_THREADS), DepthFrontLds<S2DCFG>::BYTES, stream, p);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

extern "C" {

int kbn_kb1_depth_front_forward(const float* depth, long long depth_batch_stride, const float* kinv, const void* packed_weight,
                                float* out_depth, long long out_depth_batch_stride, float* xyz, long long xyz_batch_stride, int n,
                                int depth_channels, int conv0_filters, int kb_filters, int height, int width,
                                float conv0_negative_slope, float kb_negative_slope, int proj_activation, float proj_negative_slope,
                                unsigned* out_depth_absmax, kbn_stream_t stream) {
    using namespace kbn;
    if (!depth || !kinv || !packed_weight || !out_depth || !xyz || n < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (int rc = kbn_kb1_depth_front_query(depth_channels, conv0_filters, kb_filters, height, width, conv0_negative_slope)) return rc;
    DepthFrontParams p{};
    p.depth = depth; p.depth_bstride = depth_batch_stride;
    if (int rc = depth_front_params(p, kinv, packed_weight, out_depth, out_depth_batch_stride, xyz, xyz_batch_stride, n, depth_channels, height,
                                    width, conv0_negative_slope, kb_negative_slope, proj_activation, proj_negative_slope, out_depth_absmax))
        return rc;
    if (knob(KNOB_FP16_ONE_TERM)) {   // THROUGHPUT-ONLY: h1 w1 alone
        static DeviceOnce once1;
        auto kern = kb1_depth_front_kernel<NoS2D, true>;
        if (int rc = set_max_dynamic_lds(once1, reinterpret_cast<const void*>(kern), 80 * 1024)) return rc;
        hipLaunchKernelGGL(kern, dim3(p.ntiles), dim3(FR_THREADS), DepthFrontLds<NoS2D>::BYTES, (hipStream_t)stream, p);
        KBN_CHECK_LAUNCH();
        return KBN_OK;
    }
    return depth_front_launch<NoS2D>(p, (hipStream_t)stream);
}

// ---- S2D -> conv0_depth -> KB1's depth branch in ONE launch (kb1_depth_front_kernel<pool preset>, s2d_stage.h) --------------
// which compiled pool preset (csrc/s2d_pools.h) the lists are: 0 KITTI, 1 VOID / NYUv2, 2 VOID training, -1 none
static int s2d_front_preset(const int* min_pool_sizes, int n_min, const int* max_pool_sizes, int n_max) {
    if (n_min < 0 || n_max < 0 || n_min + n_max > 8 || (n_min && !min_pool_sizes) || (n_max && !max_pool_sizes)) return -1;
    int k[8], np = 0;
    for (int i = 0; i < n_min; ++i) k[np++] = min_pool_sizes[i];
    for (int i = 0; i < n_max; ++i) k[np++] = max_pool_sizes[i];
    auto is = [&](int nm, std::initializer_list<int> ks) {
        if (n_min != nm || np != (int)ks.size()) return false;
        int i = 0;
        for (int v : ks)
            if (k[i++] != v) return false;
        return true;
    };
    if (is(5, {5, 7, 9, 11, 13, 15, 17})) return 0;
    if (is(2, {15, 17, 23, 27, 29})) return 1;
    if (is(3, {15, 17, 19, 23, 27})) return 2;
    return -1;
}

size_t kbn_s2d_depth_front_packed_weight_bytes(int n_pools) {
    if (n_pools < 1 || n_pools > 8) return 0;
    return (size_t)kbn::SF_TAB * 4 + 2 * (size_t)(kbn::SF_CHAIN_HALVES + kbn::SF_CONV_HALVES);
}

int kbn_s2d_depth_front_pack_weight(const float* w_pool_conv0, const float* w_pool_conv1, const float* w_pool_conv2, const float* w_conv,
                                    void* packed, int n_pools, kbn_stream_t stream) {
    using namespace kbn;
    if (!w_pool_conv0 || !w_pool_conv1 || !w_pool_conv2 || !w_conv || !packed) return KBN_ERR_INVALID_ARGUMENT;
    if (n_pools < 1 || n_pools > 8) return KBN_ERR_UNSUPPORTED;
    float* tab = static_cast<float*>(packed);
    _Float16* wchain = reinterpret_cast<_Float16*>(tab + SF_TAB);
    hipLaunchKernelGGL(s2d_stage_pack_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w_pool_conv0, w_pool_conv1, w_pool_conv2, w_conv, tab,
                       wchain, wchain + SF_CHAIN_HALVES, n_pools);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_s2d_depth_front_query(int input_channels, const int* min_pool_sizes, int n_min, const int* max_pool_sizes, int n_max,
                              int n_convolution, int n_filter, int conv0_filters, int kb_filters, int height, int width,
                              float s2d_negative_slope, float conv0_negative_slope) {
    using namespace kbn;
    if (height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (knob(KNOB_NO_DEPTH_FRONT_FUSION)) return KBN_ERR_UNSUPPORTED;
    if (input_channels != 2 || n_convolution != 3 || n_filter != 8) return KBN_ERR_UNSUPPORTED;   // [sparse depth, validity] -> KBNet's S2D
    if (s2d_front_preset(min_pool_sizes, n_min, max_pool_sizes, n_max) < 0) return KBN_ERR_UNSUPPORTED;
    if (!(s2d_negative_slope >= 0.f && s2d_negative_slope <= 1.f)) return KBN_ERR_UNSUPPORTED;
    return kbn_kb1_depth_front_query(n_filter, conv0_filters, kb_filters, height, width, conv0_negative_slope);
}

int kbn_s2d_depth_front_forward(const float* x, long long x_batch_stride, const float* kinv, const void* packed_s2d, const void* packed_weight,
                                float* out_depth, long long out_depth_batch_stride, float* xyz, long long xyz_batch_stride, int n,
                                int input_channels, const int* min_pool_sizes, int n_min, const int* max_pool_sizes, int n_max,
                                int n_convolution, int n_filter, int conv0_filters, int kb_filters, int height, int width,
                                float s2d_negative_slope, float conv0_negative_slope, float kb_negative_slope, int proj_activation,
                                float proj_negative_slope, unsigned* out_depth_absmax, kbn_stream_t stream) {
    using namespace kbn;
    if (!x || !kinv || !packed_s2d || !packed_weight || !out_depth || !xyz || n < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (int rc = kbn_s2d_depth_front_query(input_channels, min_pool_sizes, n_min, max_pool_sizes, n_max, n_convolution, n_filter, conv0_filters,
                                           kb_filters, height, width, s2d_negative_slope, conv0_negative_slope))
        return rc;
    DepthFrontParams p{};
    if (int rc = depth_front_params(p, kinv, packed_weight, out_depth, out_depth_batch_stride, xyz, xyz_batch_stride, n, n_filter, height, width,
                                    conv0_negative_slope, kb_negative_slope, proj_activation, proj_negative_slope, out_depth_absmax))
        return rc;
    p.s2d.x = x; p.s2d.x_bstride = x_batch_stride;
    p.s2d.tab = static_cast<const float*>(packed_s2d);
    p.s2d.wchain = reinterpret_cast<const _Float16*>(p.s2d.tab + SF_TAB);
    p.s2d.wconv = p.s2d.wchain + SF_CHAIN_HALVES;
    p.s2d.slope = s2d_negative_slope;
    p.s2d.dbg = knob(KNOB_S2D_DEBUG);
    switch (s2d_front_preset(min_pool_sizes, n_min, max_pool_sizes, n_max)) {
        case 0: return depth_front_launch<KittiPools>(p, (hipStream_t)stream);
        case 1: return depth_front_launch<VoidPools>(p, (hipStream_t)stream);
        default: return depth_front_launch<VoidTrainPools>(p, (hipStream_t)stream);
    }
}

}  // extern "C"
