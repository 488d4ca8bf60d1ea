// s2d_hpass_probe.hip -- the horizontal min/max pass of the S2D pyramid (reference src/networks.py:2175-2186: max_pool2d of
// the sparse depth with kernel sizes 5 .. 17), two ways, on an LDS-resident tile of vertical-pass results:
//   blocked  what s2d_kernel does (csrc/s2d.hip, P3): a thread owns 4 consecutive pixels of a row, reads the 2r + 4 values it
//            needs as 16-byte LDS words and forms the four windows from a shared core + prefix / suffix minima
//            (2r + 5 compares per 4 pixels and pool)
//   dpp      north_star's wavefront shuffles: a lane owns a 4-column block (ONE 16-byte LDS read per pool), a 16-lane DPP row
//            covers 64 columns; the blocks a window overlaps come from the neighbouring lanes by v_mov_b32_dpp row_shr / row_shl
//            (whole-block minima of lanes l -+ 1, l -+ 2, suffix / prefix minima of the farthest block); lanes 0, 1, 14, 15 of a
//            row are halo (48 of 64 columns valid for r <= 8)
// Both produce the same values (checked), the KITTI pool list (r = 2..6 min, 7, 8 max), 18 rows x 64 z-columns per wave.
// build: hipcc --offload-arch=gfx950 -O3 s2d_hpass_probe.hip -o s2d_hpass_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NPOOL = 7, ROWS = 18, COLS = 64 + 16, PITCH = 96;     // z columns a tile of 64 outputs needs for R = 8, padded rows
__device__ __host__ constexpr int radius(int pi) { return pi + 2; }  // 2, 3, 4, 5, 6, 7, 8

template <bool IS_MIN> __device__ __forceinline__ float mn(float a, float b) { return IS_MIN ? fminf(a, b) : fmaxf(a, b); }

// ---- blocked: thread = (row, 4-column group q): outputs columns 4q .. 4q + 3, window of output x = z columns x + (8 - r) .. x + 8 + r
template <int PI>
__device__ __forceinline__ f4 blocked_pool(const float* V, int row, int q) {
    constexpr int r = radius(PI), OFF = 8 - r, A0 = OFF & ~3, SH = OFF & 3, NV = 2 * r + 4, NB = (SH + NV + 3) / 4;
    constexpr bool IS_MIN = PI < 5;
    f4 w[NB];
    const float* s = V + (PI * ROWS + row) * PITCH + 4 * q + A0;
#pragma unroll
    for (int m = 0; m < NB; ++m) w[m] = *reinterpret_cast<const f4*>(s + 4 * m);
    auto v = [&](int o) { return w[(SH + o) >> 2][(SH + o) & 3]; };
    float core = v(3);
#pragma unroll
    for (int o = 4; o <= 2 * r; ++o) core = mn<IS_MIN>(core, v(o));
    const float l1 = mn<IS_MIN>(v(1), v(2)), l0 = mn<IS_MIN>(v(0), l1);
    const float h2 = mn<IS_MIN>(v(2 * r + 1), v(2 * r + 2)), h3 = mn<IS_MIN>(h2, v(2 * r + 3));
    return (f4){mn<IS_MIN>(core, l0), mn<IS_MIN>(core, mn<IS_MIN>(l1, v(2 * r + 1))), mn<IS_MIN>(core, mn<IS_MIN>(v(2), h2)), mn<IS_MIN>(core, h3)};
}

// ---- dpp: lane l of a 16-lane row owns z columns 4l .. 4l + 3 of a 64-column segment; output x (centre column) = min over
// columns x - r .. x + r; valid for lanes 2 .. 13
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int D> __device__ __forceinline__ float from_left(float v) { return dpp<0x110 + D>(v); }    // row_shr:D: lane l gets lane l - D
template <int D> __device__ __forceinline__ float from_right(float v) { return dpp<0x100 + D>(v); }   // row_shl:D: lane l gets lane l + D

template <int PI>
__device__ __forceinline__ f4 dpp_pool(const float* V, int row, int seg_col0, int l16) {
    constexpr int r = radius(PI);
    constexpr bool IS_MIN = PI < 5;
    const f4 v = *reinterpret_cast<const f4*>(V + (PI * ROWS + row) * PITCH + seg_col0 + 4 * l16);
    // prefix / suffix minima of the own block
    const float p2 = mn<IS_MIN>(v[0], v[1]), p3 = mn<IS_MIN>(p2, v[2]), full = mn<IS_MIN>(p3, v[3]);
    const float s2 = mn<IS_MIN>(v[3], v[2]), s3 = mn<IS_MIN>(s2, v[1]);
    const float P[5] = {0.f, v[0], p2, p3, full}, S[5] = {0.f, v[3], s2, s3, full};   // P[b]: first b values, S[b]: last b values
    f4 out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // own block: columns max(0, j - r) .. min(3, j + r)
        float a;
        {
            constexpr int dummy = 0; (void)dummy;
            const int lo = j - r > 0 ? j - r : 0, hi = j + r < 3 ? j + r : 3;
            a = v[lo];
            for (int c = lo + 1; c <= hi; ++c) a = mn<IS_MIN>(a, v[c]);
        }
        const int L = r - j, Rr = r - (3 - j);   // columns the window reaches beyond the block on either side
        if (L > 0) {
            const int fa = L / 4, fb = L % 4;
            if (fa >= 1) a = mn<IS_MIN>(a, from_left<1>(full));
            if (fa >= 2) a = mn<IS_MIN>(a, from_left<2>(full));
            if (fb > 0) a = mn<IS_MIN>(a, fa == 0 ? from_left<1>(S[fb]) : (fa == 1 ? from_left<2>(S[fb]) : from_left<3>(S[fb])));
        }
        if (Rr > 0) {
            const int fa = Rr / 4, fb = Rr % 4;
            if (fa >= 1) a = mn<IS_MIN>(a, from_right<1>(full));
            if (fa >= 2) a = mn<IS_MIN>(a, from_right<2>(full));
            if (fb > 0) a = mn<IS_MIN>(a, fa == 0 ? from_right<1>(P[fb]) : (fa == 1 ? from_right<2>(P[fb]) : from_right<3>(P[fb])));
        }
        out[j] = a;
    }
    return out;
}

template <int I, int N, typename F> __device__ __forceinline__ void unroll(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); unroll<I + 1, N>(static_cast<F&&>(f)); }
}

// MODE 0 blocked, 1 dpp.  One workgroup of 256 threads; tile = 18 rows x 64 output columns x 7 pools (blocked: 288 items of 4
// pixels; dpp: 18 rows x 21.3 lanes ... rows are packed 4 per wave: a wave = 4 rows x 16 lanes, 48 valid columns per row)
template <int MODE>
__global__ __launch_bounds__(256) void hpass(const float* src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float V[NPOOL * ROWS * PITCH];
    for (int i = threadIdx.x; i < NPOOL * ROWS * PITCH; i += 256) V[i] = src[i];
    __syncthreads();
    const int tid = threadIdx.x;
    f4 acc[NPOOL];
    for (int pi = 0; pi < NPOOL; ++pi) acc[pi] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
        if (MODE == 0) {
            // 18 rows x 16 groups = 288 items over 256 threads: two rounds, the second 1/8 full (as in s2d_kernel: 306 items)
            for (int t = tid; t < ROWS * 16; t += 256) {
                const int row = t >> 4, q = t & 15;
                unroll<0, NPOOL>([&](auto pic) {
                    constexpr int pi = decltype(pic)::value;
                    const f4 o = blocked_pool<pi>(V, row, q);
                    acc[pi] += o;
                });
            }
        } else {
            // a 16-lane row yields 48 columns: 64 output columns = 1.33 segments; items = (row, segment of 48): 18 x 2 = 36
            // row-segments of 16 lanes = 576 lanes over 256 threads: three rounds (the same 64 columns + 32 spare per row)
            for (int t = tid; t < ROWS * 2 * 16; t += 256) {
                const int rs = t >> 4, l16 = t & 15, row = rs >> 1, seg = rs & 1;
                unroll<0, NPOOL>([&](auto pic) {
                    constexpr int pi = decltype(pic)::value;
                    const f4 o = dpp_pool<pi>(V, row, seg * 16, l16);   // second segment starts 16 columns on (overlap: probe only)
                    acc[pi] += (l16 >= 2 && l16 < 14) ? o : (f4){0.f, 0.f, 0.f, 0.f};
                });
            }
        }
    }
    float s = 0.f;
    for (int pi = 0; pi < NPOOL; ++pi) s += acc[pi][0] + acc[pi][1] + acc[pi][2] + acc[pi][3];
    out[blockIdx.x * 256 + tid] = s;
}

// value check: both forms on the same row against a scalar window minimum
__global__ void check(const float* src, int* bad) {
    __shared__ __attribute__((aligned(16))) float V[NPOOL * ROWS * PITCH];
    for (int i = threadIdx.x; i < NPOOL * ROWS * PITCH; i += 64) V[i] = src[i];
    __syncthreads();
    const int l = threadIdx.x, l16 = l & 15, row = 3 + (l >> 4);
    unroll<0, NPOOL>([&](auto pic) {
        constexpr int pi = decltype(pic)::value;
        constexpr int r = radius(pi);
        constexpr bool IS_MIN = pi < 5;
        const f4 d = dpp_pool<pi>(V, row, 0, l16);                 // centre columns 4 l16 + j of the segment
        // blocked form: output x = window x + 8 - r .. x + 8 + r, i.e. centre column x + 8: group q covers centres 4q + 8 ..
        const f4 b = blocked_pool<pi>(V, row, l16);
        for (int j = 0; j < 4; ++j) {
            const int cd = 4 * l16 + j, cb = 4 * l16 + j + 8;
            float rd = V[(pi * ROWS + row) * PITCH + cd], rb = V[(pi * ROWS + row) * PITCH + cb];
            for (int o = -r; o <= r; ++o) {
                if (cd + o >= 0) rd = mn<IS_MIN>(rd, V[(pi * ROWS + row) * PITCH + cd + o]);
                rb = mn<IS_MIN>(rb, V[(pi * ROWS + row) * PITCH + cb + o]);
            }
            if (l16 >= 2 && l16 < 14 && d[j] != rd) atomicAdd(bad, 1);
            if (b[j] != rb) atomicAdd(bad + 1, 1);
        }
    });
}

int main() {
    const int n = NPOOL * ROWS * PITCH;
    std::vector<float> h(n);
    srand(3);
    for (auto& v : h) v = (rand() % 1000) * 0.01f;
    float *src, *out; int* bad;
    (void)hipMalloc(&src, n * 4); (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&bad, 8);
    (void)hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, src, bad);
    int hb[2]; (void)hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    printf("value check: dpp mismatches %d, blocked mismatches %d\n", hb[0], hb[1]);
    int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000, blocks = cus * 8;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(hpass<0>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
            else hipLaunchKernelGGL(hpass<1>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
            // outputs per iteration and workgroup: 18 rows x 64 columns x 7 pools (dpp: 18 x 2 x 48 = 1728 columns for 1152 needed)
            const double px = (double)blocks * iters * ROWS * 64;
            printf("%-8s: %8.2f ms  %6.2f ps per output pixel (7 pools) -> %5.1f us per 8 KITTI frames' feature pixels (3.9 M incl. halo)\n",
                   mode == 0 ? "blocked" : "dpp", ms, ms * 1e9 / px, ms * 1e3 / px * 3.9e6);
        }
    }
    return 0;
}
