// fp32 products on the bf16 matrix core: a = a1 + a2 + a3 (three bf16 terms, exact), b likewise, and a*b taken as a sum
// of bf16 x bf16 MFMA products with fp32 accumulation.  Two questions:
//   (1) ACCURACY: error of a K-long dot product vs an fp64 evaluation, for the fp32 MFMA chain (== fmaf chain) and for
//       3 / 6 / 9 product terms, one accumulator or the small terms kept apart.
//   (2) CONCURRENCY: does a wave of bf16 MFMAs share issue time with a wave of v_fma_f32 on the same SIMD (fp32 MFMAs
//       do: tools/probe/valu_probe.hip), or do the two overlap?
// Build: hipcc --offload-arch=gfx950 -O3 bf16x_probe.hip -o bf16x_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static unsigned short bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// parts: [3][rows][K] bf16.  One wave computes a 32 x 32 block over K; `terms` lists (i, j) pairs of parts; pairs with
// index >= split go to a second accumulator that is added at the end.
__global__ void gemm_terms(const unsigned short* A, const unsigned short* B, float* C, int K, const int* terms, int nterms, int split) {
    const int lane = threadIdx.x, r = lane & 31, kg = lane >> 5;
    f32x16 hi = {}, lo = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int t = 0; t < nterms; ++t) {
            const int i = terms[2 * t], j = terms[2 * t + 1];
            const bf16x8 a = *(const bf16x8*)(A + ((size_t)i * 32 + r) * K + k0 + kg * 8);
            const bf16x8 b = *(const bf16x8*)(B + ((size_t)j * 32 + r) * K + k0 + kg * 8);
            if (t < split) hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, hi, 0, 0, 0);
            else lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, lo, 0, 0, 0);
        }
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i / 4) * 8 + kg * 4 + (i % 4);
        C[row * 32 + r] = hi[i] + lo[i];
    }
}

// fp16 scheme: a' = a 2^-6 = h1 + 2^-11 h2 (h2 = fp16((a' - h1) 2^11)), w' = w 2^e = w1 + w2; products h1 w1 + h1 w2 + h2 (w1 2^-11):
// THREE MFMAs per product, every term in fp16's normal range whatever the magnitude of a.  A: [2][32][K], B: [3][32][K] fp16.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void gemm_f16x3(const _Float16* A, const _Float16* B, float* C, int K, float unscale) {
    const int lane = threadIdx.x, r = lane & 31, kg = lane >> 5;
    f32x16 acc = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        const f16x8 h1 = *(const f16x8*)(A + ((size_t)0 * 32 + r) * K + k0 + kg * 8), h2 = *(const f16x8*)(A + ((size_t)1 * 32 + r) * K + k0 + kg * 8);
        const f16x8 w1 = *(const f16x8*)(B + ((size_t)0 * 32 + r) * K + k0 + kg * 8), w2 = *(const f16x8*)(B + ((size_t)1 * 32 + r) * K + k0 + kg * 8);
        const f16x8 w1s = w1 * (_Float16)0.00048828125f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, w1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, w2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, w1s, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) C[((i / 4) * 8 + kg * 4 + (i % 4)) * 32 + r] = acc[i] * unscale;
}

// fp32 MFMA chain over the same data (A: [32][K], B: [32][K] fp32), 16x16x4.
__global__ void gemm_f32(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
    for (int mb = 0; mb < 2; ++mb)
        for (int nb = 0; nb < 2; ++nb) {
            f32x4 acc = {};
            for (int k0 = 0; k0 < K; k0 += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(size_t)(mb * 16 + r) * K + k0 + kq], B[(size_t)(nb * 16 + r) * K + k0 + kq], acc, 0, 0, 0);
            for (int i = 0; i < 4; ++i) C[(mb * 16 + kq * 4 + i) * 32 + nb * 16 + r] = acc[i];
        }
}

// concurrency: waves [0, nm) of a workgroup issue bf16 MFMAs, the others v_fma_f32.
__global__ __launch_bounds__(512) void mix(float* out, int iters, int nm_per_simd, int nv_per_simd, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6;          // waves are dealt round robin to the 4 SIMDs
    const int slot = wave >> 2;                 // index of this wave on its SIMD
    const bool is_m = slot < nm_per_simd;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
    f32x16 acc[2] = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)1.0f; }
    const float bb = 1.0001f, cc = 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (is_m) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 1], 0, 0, 0);
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 128; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(bb), "v"(cc));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float rsum = 0;
    for (int i = 0; i < 8; ++i) rsum += v[i];
    for (int i = 0; i < 16; ++i) rsum += acc[0][i] + acc[1][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = rsum;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    (void)nv_per_simd;
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static float gauss() { return (float)(sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand())); }

int main() {
    srand(7);
    const int Ks[] = {576, 2304, 6912};
    for (int kc = 0; kc < 3; ++kc) {
        const int K = Ks[kc];
        std::vector<float> A(32 * K), B(32 * K);
        for (auto& x : A) { x = gauss(); if (x < 0) x *= 0.2f; }        // post-LeakyReLU-like activations
        for (auto& x : B) x = gauss() / sqrtf((float)K);
        std::vector<unsigned short> Ap(3 * 32 * K), Bp(3 * 32 * K);
        auto split = [&](const std::vector<float>& src, std::vector<unsigned short>& dst) {
            for (int i = 0; i < 32 * K; ++i) {
                float r = src[i];
                for (int p = 0; p < 3; ++p) { const unsigned short h = bf16_rne(r); dst[(size_t)p * 32 * K + i] = h; r -= bf16_f(h); }
                if (r != 0.0f && fabsf(src[i]) > 1e-30f) { printf("split residual %g of %g\n", r, src[i]); }
            }
        };
        split(A, Ap); split(B, Bp);
        std::vector<double> ref(32 * 32);
        std::vector<float> chain(32 * 32);
        double rms = 0;
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                double s = 0; float c = 0;
                for (int k = 0; k < K; ++k) { s += (double)A[m * K + k] * (double)B[n * K + k]; c = fmaf(A[m * K + k], B[n * K + k], c); }
                ref[m * 32 + n] = s; chain[m * 32 + n] = c; rms += s * s;
            }
        rms = sqrt(rms / 1024);
        auto report = [&](const char* what, const float* c) {
            double mx = 0, sq = 0, bias = 0;
            for (int i = 0; i < 1024; ++i) { const double e = (c[i] - ref[i]) / rms; mx = fmax(mx, fabs(e)); sq += e * e; bias += e; }
            printf("K=%5d %-46s max %.3e  rms %.3e  mean %+.3e (of the output's rms)\n", K, what, mx, sqrt(sq / 1024), bias / 1024);
        };
        report("host fmaf chain, k ascending", chain.data());
        unsigned short *dA, *dB; float *dAf, *dBf, *dC; int* dT;
        (void)hipMalloc(&dA, Ap.size() * 2); (void)hipMalloc(&dB, Bp.size() * 2); (void)hipMalloc(&dC, 4096); (void)hipMalloc(&dT, 128);
        (void)hipMalloc(&dAf, A.size() * 4); (void)hipMalloc(&dBf, B.size() * 4);
        (void)hipMemcpy(dA, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dB, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(dAf, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dBf, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> C(1024);
        hipLaunchKernelGGL(gemm_f32, dim3(1), dim3(64), 0, 0, dAf, dBf, dC, K);
        (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        report("v_mfma_f32_16x16x4_f32 chain", C.data());
        struct Case { const char* what; int n, split; int t[18]; };
        const Case cases[] = {
            {"bf16 x1  (a1b1)", 1, 1, {0, 0}},
            {"bf16 x3  (a1b1 a1b2 a2b1)", 3, 3, {0, 0, 0, 1, 1, 0}},
            {"bf16 x6  one accumulator, big term first", 6, 6, {0, 0, 0, 1, 1, 0, 1, 1, 0, 2, 2, 0}},
            {"bf16 x6  one accumulator, small terms first", 6, 6, {0, 2, 2, 0, 1, 1, 0, 1, 1, 0, 0, 0}},
            {"bf16 x6  a1b1 apart from the five small terms", 6, 1, {0, 0, 0, 1, 1, 0, 1, 1, 0, 2, 2, 0}},
            {"bf16 x6  {a1b1,a1b2,a2b1} | {a2b2,a1b3,a3b1}", 6, 3, {0, 0, 0, 1, 1, 0, 1, 1, 0, 2, 2, 0}},
            {"bf16 x9  one accumulator", 9, 9, {0, 0, 0, 1, 1, 0, 1, 1, 0, 2, 2, 0, 1, 2, 2, 1, 2, 2}},
            {"bf16 x9  a1b1 apart", 9, 1, {0, 0, 0, 1, 1, 0, 1, 1, 0, 2, 2, 0, 1, 2, 2, 1, 2, 2}},
        };
        for (const Case& c : cases) {
            (void)hipMemcpy(dT, c.t, sizeof(int) * 2 * c.n, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(gemm_terms, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, dT, c.n, c.split);
            (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
            report(c.what, C.data());
        }
        for (int amag = 0; amag < 3; ++amag) {   // activations as drawn, x 1e-4 (fp16 subnormal range without the scaled residual), x 1e4
            const float am = amag == 0 ? 1.f : (amag == 1 ? 1e-4f : 1e4f);
            float wmax = 0; for (auto x : B) wmax = fmaxf(wmax, fabsf(x));
            int ex; frexpf(wmax, &ex); const float wscale = ldexpf(1.f, 13 - ex);
            std::vector<_Float16> Ah(2 * 32 * K), Bh(2 * 32 * K);
            for (int i = 0; i < 32 * K; ++i) {
                const float a = A[i] * am, ap = a * 0.015625f;
                const _Float16 h1 = (_Float16)ap;
                Ah[i] = h1; Ah[32 * K + i] = (_Float16)(fmaf((float)h1, -2048.f, a * 32.f));
                const float wp = B[i] * wscale; const _Float16 w1 = (_Float16)wp;
                Bh[i] = w1; Bh[32 * K + i] = (_Float16)(wp - (float)w1);
            }
            _Float16 *dAh, *dBh;
            (void)hipMalloc(&dAh, Ah.size() * 2); (void)hipMalloc(&dBh, Bh.size() * 2);
            (void)hipMemcpy(dAh, Ah.data(), Ah.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dBh, Bh.data(), Bh.size() * 2, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(gemm_f16x3, dim3(1), dim3(64), 0, 0, dAh, dBh, dC, K, 64.f / wscale / am);
            (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
            char what[96]; snprintf(what, sizeof what, "fp16 x3 scaled residual, activations x %g", am);
            report(what, C.data());
            (void)hipFree(dAh); (void)hipFree(dBh);
        }
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dT); (void)hipFree(dAf); (void)hipFree(dBf);
    }
    // concurrency
    float* out; unsigned long long *cyc, h[8];
    (void)hipMalloc(&out, 4 * 512 * 256); (void)hipMalloc(&cyc, 64);
    const int iters = 2000;
    const int combos[][2] = {{1, 0}, {0, 1}, {1, 1}, {2, 0}, {0, 2}};
    for (auto& c : combos) {
        const int nm = c[0], nv = c[1];
        hipLaunchKernelGGL(mix, dim3(256), dim3(256 * (nm + nv)), 0, 0, out, iters, nm, nv, cyc);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("per SIMD: %d bf16-MFMA wave(s) + %d v_fma wave(s):", nm, nv);
        if (nm) printf("  MFMA wave %.1f clk per v_mfma_f32_32x32x16_bf16", (double)h[0] / (iters * 16.0));
        if (nv) printf("  VALU wave %.2f clk per v_fma_f32", (double)h[4 * nm] / (iters * 128.0));
        printf("\n");
    }
    return 0;
}
