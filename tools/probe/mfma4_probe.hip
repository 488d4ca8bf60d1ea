// Micro-benchmark for the S2D redesign: v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products per
// instruction: a lane contributes ONE A value and ONE B value, and gets 4 results -- D[i][lane] +=
// A[4*(lane/4)+i] * B[lane] -- i.e. "4 FMAs per lane" on the matrix pipe).
//   (1) layout check against the formula above,
//   (2) issue rate with independent accumulators, dependent-accumulator latency,
//   (3) a matrix-only wave and a VALU-only wave sharing a SIMD: do both run at their own rate?
// Build: hipcc --offload-arch=gfx950 -O3 mfma4_probe.hip -o mfma4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), (float)(100 * (l + 1)), acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[i * 64 + l] = acc[i];
}

// mode bit 0: waves 0-3 run the MFMA loop; bit 1: waves 4-7 run the VALU loop; bit 2: waves 4-7 run an LDS-read loop
template <int NACC>
__global__ __launch_bounds__(512) void mix(float* out, int iters, int mode, unsigned long long* cyc) {
    __shared__ float lds[8192];
    for (int e = threadIdx.x; e < 8192; e += blockDim.x) lds[e] = e * 1e-4f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float r = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (mode & 1) {
            f32x4 acc[NACC];
            for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
            float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 64; ++i) {
                    acc[i % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i % NACC], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else {
        if (mode & 2) {
            float v[8];
            for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
            const float b = 1.0001f, c = 0.5f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 64; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(b), "v"(c));
            }
            for (int i = 0; i < 8; ++i) r += v[i];
        }
        if (mode & 4) {
            float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const unsigned lp = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)(lds + (threadIdx.x & 63));
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 64; ++i)
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(l[i & 7]) : "v"(lp), "n"((i & 31) * 256));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            for (int i = 0; i < 8; ++i) r += l[i];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC>
static void run(int mode, const char* what) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 4 * 512 * 256);
    hipMalloc(&cyc, 8 * 8);
    hipMemset(cyc, 0, 64);
    const int iters = 2000;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    auto launch = [&] { hipLaunchKernelGGL((mix<NACC>), dim3(256), dim3(512), 0, 0, out, iters, mode, cyc); };
    launch(); hipDeviceSynchronize();
    hipEventRecord(s);
    launch();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double n = (double)iters * 64;
    printf("%-44s acc=%2d  %8.1f us   wave0 %6.2f clk/instr   wave4 %6.2f clk/instr   (wall %5.2f ns per wave-instr)\n", what, NACC,
           ms * 1e3, h[0] / n, h[4] / n, ms * 1e6 / n);
    hipFree(out); hipFree(cyc);
}

int main() {
    float* out; hipMalloc(&out, 4 * 256);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
    float h[256]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 4; ++i)
        for (int l = 0; l < 64; ++l) {
            const float want = (float)(4 * (l / 4) + i + 1) * (float)(100 * (l + 1));
            if (h[i * 64 + l] != want) { if (bad < 4) printf("layout mismatch i=%d lane=%d got %g want %g\n", i, l, h[i * 64 + l], want); ++bad; }
        }
    printf("layout D[i][lane] = A[4*(lane/4)+i] * B[lane]: %s\n", bad ? "MISMATCH" : "ok");
    run<16>(1, "mfma 4x4x1 alone, 16 independent acc");
    run<8>(1, "mfma 4x4x1 alone");
    run<4>(1, "mfma 4x4x1 alone");
    run<2>(1, "mfma 4x4x1 alone");
    run<1>(1, "mfma 4x4x1 alone, dependent chain");
    run<8>(2, "valu fma alone (waves 4-7)");
    run<8>(3, "mfma (waves 0-3) + valu (waves 4-7)");
    run<8>(4, "ds_read_b32 alone (waves 4-7)");
    run<8>(5, "mfma (waves 0-3) + ds_read (waves 4-7)");
    run<8>(7, "mfma + valu + ds_read (waves 4-7 do both)");
    return 0;
}
