// Micro-benchmark: what do non-MFMA instructions cost when they sit between the MFMAs of an in-order
// wave?  One workgroup per CU, NW waves per SIMD, a loop of 64 v_mfma_f32_16x16x4_f32 (16 independent
// accumulators x 4) with NV v_add, NS s_add and NL ds_read_b32 after every MFMA, order pinned.
// Prints cycles per MFMA slot per SIMD (32 = matrix pipe bound).
// Build: hipcc --offload-arch=gfx950 -O3 issue_probe.hip -o issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NS, int NL, int NW2>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    __shared__ float lds[4096];
    for (int e = threadIdx.x; e < 4096; e += blockDim.x) lds[e] = e * 1e-4f;
    __syncthreads();
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = i + threadIdx.x;
    int s = blockIdx.x;
    float l[4] = {0, 0, 0, 0};
    const unsigned lp = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)(lds + (threadIdx.x & 63));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 15], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(i * NV + k) & 7]) : "v"(b));
#pragma unroll
            for (int k = 0; k < NS; ++k) asm volatile("s_add_i32 %0, %0, 1" : "+s"(s));
#pragma unroll
            for (int k = 0; k < NL; ++k) asm volatile("ds_read_b32 %0, %1" : "=v"(l[(i * NL + k) & 3]) : "v"(lp));
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) r += v[i];
    for (int i = 0; i < 4; ++i) r += l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + s;
}

template <int NV, int NS, int NL>
static void run(int nw) {
    float* out;
    hipMalloc(&out, 4 * 512 * 256);
    const int iters = 2000;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    auto launch = [&] { hipLaunchKernelGGL((probe<NV, NS, NL, 0>), dim3(256), dim3(256 * nw), 0, 0, out, iters); };
    launch(); hipDeviceSynchronize();
    hipEventRecord(s);
    launch();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double slots = (double)iters * 64 * nw;  // MFMA slots per SIMD
    printf("waves/SIMD=%d  per MFMA: %d valu %d salu %d ds_read : %7.1f us  %6.1f ns/slot  %5.1f cycles/slot @2.4GHz  %6.1f TFLOP/s\n",
           nw, NV, NS, NL, ms * 1e3, ms * 1e6 / slots, ms * 1e6 / slots * 2.4, 2048.0 * slots * 1024 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int nw : {1, 2}) {
        run<0, 0, 0>(nw);
        run<1, 0, 0>(nw);
        run<2, 0, 0>(nw);
        run<4, 0, 0>(nw);
        run<6, 0, 0>(nw);
        run<0, 2, 0>(nw);
        run<0, 4, 0>(nw);
        run<0, 0, 1>(nw);
        run<0, 0, 2>(nw);
        run<2, 2, 1>(nw);
        run<3, 3, 1>(nw);
    }
    return 0;
}
