// Micro-benchmark: what does the LDS -> v_mfma_f32_16x16x4_f32 loop shape of conv_dma_kernel
// deliver without any staging?  Build: hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#ifndef RANDOM
#define RANDOM 1
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// V0: registers only, 16 independent accumulators
__global__ __launch_bounds__(256) void v0(float* out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = RANDOM ? __sinf(threadIdx.x * 12.9898f) : threadIdx.x * 0.001f, b = RANDOM ? __cosf(threadIdx.x * 78.233f) : 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 18; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V1: the conv kernel's k-step: MW A reads + NB B reads (ds_read_b32) -> MW*NB MFMAs, 18 k-steps per "chunk",
// a barrier per chunk (BAR=1) or none.
template <int MW, int NB, int BAR, int PLANE>
__global__ __launch_bounds__(256) void v1(float* out, int iters) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    for (int e = tid; e < 8 * PLANE + 8 * 9 * 64; e += 256) { unsigned h = (e + 1u) * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; smem[e] = RANDOM ? ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x800000) : e * 1e-6f; }
    __syncthreads();
    f32x4 acc[MW][NB];
    for (int i = 0; i < MW; ++i)
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    int mbase[MW];
    for (int mi = 0; mi < MW; ++mi) mbase[mi] = ((wave * MW + mi) / 2) * 40 + ((wave * MW + mi) & 1) * 16 + li + 3 + lk * PLANE;
    const int boff = (lk >> 1) * 2 * 64 + li * 2 + (lk & 1);
    const float* As = smem;
    const float* Bs = smem + 8 * PLANE;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * 40 + tap % 3;
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4) {
                const float* Ab = As + c4 * 4 * PLANE + toff;
                const float* Bb = Bs + (tap * 2 + c4) * 4 * 64 + boff;
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
        if (BAR) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < MW; ++i)
        for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V2: same data flow on v_mfma_f32_32x32x2_f32: wave tile 64 px x 64 ch = 2x2 blocks of 32x32
template <int BAR, int PLANE>
__global__ __launch_bounds__(256) void v2(float* out, int iters) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, lk = lane >> 5;
    for (int e = tid; e < 8 * PLANE + 8 * 9 * 64; e += 256) { unsigned h = (e + 1u) * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; smem[e] = RANDOM ? ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x800000) : e * 1e-6f; }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    int mbase[2];
    for (int mi = 0; mi < 2; ++mi) mbase[mi] = (wave * 2 + mi) * 40 + l32 + 3 + lk * PLANE;
    const int boff = lk * 64 + l32;
    const float* As = smem;
    const float* Bs = smem + 8 * PLANE;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * 40 + tap % 3;
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                const float* Ab = As + c2 * 2 * PLANE + toff;
                const float* Bb = Bs + (tap * 4 + c2) * 2 * 64 + boff;
                float a[2], b[2];
                a[0] = Ab[mbase[0]]; a[1] = Ab[mbase[1]];
                b[0] = Bb[0]; b[1] = Bb[32];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[nb], acc[mi][nb], 0, 0, 0);
            }
        }
        if (BAR) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, double flop_per_block_iter, int blocks, int iters) {
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    launch(); hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    ms /= 3;
    printf("%-44s blocks=%5d  %8.1f us  %7.1f TFLOP/s\n", name, blocks, ms * 1e3, flop_per_block_iter * blocks * iters / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 4 * 256 * 8192);
    const int iters = 64;
    const double f16 = 2.0 * 16 * 16 * 4 * 64 / 64;  // flop per wave-MFMA 16x16x4 = 2048
    for (int bpc : {1, 2, 3, 4}) {
        int blocks = 256 * bpc * 4;
        char nm[128];
        snprintf(nm, sizeof nm, "V0 regs-only 16 acc, %d WG/CU-rounds", bpc);
        run(nm, [&] { hipLaunchKernelGGL(v0, dim3(blocks), dim3(256), 0, 0, out, iters); }, 4.0 * 18 * 16 * 2048, blocks, iters);
    }
    size_t lds1 = 4 * (8 * 432 + 8 * 9 * 64);
    hipFuncSetAttribute((const void*)v1<4, 4, 1, 432>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)v1<4, 4, 0, 432>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)v1<4, 4, 1, 400>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)v2<1, 432>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (size_t pad : {(size_t)0, (size_t)30 * 1024, (size_t)50 * 1024, (size_t)100 * 1024}) {  // pad LDS to force 4/3/2/1 WG per CU
        int blocks = 256 * 8;
        char nm[128];
        snprintf(nm, sizeof nm, "V1 16x16x4 MW4 NB4 barrier, lds=%zuK", (lds1 + pad) / 1024);
        run(nm, [&] { hipLaunchKernelGGL((v1<4, 4, 1, 432>), dim3(blocks), dim3(256), lds1 + pad, 0, out, iters); }, 4.0 * 18 * 16 * 2048, blocks, iters);
        snprintf(nm, sizeof nm, "V1 16x16x4 MW4 NB4 no-barrier, lds=%zuK", (lds1 + pad) / 1024);
        run(nm, [&] { hipLaunchKernelGGL((v1<4, 4, 0, 432>), dim3(blocks), dim3(256), lds1 + pad, 0, out, iters); }, 4.0 * 18 * 16 * 2048, blocks, iters);
        snprintf(nm, sizeof nm, "V1 plane=400 (2-way A conflicts), lds=%zuK", (lds1 + pad) / 1024);
        run(nm, [&] { hipLaunchKernelGGL((v1<4, 4, 1, 400>), dim3(blocks), dim3(256), lds1 + pad, 0, out, iters); }, 4.0 * 18 * 16 * 2048, blocks, iters);
        snprintf(nm, sizeof nm, "V2 32x32x2 2x2 barrier, lds=%zuK", (lds1 + pad) / 1024);
        run(nm, [&] { hipLaunchKernelGGL((v2<1, 432>), dim3(blocks), dim3(256), lds1 + pad, 0, out, iters); }, 4.0 * 18 * 16 * 2048, blocks, iters);
    }
    return 0;
}
