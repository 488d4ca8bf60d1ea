// mfma_power_probe.hip -- sustained rate of a pure fp16 MFMA stream on the whole chip (256 CUs x 8 waves), long enough for
// the power management to settle: v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16, random against zero operands.
// The split-operand convs (csrc/conv_split.hip) sit at the chip's power limit; this measures what that limit is for a
// stream that does nothing else, and whether the 16x16x32 shape (half the accumulator traffic per FLOP, twice the A/B
// operand traffic) draws less per FLOP.   build: hipcc --offload-arch=gfx950 -O3 mfma_power_probe.hip -o mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(512, 1) void stream(const _Float16* src, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a[i][k] = src[((i * 64 + lane) * 8 + k) & 4095];
            b[i][k] = src[((i * 64 + lane) * 8 + k + 2048) & 4095];
        }
    if constexpr (SHAPE == 0) {
        f16v acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[i], acc[i], 0, 0, 0);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
        if (s == 12345.678f) out[0] = s;
    } else {
        f4v acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f4v){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) & 3], b[i & 3], acc[i], 0, 0, 0);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.678f) out[0] = s;
    }
}

// The same streams with the operand traffic a conv kernel has: every MFMA operand comes out of LDS (ds_read_b128), each
// fragment used by four MFMAs -- 32x32x16 on a 2 x 2 block tile: 4 reads per 4 MFMAs (128 clk); 16x16x32 on a 4 x 4 block
// tile: 8 reads per 16 MFMAs (256 clk).  MIX 2: one 32x32x16 (main term) + four 16x16x32 (the two small terms fused along
// K) per 32 x 32 block, the hybrid form.
template <int MIX>
__global__ __launch_bounds__(512, 1) void stream_lds(const _Float16* src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[32 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32 * 1024; i += 512) lds[i] = src[i];   // 32 Ki independent values: no two fragments alike
    __syncthreads();
    const h8* base = reinterpret_cast<const h8*>(lds) + lane + (threadIdx.x >> 6) * 64;
    float s = 0.f;
    if constexpr (MIX == 0) {
        f16v acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ((it * 4 + r) & 7) * 512;
                h8 a0 = base[o], a1 = base[o + 128 * 4], b0 = base[o + 256 * 4], b1 = base[o + 384 * 4];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    } else if constexpr (MIX == 1) {
        f4v acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (f4v){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            const int o = (it & 7) * 512;
            h8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = base[o + i * 512]; b[i] = base[o + 2048 + i * 512]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f16v accm[4];
        f4v accs[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) accm[i][k] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) accs[i] = (f4v){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            const int o = (it & 7) * 512;
            h8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = base[o + i * 512]; b[i] = base[o + 2048 + i * 512]; }
            h8 a0 = base[o + 64], a1 = base[o + 576], b0 = base[o + 1088], b1 = base[o + 1600];
            accm[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, accm[0], 0, 0, 0);
            accm[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, accm[1], 0, 0, 0);
            accm[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, accm[2], 0, 0, 0);
            accm[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, accm[3], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) accs[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], accs[i * 4 + j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s += accm[i][0] + accm[i][7];
#pragma unroll
        for (int i = 0; i < 16; ++i) s += accs[i][0] + accs[i][3];
    }
    if (s == 12345.678f) out[0] = s;
}

int main() {
    _Float16* src; float* out;
    hipMalloc(&src, 32768 * sizeof(_Float16)); hipMalloc(&out, 64);
    std::vector<_Float16> h(32768);
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero) {
        srand(1);
        for (auto& v : h) v = zero ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 8.f);
        hipMemcpy(src, h.data(), 32768 * sizeof(_Float16), hipMemcpyHostToDevice);
        for (int shape = 0; shape < 2; ++shape) {
            const int iters = 200000;                      // x 16 (32 for 16x16x32) MFMAs per wave
            const double flop_per_wave = shape == 0 ? iters * 16.0 * 32768.0 : iters * 32.0 * 16384.0;
            auto launch = [&](int it) {
                if (shape == 0) hipLaunchKernelGGL(stream<0>, dim3(cus), dim3(512), 0, 0, src, out, it);
                else hipLaunchKernelGGL(stream<1>, dim3(cus), dim3(512), 0, 0, src, out, it);
            };
            launch(iters / 10); hipDeviceSynchronize();
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
                const double tf = flop_per_wave * 8.0 * cus / (ms * 1e-3) / 1e12;
                printf("%s operands, %-24s: %8.2f ms  %7.1f TFLOP/s  (%.2f GHz-equivalent of the 2.5 PFLOP/s @ 2.4 GHz pipe)\n",
                       zero ? "zero  " : "random", shape == 0 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", ms, tf, tf / 2500.0 * 2.4);
            }
        }
    }
    // operands through LDS (random data)
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 8.f);
    hipMemcpy(src, h.data(), 32768 * sizeof(_Float16), hipMemcpyHostToDevice);
    for (int mix = 0; mix < 3; ++mix) {
        const int iters = mix == 0 ? 300000 : mix == 1 ? 600000 : 400000;   // >= 150 ms each: the power management needs tens of ms to settle
        // FLOPs per wave and iteration: MIX 0: 16 x 32768; MIX 1: 16 x 16384; MIX 2: 4 x 32768 + 16 x 16384
        const double flop_per_wave = iters * (mix == 0 ? 16 * 32768.0 : mix == 1 ? 16 * 16384.0 : 4 * 32768.0 + 16 * 16384.0);
        auto launch = [&](int it) {
            if (mix == 0) hipLaunchKernelGGL(stream_lds<0>, dim3(cus), dim3(512), 0, 0, src, out, it);
            else if (mix == 1) hipLaunchKernelGGL(stream_lds<1>, dim3(cus), dim3(512), 0, 0, src, out, it);
            else hipLaunchKernelGGL(stream_lds<2>, dim3(cus), dim3(512), 0, 0, src, out, it);
        };
        launch(iters / 10); hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            printf("operands from LDS, %-58s: %8.2f ms  %7.1f TFLOP/s\n",
                   mix == 0 ? "32x32x16, 4 ds_read_b128 per 4 MFMAs" : mix == 1 ? "16x16x32, 8 ds_read_b128 per 16 MFMAs"
                                                                     : "hybrid: 4 x 32x32x16 + 16 x 16x16x32, 12 reads",
                   ms, flop_per_wave * 8.0 * cus / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
