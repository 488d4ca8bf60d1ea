// VALU issue rate vs waves per SIMD (one workgroup per CU): v_fma_f32 with a VGPR or an SGPR multiplicand,
// v_pk_fma_f32, v_min3_f32; 8 independent chains per wave.  Prints clk per wave-instruction and the SIMD's
// aggregate rate.  Build: hipcc --offload-arch=gfx950 -O3 valu_probe.hip -o valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float sb, unsigned long long* cyc) {
    float v[8]; f32x2 p[8];
    for (int i = 0; i < 8; ++i) { v[i] = i + threadIdx.x; p[i] = (f32x2){v[i], v[i] + 1}; }
    const float b = 1.0001f, c = 0.5f;
    const f32x2 b2 = {b, b}, c2 = {c, c};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(b), "v"(c));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 7]) : "s"(sb), "v"(c));
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "v"(b2), "v"(c2));
            if (KIND == 3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(b), "v"(c));
            if (KIND == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i & 7]) : "s"(sb), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += v[i] + p[i].x + p[i].y;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND>
static void run(const char* what) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 4 * 1024 * 256); (void)hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int nw = 1; nw <= 4; ++nw) {
        hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
        auto launch = [&] { hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256 * nw), 0, 0, out, iters, 1.0001f, cyc); };
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(s); launch(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
        float ms; (void)hipEventElapsedTime(&ms, s, e);
        unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * 64;
        printf("%-28s waves/SIMD=%d  %7.2f clk/instr/wave  -> SIMD issues one every %5.2f clk  (wall %6.1f us)\n", what, nw, h / n, h / n / nw, ms * 1e3);
    }
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0>("v_fma_f32 vgpr operands");
    run<1>("v_fma_f32 sgpr multiplicand");
    run<4>("v_fmac_f32 sgpr multiplicand");
    run<2>("v_pk_fma_f32");
    run<3>("v_min3_f32");
    return 0;
}
