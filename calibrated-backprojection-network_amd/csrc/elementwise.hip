// elementwise.hip -- the two byte-bound passes of the LAYER-BY-LAYER form of the network, which runs when the model's activation is
// not a (leaky) ReLU (run_kbnet.py --activation_func elu | sigmoid | linear; reference src/net_utils.py:23-45) and the fused kernels
// -- all written around max(v, slope v) -- step aside:
//   kbn_activation_forward   x <- act(x) in place after a conv launched without activation: ELU (alpha 1, torch.nn.ELU()) or sigmoid; folds
//                            max |act(x)| per frame into the tensor's slot, so that split-operand convs downstream place their windows on it
//   kbn_scale_planes_forward xyz = coordinates * z, the KB block's backprojection (src/net_utils.py:1352-1359) once z = act(proj_depth . depth)
//                            is a tensor of its own instead of a value inside the conv's staging
// Both are one read and one write per element: 16-byte accesses where the planes allow, a grid-stride loop over (frame, element).
#include <math.h>

#include "kbn_common.h"

namespace kbn {

template <int KIND>   // 1 ELU, 2 sigmoid
__device__ __forceinline__ float act_apply(float v) {
    if constexpr (KIND == 1) return v > 0.f ? v : expm1f(v);
    else return 1.0f / (1.0f + expf(-v));
}

template <int KIND, bool VEC>
__global__ __launch_bounds__(256) void activation_kernel(float* __restrict__ x, long long batch_stride, long long per_frame,
                                                         unsigned* __restrict__ out_absmax) {
    float* xn = x + (long long)blockIdx.y * batch_stride;
    const long long step = (long long)gridDim.x * 256;
    float m = 0.f;   // max |act(v)| of this thread's elements: the frame's slot takes the activated tensor's maximum (kbn_common.h)
    if constexpr (VEC) {
        const long long quads = per_frame >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < quads; i += step) {
            float4 v = reinterpret_cast<float4*>(xn)[i];
            v.x = act_apply<KIND>(v.x); v.y = act_apply<KIND>(v.y); v.z = act_apply<KIND>(v.z); v.w = act_apply<KIND>(v.w);
            reinterpret_cast<float4*>(xn)[i] = v;
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (long long i = (quads << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < per_frame; i += step) {
            const float v = act_apply<KIND>(xn[i]);
            xn[i] = v;
            m = fmaxf(m, fabsf(v));
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_frame; i += step) {
            const float v = act_apply<KIND>(xn[i]);
            xn[i] = v;
            m = fmaxf(m, fabsf(v));
        }
    }
    if (out_absmax) absmax_commit(out_absmax + blockIdx.y, m);
}

__global__ __launch_bounds__(256) void scale_planes_kernel(const float* __restrict__ x, long long x_bs, const float* __restrict__ z, long long z_bs,
                                                           float* __restrict__ out, long long out_bs, int channels, long long plane) {
    const int n = blockIdx.y;
    const float* zn = z + (long long)n * z_bs;
    const float* xn = x + (long long)n * x_bs;
    float* on = out + (long long)n * out_bs;
    const long long step = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < plane; i += step) {
        const float s = zn[i];
        for (int c = 0; c < channels; ++c) on[c * plane + i] = xn[c * plane + i] * s;
    }
}

}  // namespace kbn

extern "C" {

int kbn_activation_forward(float* x, long long batch_stride, int n, long long per_frame, int kind, unsigned* out_absmax,
                           kbn_stream_t stream) {
    using namespace kbn;
    if (!x || n < 1 || per_frame < 1 || (kind != KBN_ACT_ELU && kind != KBN_ACT_SIGMOID)) return KBN_ERR_INVALID_ARGUMENT;
    if (n > 65535) return KBN_ERR_UNSUPPORTED;
    const bool vec = !((reinterpret_cast<uintptr_t>(x) & 15) || (batch_stride & 3));
    const int blocks = (int)std::min<long long>(std::max(1, 4096 / n), (per_frame + 1023) / 1024);
    const dim3 grid(blocks, n);
    hipStream_t st = (hipStream_t)stream;
    if (kind == KBN_ACT_ELU) {
        if (vec) hipLaunchKernelGGL((activation_kernel<1, true>), grid, dim3(256), 0, st, x, batch_stride, per_frame, out_absmax);
        else hipLaunchKernelGGL((activation_kernel<1, false>), grid, dim3(256), 0, st, x, batch_stride, per_frame, out_absmax);
    } else {
        if (vec) hipLaunchKernelGGL((activation_kernel<2, true>), grid, dim3(256), 0, st, x, batch_stride, per_frame, out_absmax);
        else hipLaunchKernelGGL((activation_kernel<2, false>), grid, dim3(256), 0, st, x, batch_stride, per_frame, out_absmax);
    }
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_scale_planes_forward(const float* x, long long x_batch_stride, const float* z, long long z_batch_stride, float* out,
                             long long out_batch_stride, int n, int channels, int height, int width, kbn_stream_t stream) {
    using namespace kbn;
    if (!x || !z || !out || n < 1 || channels < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    if (n > 65535) return KBN_ERR_UNSUPPORTED;
    const long long plane = (long long)height * width;
    const int blocks = (int)std::min<long long>(std::max(1, 4096 / n), (plane + 255) / 256);
    hipLaunchKernelGGL(scale_planes_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, x, x_batch_stride, z, z_batch_stride, out,
                       out_batch_stride, channels, plane);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

}  // extern "C"
