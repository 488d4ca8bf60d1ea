// kb.hip -- calibrated backprojection (KB) layer pieces for gfx950.
//
//  * kbn_intrinsics_inverse : K (.) scale, then K^-1           reference src/networks.py:328, 333-352
//  * kbn_camera_coordinates : K^-1 [x y 1]^T per pixel         reference src/networks.py:317-331
//  * kbn_kb_block_forward   : CalibratedBackprojectionBlock    reference src/net_utils.py:1343-1371
//      conv_image  3x3 s2                          -> conv_igemm<3,2>
//      conv_depth  3x3 s2 on cat[depth, coords]    -> conv_igemm<3,2>, coords staged in-kernel
//      conv_fused  1x1 s2 on cat[image, coords * act(proj . depth), fused]
//                                                  -> conv_igemm<1,2>, the backprojection
//                                                     (proj_depth dot product, xyz = coords*z)
//                                                     happens while the tile is staged, and only
//                                                     at the even pixels the stride-2 conv reads.
#include "conv_common.h"

namespace kbn {

__global__ void intrinsics_inverse_kernel(const float* __restrict__ k, float* __restrict__ kinv, int n,
                                          float sx, float sy) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = k + (long long)i * 9;
    // element-wise scale in fp32, as `k * scale` does (reference src/networks.py:346-352)
    float m[9] = {a[0] * sx, a[1] * 1.0f, a[2] * sx, a[3] * 1.0f, a[4] * sy, a[5] * sy,
                  a[6] * 1.0f, a[7] * 1.0f, a[8] * 1.0f};
    double A = m[0], B = m[1], C = m[2], D = m[3], E = m[4], F = m[5], G = m[6], H = m[7], I = m[8];
    double c00 = E * I - F * H, c01 = -(D * I - F * G), c02 = D * H - E * G;
    double det = A * c00 + B * c01 + C * c02;
    double r = 1.0 / det;
    float* o = kinv + (long long)i * 9;
    o[0] = (float)(c00 * r);
    o[1] = (float)(-(B * I - C * H) * r);
    o[2] = (float)((B * F - C * E) * r);
    o[3] = (float)(c01 * r);
    o[4] = (float)((A * I - C * G) * r);
    o[5] = (float)(-(A * F - C * D) * r);
    o[6] = (float)(c02 * r);
    o[7] = (float)(-(A * H - B * G) * r);
    o[8] = (float)((A * E - B * D) * r);
}

__global__ void camera_coordinates_kernel(const float* __restrict__ kinv, float* __restrict__ out, int n,
                                          int H, int W) {
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long HW = (long long)H * W;
    if (e >= (long long)n * HW) return;
    int b = (int)(e / HW);
    int pix = (int)(e - (long long)b * HW);
    int y = pix / W, x = pix - y * W;
    const float* k = kinv + (long long)b * 9;
    float* o = out + (long long)b * 3 * HW + pix;
#pragma unroll
    for (int j = 0; j < 3; ++j) o[(long long)j * HW] = fmaf(k[j * 3 + 1], (float)y, k[j * 3 + 0] * (float)x) + k[j * 3 + 2];
}

}  // namespace kbn

extern "C" {

int kbn_intrinsics_inverse(const float* intrinsics, float* kinv, int n, float scale_x, float scale_y,
                           kbn_stream_t stream) {
    if (!intrinsics || !kinv || n < 1) return KBN_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(kbn::intrinsics_inverse_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       intrinsics, kinv, n, scale_x, scale_y);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_camera_coordinates(const float* kinv, float* coordinates, int n, int height, int width,
                           kbn_stream_t stream) {
    if (!kinv || !coordinates || n < 1 || height < 1 || width < 1) return KBN_ERR_INVALID_ARGUMENT;
    long long total = (long long)n * height * width;
    hipLaunchKernelGGL(kbn::camera_coordinates_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, kinv, coordinates, n, height, width);
    KBN_CHECK_LAUNCH();
    return KBN_OK;
}

int kbn_kb_block_forward(const float* image, long long image_batch_stride, const float* depth,
                         long long depth_batch_stride, const float* coordinates, const float* kinv,
                         const float* fused, long long fused_batch_stride, const float* packed_w_image,
                         const float* packed_w_depth,
                         const float* proj_weight, const float* packed_w_fused, float* out_image,
                         long long out_image_batch_stride, float* out_depth, long long out_depth_batch_stride,
                         float* out_fused, long long out_fused_batch_stride, int n, int height, int width,
                         int channels_image, int channels_depth, int channels_fused, int filters_image,
                         int filters_depth, int filters_fused, float negative_slope, unsigned* out_image_absmax,
                         unsigned* out_depth_absmax, unsigned* out_fused_absmax, kbn_stream_t stream) {
    if (!image || !depth || (!coordinates && !kinv) || !packed_w_image || !packed_w_depth || !proj_weight ||
        !packed_w_fused || !out_image || !out_depth || !out_fused)
        return KBN_ERR_INVALID_ARGUMENT;
    if ((fused == nullptr) != (channels_fused == 0)) return KBN_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    const long long HW = (long long)height * width;
    int rc;

    kbn_conv_src s_img{};
    s_img.kind = KBN_SRC_TENSOR; s_img.channels = channels_image; s_img.data = image;
    s_img.batch_stride = image_batch_stride; s_img.src_height = height; s_img.src_width = width;

    // Fast path: conv_image and conv_fused in one launch (kb_pair.hip) -- they read the same image tile -- with
    // conv_depth's MFMAs riding along in the same workgroups when its filter count fits.
    bool paired = false, depth_done = false;
    if (filters_image == filters_fused) {
        kbn::KbPairArgs a{};
        a.image = image; a.fused = fused; a.depth = depth; a.coords = coordinates; a.kinv = kinv; a.proj = proj_weight;
        a.wp_image = packed_w_image; a.wp_fused = packed_w_fused; a.out_image = out_image; a.out_fused = out_fused;
        a.image_bstride = image_batch_stride; a.fused_bstride = fused_batch_stride; a.depth_bstride = depth_batch_stride;
        a.coords_bstride = 3 * HW; a.out_image_bstride = out_image_batch_stride; a.out_fused_bstride = out_fused_batch_stride;
        a.n = n; a.height = height; a.width = width; a.channels_image = channels_image; a.channels_depth = channels_depth;
        a.channels_fused = channels_fused; a.filters = filters_image; a.slope = negative_slope;
        a.wp_depth = packed_w_depth; a.out_depth = out_depth; a.out_depth_bstride = out_depth_batch_stride;
        a.filters_depth = filters_depth;
        a.absmax_image = out_image_absmax; a.absmax_fused = out_fused_absmax; a.absmax_depth = out_depth_absmax;
        rc = kbn::kb_pair_launch(a, st, &depth_done);
        if (rc == KBN_OK) paired = true;
        else if (rc != KBN_ERR_UNSUPPORTED) return rc;
    }

    // conv_image = act(conv3x3 s2 (image))                      src/net_utils.py:1348
    if (!paired) {
        rc = kbn::conv2d_launch(&s_img, 1, packed_w_image, out_image, out_image_batch_stride, n, filters_image, 3, 2,
                                height, width, KBN_RESIZE_NONE, 1, negative_slope, out_image_absmax, st);
        if (rc != KBN_OK) return rc;
    }

    // conv_depth = act(conv3x3 s2 (cat[depth, coordinates]))    src/net_utils.py:1351
    kbn_conv_src s_dep[2] = {};
    s_dep[0].kind = KBN_SRC_TENSOR; s_dep[0].channels = channels_depth; s_dep[0].data = depth;
    s_dep[0].batch_stride = depth_batch_stride; s_dep[0].src_height = height; s_dep[0].src_width = width;
    if (coordinates) {
        s_dep[1].kind = KBN_SRC_TENSOR; s_dep[1].channels = 3; s_dep[1].data = coordinates;
        s_dep[1].batch_stride = 3 * HW; s_dep[1].src_height = height; s_dep[1].src_width = width;
    } else {
        s_dep[1].kind = KBN_SRC_COORDS; s_dep[1].channels = 3; s_dep[1].kinv = kinv;
    }
    if (!depth_done) {
        rc = kbn::conv2d_launch(s_dep, 2, packed_w_depth, out_depth, out_depth_batch_stride, n, filters_depth, 3, 2,
                                height, width, KBN_RESIZE_NONE, 1, negative_slope, out_depth_absmax, st);
        if (rc != KBN_OK) return rc;
    }
    if (paired) return KBN_OK;

    // conv_fused = act(conv1x1 s2 (cat[image, coordinates * act(proj_depth(depth)), fused]))
    //                                                            src/net_utils.py:1354-1369
    kbn_conv_src s_fus[3] = {};
    s_fus[0] = s_img;
    s_fus[1].kind = KBN_SRC_XYZ; s_fus[1].channels = 3; s_fus[1].data = depth;
    s_fus[1].batch_stride = depth_batch_stride; s_fus[1].aux_channels = channels_depth;
    s_fus[1].proj_weight = proj_weight; s_fus[1].coordinates = coordinates;
    s_fus[1].coordinates_batch_stride = 3 * HW; s_fus[1].kinv = kinv;
    int nsrc = 2;
    if (fused) {
        s_fus[2].kind = KBN_SRC_TENSOR; s_fus[2].channels = channels_fused; s_fus[2].data = fused;
        s_fus[2].batch_stride = fused_batch_stride; s_fus[2].src_height = height; s_fus[2].src_width = width;
        nsrc = 3;
    }
    return kbn::conv2d_launch(s_fus, nsrc, packed_w_fused, out_fused, out_fused_batch_stride, n, filters_fused, 1,
                              2, height, width, KBN_RESIZE_NONE, 1, negative_slope, out_fused_absmax, st);
}

}  // extern "C"
