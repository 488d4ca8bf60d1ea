/*
 * kbnet_hip.h -- C ABI of the MI355X (gfx950) KBNet inference hot path.
 *
 * The reference (alexklwong/calibrated-backprojection-network) is pure Python on
 * PyTorch and has no FFI of its own; the entry points below are what a binding for
 * its hot path would call, one per reference operator.  Each declaration cites the
 * reference interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - every tensor is fp32, NCHW, W-contiguous, device (HBM) memory owned by the
 *     caller; the library allocates nothing and keeps no pointer after return;
 *   - every call only enqueues work on `stream` (a hipStream_t of the calling thread's
 *     current device) and returns: no synchronisation, no timing, no file I/O -- unless
 *     the caller has switched first-use tuning on (kbn_set_autotune, OFF by default);
 *   - return value: KBN_OK (0) or a negative kbn_status; nothing throws across the ABI;
 *   - the library is re-entrant and multi-device: its only process state is (a) one-time
 *     kernel attribute setup, done per device, (b) the debug knobs, read from the
 *     environment once at load time (kbn_reload_env re-reads them), (c) the launch-geometry
 *     cache, per device, written only while tuning is on.
 */
#ifndef KBNET_HIP_H
#define KBNET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KBN_ABI_VERSION 7

typedef void* kbn_stream_t; /* hipStream_t */

typedef enum kbn_status {
    KBN_OK = 0,
    KBN_ERR_INVALID_ARGUMENT = -1, /* null pointer, non-positive size, bad enum      */
    KBN_ERR_UNSUPPORTED = -2,      /* legal for the reference, outside kernel limits */
    KBN_ERR_WORKSPACE = -3,        /* caller-provided buffer too small               */
    KBN_ERR_LAUNCH = -4            /* HIP reported a launch error                    */
} kbn_status;

int kbn_version(void);
const char* kbn_status_string(int status);

/* Re-reads the KBN_* debug / experiment variables and KBN_TUNE_CACHE from the environment
 * (they are otherwise read once, when the library is loaded; no launch path calls getenv). */
void kbn_reload_env(void);

/* Value of a KBN_* switch AS THE LIBRARY READ IT (at load time / the last kbn_reload_env()): 0 when unset or unknown.  The
 * host mirror asks here instead of reading the environment itself, so its A/B switches (KBN_NO_SPLIT, KBN_NO_PAIR,
 * KBN_NO_PAIR_MID, KBN_NO_PAIR_ENC, KBN_NO_PAIR_TAIL, KBN_NO_OVERLAP, KBN_DEPTH_FRONT_FUSION, KBN_NO_DEPTH_FRONT_FUSION,
 * KBN_FP16_ONE_TERM, KBN_NO_FRONT_NEXT) change together with the library's.  Values are integers (atoi); a variable that is set to
 * something that is not a number ("true", "yes", "on") reads as 1, so that KBN_NO_PAIR=true still switches the path off. */
int kbn_knob(const char* name);

/* First-use tuning of launch geometry (tile / region shapes; csrc/tune.hip).  OFF by default:
 * calls launch the analytic choice or a cached one.  While ON, the first call of a problem
 * shape on a device times every candidate geometry on `stream` and waits for its own events --
 * results are bit-identical whatever is picked -- so switch it on only around a warm-up pass:
 *     kbn_set_autotune(1); <one forward per shape>; kbn_set_autotune(0);
 * Never tunes while `stream` is being captured.  KBN_AUTOTUNE=1 in the environment at load time
 * is the same switch; KBN_TUNE_CACHE=<file> preloads / records choices.  Returns the old state. */
int kbn_set_autotune(int enabled);
int kbn_get_autotune(void);

/* ------------------------------------------------------------------ S2D -------
 * networks.SparseToDensePool.forward(x)            reference src/networks.py:2168-2196
 * (pool construction :2112-2132, 1x1 convs :2138-2152, 3x3 conv :2156-2166).
 *
 *   x              N x input_channels x H x W; channel 0 is the sparse depth
 *   w_pool_convs   n_convolution pointers: [0] is n_filter x (n_min+n_max) x 1 x 1,
 *                  the others n_filter x n_filter x 1 x 1  (pool_convs.{i}.conv.weight)
 *   w_conv         n_filter x (n_filter+input_channels) x 3 x 3   (conv.conv.weight)
 *   out            N x n_filter x H x W
 *   pool sizes     odd kernel sizes > 1 (the caller drops sizes <= 1 like the
 *                  reference does); min pools ignore zeros (999 sentinel semantics)
 * Limits: n_filter <= 8, input_channels <= 2, n_min+n_max <= 8, n_convolution <= 4,
 *         pool size <= 31 and odd.
 */
int kbn_s2d_forward(const float* x, const float* const* w_pool_convs, const float* w_conv,
                    float* out, int n, int height, int width, int input_channels,
                    const int* min_pool_sizes, int n_min, const int* max_pool_sizes, int n_max,
                    int n_convolution, int n_filter, float negative_slope, kbn_stream_t stream);

/* Optional debug/parity output of the min/max pyramid alone (the tensor the
 * reference concatenates at src/networks.py:2189): N x (n_min+n_max) x H x W. */
int kbn_s2d_pyramid(const float* x_depth, long long batch_stride, float* pyramid, int n,
                    int height, int width, const int* min_pool_sizes, int n_min,
                    const int* max_pool_sizes, int n_max, kbn_stream_t stream);

/* ----------------------------------------------------------- intrinsics --------
 * KBNetEncoder.forward closures scale_intrinsics + torch.inverse
 *                                                  reference src/networks.py:328, 333-352
 * kinv[n] = inverse(K[n] (.) [[sx,1,sx],[1,sy,sy],[1,1,1]]); pass sx = sy = 1 for level 0.
 */
int kbn_intrinsics_inverse(const float* intrinsics, float* kinv, int n, float scale_x,
                           float scale_y, kbn_stream_t stream);

/* camera_coordinates closure + net_utils.meshgrid   reference src/networks.py:317-331,
 * src/net_utils.py:1601-1636.  coordinates: N x 3 x H x W = kinv . [x y 1]^T. */
int kbn_camera_coordinates(const float* kinv, float* coordinates, int n, int height, int width,
                           kbn_stream_t stream);

/* ------------------------------------------------------------- conv2d ----------
 * net_utils.Conv2d.forward (bias-free conv, padding k//2, optional LeakyReLU)
 *                                                  reference src/net_utils.py:85-93, 120-141
 * fused with the ops the reference runs around it:
 *   - torch.cat of the inputs along channels (src/net_utils.py:1351, 1366, 1483):
 *     up to KBN_MAX_SRC sources, concatenated in order;
 *   - UpConv2d's nearest interpolate (src/net_utils.py:497): KBN_RESIZE_NEAREST
 *     (only with a single tensor source);
 *   - the KB layer's coordinate / backprojection channels (src/net_utils.py:1351-1360),
 *     synthesized while the input tile is staged (source kinds below).
 */
#define KBN_MAX_SRC 3

typedef enum kbn_src_kind {
    KBN_SRC_TENSOR = 0, /* data: N x channels x src_height x src_width                     */
    KBN_SRC_COORDS = 1, /* 3 channels kinv.[x y 1]^T, kinv: N x 3 x 3 (data unused)        */
    KBN_SRC_XYZ = 2,    /* 3 channels coords * z, z = act(proj_weight . depth[:, y, x]);   */
                        /* data = depth N x aux_channels x H x W, proj_weight = aux_channels */
                        /* floats, coords from `coordinates` (N x 3 x H x W) if non-null,  */
                        /* else from kinv                                                  */
    KBN_SRC_PAIR = 3    /* a PAIR tensor (below): activations already split into two fp16 terms by  */
                        /* their producer; kbn_conv3x3_split_forward only, source 0 only            */
} kbn_src_kind;

typedef struct kbn_conv_src {
    int kind;               /* kbn_src_kind                                       */
    int channels;           /* channels this source contributes to the concat     */
    const float* data;
    long long batch_stride; /* elements between consecutive frames of `data`      */
    int src_height, src_width; /* size of `data` planes (KBN_SRC_TENSOR)          */
    int aux_channels;       /* KBN_SRC_XYZ: channels of the depth feature tensor  */
    const float* proj_weight;  /* KBN_SRC_XYZ                                     */
    const float* coordinates;  /* KBN_SRC_XYZ, optional                           */
    long long coordinates_batch_stride;
    const float* kinv;      /* KBN_SRC_COORDS / KBN_SRC_XYZ without coordinates    */
    const unsigned* absmax; /* KBN_SRC_TENSOR / KBN_SRC_PAIR, optional: the per-frame max |a| slots of this tensor
                             * (see "activation statistics" below); read by the split-operand convs only  */
    const float* scale;     /* KBN_SRC_PAIR: the per-frame 2^k its producer wrote (pair_out_scale)         */
} kbn_conv_src;

/* ------------------------------------------------ PAIR tensors (producer-written split operands) --
 * The reference passes fp32 tensors between its convs (src/net_utils.py:1483-1487: conv over cat[deconv, skip]).  Where a
 * tensor is read ONLY by split-operand convs -- the decoder's up-conv outputs and concat-conv outputs -- the producer
 * can write the two fp16 terms itself, once, instead of every consumer splitting every value it stages (2.4-4.8 times
 * per value: halos x filter tiles).  Layout per frame, C channels (C % 8 == 0), H x W pixels:
 *     [k-group = channel / 8][term: h1 | h2][H * W + 1 pixels][8 channels] fp16      (4 bytes per value, like fp32)
 * a 2^k = h1 + 2^-11 h2; the extra pixel that ends every plane is ZERO (written by the producer; a consumer's halo pixels
 * outside the map read it).  Frames are batch_stride fp16 elements apart (>= (C / 8) * 2 * (H * W + 1) * 8, a multiple
 * of 8; base 16-byte aligned).  k is chosen per frame by the producer from a bound of its output -- (max |a| of each
 * input, from the inputs' absmax slots) x (a table of weight norms kept behind the packed weights) -- and written to
 * `pair_out_scale[frame]` as the float 2^k; the consumer receives that array as kbn_conv_src.scale.  The producer still
 * folds the true max |out| into out_absmax.  kbn_conv3x3_split_forward: `pair_out` for modes 0 and 2 and for mode 3 with whole
 * 64-filter tiles (needs absmax slots on every source and out_channels % 8 == 0; mode 3 with at most 16 filters writes 16 channels, zeros past out_channels); a KBN_SRC_PAIR source 0 for mode 0
 * (beside an fp32 source 1), mode 2 (one source) and mode 3 (64-filter tiles, or at most 16 filters and channels % 32 == 0); KBN_ERR_UNSUPPORTED otherwise -- the
 * caller then keeps the tensor in fp32. */

/* ----------------------------------------------- activation statistics ("absmax slots") --
 * The split-operand convs further down represent an activation as two fp16 terms under a per-frame
 * exponent, which has to follow the magnitude of the data.  It does so on the device: a SLOT is an array of
 * n unsigned values, one per frame, holding the bit pattern of max |a| over that frame of a tensor
 * (bit patterns of non-negative floats order like the floats).  Every conv entry point takes `out_absmax`
 * (may be NULL): the kernel folds the values it stores into out_absmax[frame] with an atomic max in its epilogue
 * (several launches may fill channel slices of one tensor into one slot); the caller zeroes a slot before its first
 * producer.  A consumer is handed the slot through kbn_conv_src.absmax.  kbn_absmax_frames is the stand-alone
 * pass for tensors that come from elsewhere.  No host synchronisation, no state between calls: the result of a
 * forward is a function of its inputs and weights alone, frame by frame (the reference's convs are exactly
 * that: src/net_utils.py:120-141). */
int kbn_absmax_frames(const float* x, long long batch_stride, int n, long long per_frame, unsigned* slots,
                      kbn_stream_t stream);

#define KBN_RESIZE_NONE 0
#define KBN_RESIZE_NEAREST 1

/* Bytes of the packed weight blob for a conv with these dimensions (0 if unsupported).  The
 * blob layout depends on the stride the weight will be used with. */
size_t kbn_conv2d_packed_weight_bytes(int out_channels, int in_channels, int kernel_size, int stride);

/* Re-orders an OIHW weight (out_channels x in_channels x k x k) into the MFMA
 * fragment order the kernel consumes.  `packed` must hold
 * kbn_conv2d_packed_weight_bytes(...) bytes.  Do this once per weight (and stride).  For wide
 * 3x3 stride-1 convs the blob also carries the Winograd-domain weights G g G^T behind the
 * fragment-order copy; kbn_conv2d_forward picks the kernel per launch. */
int kbn_conv2d_pack_weight(const float* weight, float* packed, int out_channels, int in_channels,
                           int kernel_size, int stride, kbn_stream_t stream);

/* out[n, :, oy, ox] = act(sum_c,ky,kx W[:, c, ky, kx] * in[n, c, oy*stride+ky-pad, ox*stride+kx-pad])
 * where `in` is the channel concat of the sources, logically in_height x in_width
 * (sources are nearest-resized to that size first when resize = KBN_RESIZE_NEAREST).
 * out: frames `out_batch_stride` elements apart (lets the caller write straight into a
 * channel slice of a larger skip tensor), out_channels x ceil(in_h/stride) x ceil(in_w/stride).
 * kernel_size in {1, 3}; stride in {1, 2}.  out_absmax: n slots of max |out| per frame, or NULL. */
int kbn_conv2d_forward(const kbn_conv_src* srcs, int n_src, const float* packed_weight, float* out,
                       long long out_batch_stride, int n, int out_channels, int kernel_size,
                       int stride, int in_height, int in_width, int resize, int apply_activation,
                       float negative_slope, unsigned* out_absmax, kbn_stream_t stream);

/* --------------------------------------------- layer-by-layer form: activation, backprojection -------
 * The fused kernels are written around max(v, slope v): LeakyReLU(0.20), ReLU (slope 0) and -- with slope 1 -- no activation.
 * net_utils.activation_func also builds torch.nn.ELU() and torch.nn.Sigmoid() (reference src/net_utils.py:38-43; run_kbnet.py
 * --activation_func elu | sigmoid).  A model with one of those runs layer by layer: every conv is launched WITHOUT activation
 * (apply_activation = 0) and followed by kbn_activation_forward in place; the KB block's z = act(proj_depth . depth)
 * (:1352-1355) is then a tensor of its own and xyz = coordinates * z (:1357-1359) one kbn_scale_planes_forward.
 *   kbn_activation_forward    x: n frames of per_frame contiguous floats, batch_stride apart; kind KBN_ACT_ELU (v > 0 ? v :
 *                             expm1(v), alpha = 1) or KBN_ACT_SIGMOID (1 / (1 + exp(-v))); out_absmax (may be NULL): the per-frame
 *                             max |a| slots of the tensor (above), which receive the maxima of the ACTIVATED values -- the convs in
 *                             front of this pass are given no slot
 *   kbn_scale_planes_forward  out[n, c, y, x] = x[n, c, y, x] * z[n, 0, y, x], c < channels; dense planes, frames *_batch_stride apart */
enum { KBN_ACT_ELU = 1, KBN_ACT_SIGMOID = 2 };
int kbn_activation_forward(float* x, long long batch_stride, int n, long long per_frame, int kind, unsigned* out_absmax,
                           kbn_stream_t stream);
int kbn_scale_planes_forward(const float* x, long long x_batch_stride, const float* z, long long z_batch_stride, float* out,
                             long long out_batch_stride, int n, int channels, int height, int width, kbn_stream_t stream);

/* ------------------------------------------------------------ up-conv 2x -------
 * net_utils.UpConv2d.forward when the target size is exactly twice the input:
 * interpolate(nearest) + conv3x3 (+ activation)       reference src/net_utils.py:484-499
 * evaluated as four 2x2 convs on the low-resolution input (one per output phase) with
 * pre-summed weights: 4 instead of 9 MACs per output and input channel, no upsampled tensor;
 * maps with 16-byte aligned rows go further: with differences of neighbouring input pixels the
 * two phases of an axis share a product (o0 = -g0 (in[x]-in[x-1]) + G in[x], o1 = g2 (in[x+1]-in[x])
 * + G in[x], G = g0+g1+g2): 3 or 2.25 MACs per output and input channel (csrc/conv_up2x.hip).
 *   src   N x in_channels x src_height x src_width (frames src_batch_stride apart)
 *   out   N x out_channels x 2*src_height x 2*src_width (frames out_batch_stride apart)
 *   packed_weight from kbn_upconv2x_pack_weight (OIHW 3x3 weight in; the blob holds the 4-phase,
 *   the 3-product and, for <= 16 or a multiple of 32 filters, the 9-product weights back to back).
 * Other target sizes go through kbn_conv2d_forward(..., KBN_RESIZE_NEAREST). */
size_t kbn_upconv2x_packed_weight_bytes(int out_channels, int in_channels);
int kbn_upconv2x_pack_weight(const float* weight, float* packed, int out_channels, int in_channels,
                             kbn_stream_t stream);
int kbn_upconv2x_forward(const float* src, long long src_batch_stride, const float* packed_weight,
                         float* out, long long out_batch_stride, int n, int in_channels,
                         int out_channels, int src_height, int src_width, int apply_activation,
                         float negative_slope, unsigned* out_absmax, kbn_stream_t stream);

/* Which algebraic form kbn_upconv2x_forward runs for a problem (diagnostics, roofline accounting):
 * info[4] = {channel products per low-resolution pixel: 16 four 2x2 phases / 12 three-product columns /
 * 9 three-product rows and columns; padded output channels; padded input channels; 0}.  The launch
 * executes 2 * n * src_height * src_width * info[0] * info[1] * info[2] FLOP on the matrix cores. */
int kbn_upconv2x_query(int n, int in_channels, int out_channels, int src_height, int src_width, int* info);

/* ------------------------------------------------------- transposed conv 2x -------
 * net_utils.TransposeConv2d.forward -- the decoder blocks' up-sampling layer with deconv_type='transpose'
 * (run_kbnet.py --deconv_type transpose; reference src/net_utils.py:350-440, DecoderBlock :1468-1469):
 * torch.nn.ConvTranspose2d(in, out, kernel_size=3, stride=2, padding=1, output_padding=1, bias=False) (+ activation),
 *     out[n, o, 2i - 1 + ky, 2j - 1 + kx] += src[n, c, i, j] w[c, o, ky, kx],    output exactly 2 src_height x 2 src_width.
 * By output parity this is four small convs on the source (1, 2, 2 and 4 taps), i.e. the four-phase form of the up-conv
 * above with nine of its sixteen phase weights taken from w and seven zero: same kernels, same blob size
 * (kbn_upconv2x_packed_weight_bytes), always the four-phase form.
 *   weight  out_channels x in_channels x 3 x 3: the module's in x out x 3 x 3 parameter with its first two axes swapped
 *           (the Python mirror does that: ops.pack_upconv2x_weight(transposed=True)). */
int kbn_deconv2x_pack_weight(const float* weight, float* packed, int out_channels, int in_channels,
                             kbn_stream_t stream);
int kbn_deconv2x_forward(const float* src, long long src_batch_stride, const float* packed_weight,
                         float* out, long long out_batch_stride, int n, int in_channels,
                         int out_channels, int src_height, int src_width, int apply_activation,
                         float negative_slope, unsigned* out_absmax, kbn_stream_t stream);

/* ------------------------------- fp32-grade 3x3 convs on the 16-bit matrix core --
 * gfx950 executes fp32 MFMAs on the fp32 vector datapath (157 TFLOP/s); the matrix core proper takes
 * 16-bit operands (2.5 PFLOP/s dense).  These entry points feed it fp32 operands as pairs of fp16 values:
 * a 2^k = h1 + 2^-11 h2 (activations, split in the kernel), w 2^e = w1 + w2 (weights, split at pack time,
 * e per filter), a w = 2^-(e+k) (h1 w1 + h1 w2 + h2 (w1 2^-11)) up to 2^-22 |a w|: three fp16 MFMAs with fp32
 * accumulation per product, 3/16 of the fp32 MFMA's time, errors measured against fp64 no larger than the
 * fp32 MFMA chain's (profiles/r02/bf16x_probe.txt).  They ARE on the parity-gated path: the stride-1 3x3
 * convs of the decoder (reference src/net_utils.py:484-499 nearest-2x + conv, :1483-1487 conv over
 * cat[deconv, skip]) and the stride-2 image convs of the KB blocks run through them when the shape
 * qualifies.
 *   act_exponent  k: places the fp16 window on the activations: |a| 2^k must stay below 65504 (beyond: inf),
 *             |a| 2^k >= 2^-14 keeps the full 22 bits, smaller activations are carried by the scaled residual
 *             alone (absolute error below 2^-40 of the window's top).  USED ONLY WHEN A SOURCE HAS NO absmax SLOT:
 *             with slots on every source the kernel derives k per frame from the data, max |a| 2^k in [2^14, 2^15)
 *             (the Python mirror always provides slots; a static k is for callers that know their range: -6
 *             covers 0.0039 .. 4.2e6).  Range -60 .. 60.
 *   mode      0: 3x3 stride 1 over height x width sources; 1: nearest-2x up-conv in its nine-tap form (ONE source with
 *             (height/2) x (width/2) planes; no longer built since ABI 7: KBN_ERR_UNSUPPORTED, use mode 3); 2: 3x3 stride 2 (source planes h x w with ceil(h/2) = height,
 *             ceil(w/2) = width: the image convs of the KB blocks, reference src/net_utils.py:1348); 3: mode 1 in its folded
 *             form (four 2x2 convs on the low-resolution source, 16 instead of 36 channel products per source pixel: what the
 *             decoder runs); 4: ConvTranspose2d(kernel 3, stride 2, padding 1, output_padding 1) -- deconv_type='transpose',
 *             reference src/net_utils.py:383-390 -- on mode 3's kernels (nine of the sixteen folded weights are the layer's taps,
 *             seven are zero; `weight` for kbn_conv3x3_split_pack_weight: out x in x 3 x 3, the module's parameter with its
 *             first two axes swapped); whatever this text says of mode 3 holds for mode 4
 *   srcs      1 or 2 KBN_SRC_TENSOR sources, each a multiple of 16 channels; source 0 may be a KBN_SRC_PAIR (above)
 *   pair_out  NULL, or the output as a PAIR tensor (then `out` is ignored and may be NULL), with
 *             pair_out_batch_stride (fp16 elements) and pair_out_scale (n floats).  Mode 2 with pair_out (one source): a
 *             non-NULL `out` receives, in fp32, the output pixels (2y, 2x) only, as N x out_channels x ceil(height / 2) x
 *             ceil(width / 2) -- what the next KB level's 1x1 stride-2 conv_fused reads of this tensor
 *             (kbn_conv1x1s2_split_forward takes it as a pre-subsampled source 0)
 *   packed    from kbn_conv3x3_split_pack_weight (OIHW fp32 3x3 weight in) for the SAME mode (the
 *             filter tiling of the blob depends on it)
 *   out       N x out_channels x height x width fp32, frames out_batch_stride elements apart
 * Modes 1 and 3 return KBN_ERR_UNSUPPORTED unless width % 4 == 0 and `out` is 16-byte aligned (callers fall back
 * to kbn_upconv2x_forward); modes 0 and 2 store element-wise in that case.  KBN_NO_SPLIT=1: always unsupported. */
size_t kbn_conv3x3_split_packed_weight_bytes(int out_channels, int in_channels, int mode);
int kbn_conv3x3_split_pack_weight(const float* weight, void* packed, int out_channels, int in_channels, int mode,
                                  kbn_stream_t stream);
int kbn_conv3x3_split_forward(const kbn_conv_src* srcs, int n_src, const void* packed_weight, float* out,
                              long long out_batch_stride, int n, int out_channels, int height, int width, int mode,
                              int act_exponent, int apply_activation, float negative_slope, unsigned* out_absmax,
                              void* pair_out, long long pair_out_batch_stride, float* pair_out_scale, kbn_stream_t stream);
/* The LATENCY form of modes 0 and 2 (round 6): the same conv (reference src/net_utils.py:1483-1487 / :1348, net_utils.Conv2d.forward
 * :120-141) for launches whose tiles cannot fill the chip -- one KITTI frame gives deconv4's conv (768 -> 256 at 22 x 76) 24 workgroups
 * of 48 K-chunks each on 256 CUs.  `ksplit` workgroups share every tile, each summing a contiguous range of the 16-channel chunks
 * into a plane set of its own in `workspace` (ksplit x n x out_channels x height x width floats, 16-byte aligned), and a second kernel
 * adds the sets in split order, applies the activation and fills out_absmax.  fp32 KBN_SRC_TENSOR sources only, no pair output.
 * The result differs from kbn_conv3x3_split_forward's in summation order (same 1e-4 bar, other low bits), so a model uses one form
 * throughout (KBNetModel.latency_mode).  ksplit = 1 is kbn_conv3x3_split_forward; 1 <= ksplit <= in_channels / 16 with no empty range
 * (ceil(chunks / ksplit) * (ksplit - 1) < chunks), else KBN_ERR_INVALID_ARGUMENT. */
int kbn_conv3x3_split_forward_ksplit(const kbn_conv_src* srcs, int n_src, const void* packed_weight, float* out,
                                     long long out_batch_stride, int n, int out_channels, int height, int width, int mode,
                                     int act_exponent, int apply_activation, float negative_slope, unsigned* out_absmax,
                                     int ksplit, float* workspace, kbn_stream_t stream);

/* conv_fused of the KB block on split operands -- reference src/net_utils.py:1337-1343 (Conv2d(in_channels_fused + 3,
 * n_filter_fused, kernel_size=1, stride=2)) applied to cat[image, xyz, fused] (:1352-1368).  The tensor channels
 * (image, fused: one or two KBN_SRC_TENSOR sources of H x W planes, channels % 16 == 0; with two sources, source 0 may
 * instead hold only the pixels the conv reads: src_height x src_width = the OUTPUT size) are taken through the 16-bit
 * matrix core like the 3x3 convs above; the three backprojection channels xyz = K^-1 [x y 1]^T z, z =
 * act(proj_depth . depth) (:1352-1359), are computed once per block at the positions a stride-2 1x1 conv reads
 * (kbn_kb_xyz_s2_forward: xyz[:, :, y, x] belongs to input pixel (2y, 2x)) and enter in fp32.
 *   weight    out_channels x in_channels fp32 (the 1x1 OIHW weight), in_channels = tensor channels (+ 3);
 *             xyz_offset = index of the first of the three xyz input channels (channels of `image`), -1: none
 *   packed    kbn_conv1x1s2_split_packed_weight_bytes(out_channels, tensor_channels, has_xyz) bytes
 *   xyz       N x 3 x height x width fp32 (output size), frames xyz_batch_stride apart; null iff packed without xyz
 *   out       N x out_channels x height x width, height = ceil(H / 2), width = ceil(W / 2) */
size_t kbn_conv1x1s2_split_packed_weight_bytes(int out_channels, int tensor_channels, int has_xyz);
int kbn_conv1x1s2_split_pack_weight(const float* weight, void* packed, int out_channels, int in_channels, int xyz_offset,
                                    kbn_stream_t stream);
int kbn_conv1x1s2_split_forward(const kbn_conv_src* srcs, int n_src, const void* packed_weight, const float* xyz,
                                long long xyz_batch_stride, float* out, long long out_batch_stride, int n, int out_channels,
                                int height, int width, int act_exponent, int apply_activation, float negative_slope,
                                unsigned* out_absmax, kbn_stream_t stream);
int kbn_kb_xyz_s2_forward(const float* depth, long long depth_batch_stride, int depth_channels, int height, int width,
                          const float* proj_weight, const float* kinv, int apply_activation, float negative_slope, float* xyz,
                          long long xyz_batch_stride, int n, kbn_stream_t stream);


/* Which kernel variant / tile geometry kbn_conv2d_forward picks for a problem (diagnostics,
 * profiling): info[8] = {CK, NB, MW, TWB, TH, workgroups, staged positions per thread, kernel};
 * kernel 2 = conv_dma_kernel<kernel_size, stride, CK, NB, MW, ...> (LDS-DMA staging; needs
 * W % 4 == 0, 16-byte aligned planes, no resize), 1 / 0 = conv_igemm_kernel<...> with / without
 * register prefetch, 3 = conv_wino_kernel (Winograd F(2x2,3x3) for 3x3 stride-1 convs with
 * in_channels % 16 == 0, in_channels >= 32, out_channels >= 32, every source a multiple of 8 channels and the conv_dma alignment rules;
 * then MW x TWB = tile rows x columns of a workgroup's region and TH = its pixel rows). */
int kbn_conv2d_query(int n, int out_channels, int in_channels, int kernel_size, int stride,
                     int in_height, int in_width, int resize, int* info);

/* ----------------------------------------------------------- KB block ----------
 * net_utils.CalibratedBackprojectionBlock.forward(image, depth, coordinates, fused)
 *                                                  reference src/net_utils.py:1343-1371
 * conv_image (3x3 s2), conv_depth (3x3 s2 on cat[depth, coordinates]) and conv_fused (1x1 s2 on
 * cat[image, coordinates*z, fused]) with z = act(proj_depth(depth)) evaluated only where the
 * stride-2 conv samples it.  KBNet's shapes (filters_image == filters_fused in {48, 96, 192, 384},
 * aligned rows) run conv_image + conv_fused as ONE kernel over a shared image tile, with a
 * 16-filter conv_depth in the same launch (csrc/kb_pair.hip); anything else takes three
 * kbn_conv2d_forward-style launches.  Same results bit for bit either way.
 *   coordinates     N x 3 x H x W, or NULL to synthesize them from kinv (N x 3 x 3)
 *   fused           N x channels_fused x H x W or NULL (level 0)
 *   packed weights  from kbn_conv2d_pack_weight; proj_weight is the raw (1 x Cd x 1 x 1)
 *   outputs         each at ceil(H/2) x ceil(W/2), with its own batch stride
 *   *_batch_stride  elements between frames (inputs/outputs may be channel slices of a
 *                   larger NCHW tensor, e.g. the encoder's skip buffers; `coordinates`, when
 *                   given, is contiguous N x 3 x H x W)
 *   out_*_absmax    per-frame max |out| slots of the three outputs (each may be NULL; two may be the same slot)
 */
int kbn_kb_block_forward(const float* image, long long image_batch_stride, const float* depth,
                         long long depth_batch_stride, const float* coordinates, const float* kinv,
                         const float* fused, long long fused_batch_stride,
                         const float* packed_w_image, const float* packed_w_depth,
                         const float* proj_weight, const float* packed_w_fused, float* out_image,
                         long long out_image_batch_stride, float* out_depth,
                         long long out_depth_batch_stride, float* out_fused,
                         long long out_fused_batch_stride, int n, int height, int width,
                         int channels_image, int channels_depth, int channels_fused,
                         int filters_image, int filters_depth, int filters_fused,
                         float negative_slope, unsigned* out_image_absmax, unsigned* out_depth_absmax,
                         unsigned* out_fused_absmax, kbn_stream_t stream);

/* ------------------------------------------- encoder front (image branch) ------
 * conv0_image and the two convs of the level-0 KB block that read it, in ONE launch:
 *     conv0_image = act(conv3x3(image))                          reference src/networks.py:364-365
 *     conv_image  = act(conv3x3 s2 (conv0_image))                reference src/net_utils.py:1348
 *     conv_fused  = act(conv1x1 s2 (cat[conv0_image, xyz]))      reference src/net_utils.py:1352-1369 (level 0: no `fused` input)
 * conv0_image (conv0_filters channels at full resolution) is consumed by nothing else, so it is computed per tile and kept
 * on the CU: it never reaches HBM.  All three convs on split fp16 operands (see the split-operand section above; same
 * accuracy class, same parity gate); csrc/front.hip.
 *   image           N x image_channels x H x W (image_channels <= 4), frames image_batch_stride apart
 *                   (the fp16 windows of the image and of conv0's on-chip output follow the data tile by tile, inside the kernel)
 *   packed_weight   from kbn_kb1_front_pack_weight: conv0_image.conv.weight (conv0_filters x image_channels x 3 x 3),
 *                   the block's conv_image weight (kb_filters x conv0_filters x 3 x 3) and conv_fused weight
 *                   (kb_filters x (conv0_filters + 3) x 1 x 1, input order [image channels, xyz])
 *   xyz             N x 3 x ceil(H/2) x ceil(W/2) from kbn_kb_xyz_s2_forward (depth = conv0_depth's output), or NULL
 *   out_image/fused N x kb_filters x ceil(H/2) x ceil(W/2) each, with their own batch strides and absmax slots
 * KBN_ERR_UNSUPPORTED unless conv0_filters == kb_filters == 48 (KBNet's level 0 in all three presets) -- the caller then
 * runs kbn_conv2d_forward + kbn_kb_block_forward. */
size_t kbn_kb1_front_packed_weight_bytes(int image_channels, int conv0_filters, int kb_filters);
/* KBN_OK when kbn_kb1_front_forward would take this problem, KBN_ERR_UNSUPPORTED otherwise (widths, map size, slope outside
 * [0, 1], KBN_NO_SPLIT): lets a caller decide BEFORE it launches the depth branch whose xyz output the kernel consumes. */
int kbn_kb1_front_query(int image_channels, int conv0_filters, int kb_filters, int height, int width, float conv0_negative_slope);
int kbn_kb1_front_pack_weight(const float* w_conv0, const float* w_conv_image, const float* w_conv_fused, void* packed,
                              int image_channels, int conv0_filters, int kb_filters, kbn_stream_t stream);
int kbn_kb1_front_forward(const float* image, long long image_batch_stride, const void* packed_weight, const float* xyz, long long xyz_batch_stride, float* out_image,
                          long long out_image_batch_stride, float* out_fused, long long out_fused_batch_stride, int n,
                          int image_channels, int conv0_filters, int kb_filters, int height, int width,
                          float conv0_negative_slope, float kb_negative_slope, unsigned* out_image_absmax,
                          unsigned* out_fused_absmax, kbn_stream_t stream);

/* The same launch PLUS conv_fused of the NEXT KB level (round 5):
 *     conv_fused_next = act(conv1x1 s2 (cat[conv_image, xyz_next, conv_fused]))     reference src/net_utils.py:1352-1369, the block of
 *                                                                                   level 1 (src/networks.py:401-405)
 * A 1x1 stride-2 conv reads only the pixels (2y, 2x) of its inputs -- for a tile of this kernel the 4 x 8 even pixels of the two
 * outputs its lanes still hold: no halo, no second pass over conv_image / conv_fused (as its own launch the layer is the one
 * HBM-bound conv of the network: it fetches every other ROW of 96 channels at half resolution to use every other pixel of it).
 * The even pixels are split tile by tile (window = the tile's own maximum) and multiplied on the fp16 matrix core like every other
 * split-operand conv; the three backprojection channels of the next level enter in fp32.
 *   packed_next     from kbn_kb1_front_next_pack_weight: the next block's conv_fused weight, filters x (kb_filters + 3 + kb_filters)
 *                   x 1 x 1, input order [conv_image channels, xyz, conv_fused channels]
 *   xyz_next        N x 3 x h2 x w2 (h2 = ceil(ceil(H/2)/2)): kbn_kb_xyz_s2_forward on the level-0 conv_depth output with the next
 *                   block's proj_depth weight and the level-1 inverse intrinsics
 *   out_next_fused  N x next_filters x h2 x w2, frames out_next_fused_batch_stride apart; out_next_fused_absmax its slot (or NULL)
 * KBN_ERR_UNSUPPORTED (kbn_kb1_front_next_query tells beforehand) unless kb_filters == 48 and next_filters == 96 -- KBNet's levels 0 / 1
 * in all presets -- or with KBN_NO_FRONT_NEXT=1: the caller then runs kbn_kb1_front_forward and the next block's conv_fused on its own. */
size_t kbn_kb1_front_next_packed_weight_bytes(int image_channels, int fused_channels, int filters);
int kbn_kb1_front_next_pack_weight(const float* w_conv_fused, void* packed, int image_channels, int fused_channels, int filters,
                                   kbn_stream_t stream);
int kbn_kb1_front_next_query(int image_channels, int conv0_filters, int kb_filters, int next_filters, int height, int width,
                             float conv0_negative_slope);
int kbn_kb1_front_next_forward(const float* image, long long image_batch_stride, const void* packed_weight, const float* xyz,
                               long long xyz_batch_stride, float* out_image, long long out_image_batch_stride, float* out_fused,
                               long long out_fused_batch_stride, int n, int image_channels, int conv0_filters, int kb_filters, int height,
                               int width, float conv0_negative_slope, float kb_negative_slope, unsigned* out_image_absmax,
                               unsigned* out_fused_absmax, const void* packed_next, const float* xyz_next, long long xyz_next_batch_stride,
                               float* out_next_fused, long long out_next_fused_batch_stride, int next_filters, float next_negative_slope,
                               unsigned* out_next_fused_absmax, kbn_stream_t stream);

/* The depth branch of the same front:
 *     conv0_depth = act(conv3x3(depth))                                 reference src/networks.py:366-367
 *     conv_depth  = act(conv3x3 s2 (cat[conv0_depth, coordinates]))     reference src/net_utils.py:1351
 *     xyz         = coordinates * act(proj_depth(conv0_depth))          reference src/net_utils.py:1354-1360, sampled at (2y, 2x)
 * conv0_depth stays on the CU; the tensor channels on split fp16 operands, the three coordinate channels K^-1 [x y 1]^T in fp32.
 *   depth           N x depth_channels x H x W (the S2D output, depth_channels <= 8)
 *   kinv            N x 3 x 3 inverse intrinsics of level 0 (kbn_intrinsics_inverse)
 *   packed_weight   from kbn_kb1_depth_front_pack_weight: conv0_depth.conv.weight (16 x depth_channels x 3 x 3), the block's
 *                   conv_depth weight (16 x (16 + 3) x 3 x 3, input order [depth channels, coordinates]) and proj_depth weight (16)
 *   out_depth       N x 16 x ceil(H/2) x ceil(W/2) (frames out_depth_batch_stride apart), out_depth_absmax its slot (or NULL)
 *   xyz             N x 3 x ceil(H/2) x ceil(W/2): what kbn_kb1_front_forward / kbn_conv1x1s2_split_forward take
 * KBN_ERR_UNSUPPORTED unless conv0_filters == kb_filters == 16. */
size_t kbn_kb1_depth_front_packed_weight_bytes(int depth_channels, int conv0_filters, int kb_filters);
int kbn_kb1_depth_front_query(int depth_channels, int conv0_filters, int kb_filters, int height, int width, float conv0_negative_slope);
int kbn_kb1_depth_front_pack_weight(const float* w_conv0, const float* w_conv_depth, const float* w_proj, void* packed,
                                    int depth_channels, int conv0_filters, int kb_filters, kbn_stream_t stream);
int kbn_kb1_depth_front_forward(const float* depth, long long depth_batch_stride, const float* kinv, const void* packed_weight,
                                float* out_depth, long long out_depth_batch_stride, float* xyz, long long xyz_batch_stride, int n,
                                int depth_channels, int conv0_filters, int kb_filters, int height, int width,
                                float conv0_negative_slope, float kb_negative_slope, int proj_activation, float proj_negative_slope,
                                unsigned* out_depth_absmax, kbn_stream_t stream);

/* The same depth branch with the S2D layer in front of it, in ONE launch (csrc/s2d_stage.h, kb1_depth_front_kernel<pool preset>):
 *     s2d         = SparseToDensePool.forward(x)                        reference src/networks.py:2168-2196, src/kbnet_model.py:161-163
 *     conv0_depth, conv_depth, xyz as above                             reference src/networks.py:366-367, src/net_utils.py:1351-1360
 * Per tile of 8 x 16 half-resolution pixels the workgroup evaluates S2D for the 19 x 35 full-resolution pixels conv0_depth reads
 * (min / max pyramid bit-exact as in kbn_s2d_forward; the 1x1 chain and the 3x3 conv on split fp16 operands) and keeps the result
 * in LDS: the N x 8 x H x W S2D tensor (13.7 MB per KITTI frame written and read back) never reaches HBM.
 *   x               N x 2 x H x W: [sparse depth, validity map] (kbn_s2d_forward's input), frames x_batch_stride apart
 *   packed_s2d      from kbn_s2d_depth_front_pack_weight: pool_convs.{0,1,2}.conv.weight (8 x n_pools | 8 | 8), conv.conv.weight (8 x 10 x 3 x 3)
 *   packed_weight   from kbn_kb1_depth_front_pack_weight (depth_channels = 8)
 *   pool lists      as kbn_s2d_forward; the kernel is compiled for the reference's shipped presets -- KITTI (min 5..13, max 15, 17), VOID / NYUv2
 *                   (min 15, 17, max 23, 27, 29), VOID training (min 15, 17, 19, max 23, 27)
 * KBN_ERR_UNSUPPORTED (kbn_s2d_depth_front_query says so beforehand) for any other pool list, input_channels != 2, n_convolution != 3,
 * n_filter != 8, slopes outside [0, 1], under KBN_NO_DEPTH_FRONT_FUSION=1 or KBN_NO_SPLIT=1, and wherever kbn_kb1_depth_front_forward
 * declines: the caller runs kbn_s2d_forward + kbn_kb1_depth_front_forward. */
size_t kbn_s2d_depth_front_packed_weight_bytes(int n_pools);
int kbn_s2d_depth_front_pack_weight(const float* w_pool_conv0, const float* w_pool_conv1, const float* w_pool_conv2, const float* w_conv,
                                    void* packed, int n_pools, kbn_stream_t stream);
int kbn_s2d_depth_front_query(int input_channels, const int* min_pool_sizes, int n_min, const int* max_pool_sizes, int n_max,
                              int n_convolution, int n_filter, int conv0_filters, int kb_filters, int height, int width,
                              float s2d_negative_slope, float conv0_negative_slope);
int kbn_s2d_depth_front_forward(const float* x, long long x_batch_stride, const float* kinv, const void* packed_s2d, const void* packed_weight,
                                float* out_depth, long long out_depth_batch_stride, float* xyz, long long xyz_batch_stride, int n,
                                int input_channels, const int* min_pool_sizes, int n_min, const int* max_pool_sizes, int n_max,
                                int n_convolution, int n_filter, int conv0_filters, int kb_filters, int height, int width,
                                float s2d_negative_slope, float conv0_negative_slope, float kb_negative_slope, int proj_activation,
                                float proj_negative_slope, unsigned* out_depth_absmax, kbn_stream_t stream);

/* ------------------------------------------------------------ depth head -------
 * MultiScaleDecoder.output0 (3x3, linear)           reference src/networks.py:1842-1851, 1985
 * + KBNetModel.forward's sigmoid / depth mapping    reference src/kbnet_model.py:181-184
 *   depth = d_min / (sigmoid(conv3x3(x, w)) + d_min / d_max)
 * x: N x channels x H x W (channels <= 16), w: 1 x channels x 3 x 3 (raw OIHW).
 * `logits` may be NULL; when given it receives the pre-sigmoid conv output. */
int kbn_depth_head_forward(const float* x, const float* weight, float* depth, float* logits, int n,
                           int channels, int height, int width, float min_predict_depth,
                           float max_predict_depth, kbn_stream_t stream);

/* The decoder's tail in one launch: DecoderBlock deconv0's second conv (channels -> channels, 3x3,
 * optional LeakyReLU; reference src/net_utils.py:1485-1487 with no skip, src/networks.py:1966-1983)
 * + output0 + the depth mapping above; the channels-wide full-resolution tensor between the two convs
 * stays on the CU.  x: N x channels x H x W (frames x_batch_stride elements apart), w_conv: channels x
 * channels x 3 x 3 and w_out: 1 x channels x 3 x 3, both raw OIHW.  KBN_ERR_UNSUPPORTED unless channels is a
 * multiple of 4 (<= 16), width a multiple of 4 and the planes 16-byte aligned: the caller then runs
 * kbn_conv2d_forward + kbn_depth_head_forward (same results up to fp32 summation order). */
int kbn_conv_head_forward(const float* x, long long x_batch_stride, const float* w_conv, const float* w_out,
                          float* depth, float* logits, int n, int channels, int height, int width,
                          int apply_activation, float negative_slope, float min_predict_depth,
                          float max_predict_depth, kbn_stream_t stream);

/* The same tail with the channels -> channels conv on split fp16 operands (csrc/tail.hip; see the split-operand section: same
 * accuracy class, same parity gate): on the fp32 MFMAs that conv was what bound kbn_conv_head_forward.  packed_w_conv from
 * kbn_conv_tail_pack_weight (raw channels x channels x 3 x 3 weight in); w_out raw.  No alignment requirements.
 * KBN_ERR_UNSUPPORTED for channels > 12 (the caller runs kbn_conv_head_forward or the two-launch path). */
size_t kbn_conv_tail_packed_weight_bytes(int channels);
int kbn_conv_tail_pack_weight(const float* w_conv, void* packed, int channels, kbn_stream_t stream);
int kbn_conv_tail_forward(const float* x, long long x_batch_stride, const void* packed_w_conv, const float* w_out, float* depth,
                          float* logits, int n, int channels, int height, int width, int apply_activation, float negative_slope,
                          float min_predict_depth, float max_predict_depth, kbn_stream_t stream);
/* The same launch with the input as a PAIR tensor of 16 channels (two k-groups; channels past `channels` zero): what
 * kbn_conv3x3_split_forward(mode 3, pair_out) writes for a layer of at most 16 filters -- deconv0's up-conv, reference
 * src/net_utils.py:484-499 -- so that the 12-channel full-resolution tensor between the up-conv and the tail is staged by
 * LDS-DMA instead of being loaded, measured and split per tile. */
int kbn_conv_tail_forward_pair(const void* x_pair, long long x_pair_batch_stride, const float* x_pair_scale, const void* packed_w_conv,
                               const float* w_out, float* depth, float* logits, int n, int channels, int height, int width,
                               int apply_activation, float negative_slope, float min_predict_depth, float max_predict_depth,
                               kbn_stream_t stream);

/* ------------------------------------------------- pre-model stage --
 * What the reference's run loop does between the host->device copy and the model call:
 *   validity = where(sparse > 0, 1, sparse)                        reference src/kbnet.py:899-902
 *   OutlierRemoval(kernel_size, threshold).remove_outliers          reference src/net_utils.py:1761-1806
 *     (k x k min filter over the depth with invalid pixels and the padding set to
 *      10 * max(sparse_depth) -- a BATCH-global maximum; a point is dropped if
 *      min < depth - threshold)
 *   image / 255, or 2 (image / 255) - 1 (image_range KBN_IMAGE_RANGE_M1_1)   reference src/transforms.py:201-208
 * image/out_image: N x image_channels x H x W (both may be NULL to skip the normalisation);
 * out_validity: the filtered validity map the model is fed; out_sparse_depth (may be NULL): the
 * filtered sparse depth.  workspace: >= 4 bytes of device memory.  kernel_size odd, <= 15. */
enum { KBN_IMAGE_RANGE_0_1 = 0, KBN_IMAGE_RANGE_M1_1 = 1 };   /* run_kbnet.py --normalized_image_range 0 1 | -1 1 (0 255: pass NULL images) */
int kbn_preprocess_forward(const float* image, const float* sparse_depth, float* out_image,
                           float* out_validity, float* out_sparse_depth, void* workspace,
                           size_t workspace_bytes, int n, int image_channels, int height, int width,
                           int kernel_size, float threshold, int image_range, kbn_stream_t stream);

/* ------------------------------------------------- on-device evaluation
 * The reference's per-sample metrics                  reference src/kbnet.py:932-950,
 *                                                     src/eval_utils.py:20-78
 * over the pixels with ground_truth_validity > 0 and min < ground_truth < max.  ADDS into
 * sums[n*5 + k] (fp64, caller zeroes): k=0 sum|1000(o-g)|, 1 sum(1000(g-o))^2,
 * 2 sum|1/(.001g) - 1/(.001o)|, 3 its square, 4 pixel count.  MAE = s0/s4 (mm),
 * RMSE = sqrt(s1/s4), iMAE = s2/s4 (1/km), iRMSE = sqrt(s3/s4). */
int kbn_eval_accumulate(const float* output_depth, const float* ground_truth,
                        const float* ground_truth_validity, double* sums, int n, int height,
                        int width, float min_evaluate_depth, float max_evaluate_depth,
                        kbn_stream_t stream);

/* ------------------------------------------------- input pipeline ----
 * The reference reads every sample through PIL on one DataLoader worker:
 *   data_utils.load_image   Image.open(path).convert('RGB') -> float32   reference src/data_utils.py:58-85
 *   data_utils.load_depth   16-bit PNG / 256                              reference src/data_utils.py:123-152
 *   datasets.KBNetInferenceDataset.__getitem__: middle third of the image triplet
 *   (load_image_triplet), depth as 1 x H x W, intrinsics from .npy       reference src/datasets.py:22-46, 259-283
 *
 * kbn_png_info / kbn_png_decode: HOST-side PNG reader (zlib inflate + scanline filters; 8-bit
 * gray / RGB / RGBA / palette and 16-bit gray, non-interlaced) for pools of loader threads filling
 * pinned staging buffers.  `pixels` receives height x width x channels uint8 (palette expanded to
 * RGB), or height x width uint16 in host byte order for 16-bit files; *channels / *bit_depth say which.
 * KBN_ERR_UNSUPPORTED for interlaced files and other colour types / depths,
 * KBN_ERR_INVALID_ARGUMENT for damaged or truncated files and short output buffers. */
int kbn_png_info(const unsigned char* file, size_t file_bytes, int* width, int* height, int* channels,
                 int* bit_depth);
int kbn_png_decode(const unsigned char* file, size_t file_bytes, void* pixels, size_t pixels_bytes);
/* `n` files at once on `threads` host threads of the library's own (no interpreter lock involved);
 * status[i] (may be NULL) receives the per-file code, the return value is the first failure. */
int kbn_png_decode_batch(const unsigned char* const* files, const size_t* file_bytes, void* const* pixels,
                         const size_t* pixels_bytes, int n, int threads, int* status);

/* The output side -- data_utils.save_depth, reference src/data_utils.py:154-167 (np.uint32(z * 256.0) ->
 * Image.fromarray(mode='I').save(path): a 16-bit grayscale PNG, samples clipped at 65535), which run_kbnet.py --save_outputs
 * calls for the output, the filtered sparse depth and the ground truth (src/kbnet.py:1018-1026).
 *   kbn_depth_to_u16_forward   DEVICE: count depths (any shape, contiguous) -> samples = min(trunc(z * 256), 65535)
 *   kbn_png_encode_gray16      HOST: height x width samples (host byte order) -> a PNG file image in `out` (one IDAT, filter 0,
 *                              zlib `level` -1 .. 9); *out_bytes = its size.  out_capacity >= kbn_png_encode_gray16_bound(width,
 *                              height), else KBN_ERR_WORKSPACE.  Thread-safe, no GPU work: a pool of host threads writes a batch.
 * Readers (kbn_png_decode, PIL) recover the samples bit for bit; the file's bytes are not PIL's (other filters). */
int kbn_depth_to_u16_forward(const float* depth, unsigned short* samples, long long count, kbn_stream_t stream);
size_t kbn_png_encode_gray16_bound(int width, int height);
int kbn_png_encode_gray16(const unsigned short* pixels, int width, int height, unsigned char* out, size_t out_capacity,
                          size_t* out_bytes, int level);

/* Decoded pixels (device buffers) -> the tensors the reference's dataset returns:
 *   image_u8     N x height x raw_width x image_channels uint8 (1 gray, 3 RGB, 4 RGBA); columns
 *                [x_offset, x_offset + width) are taken (x_offset = width, raw_width = 3 * width for
 *                the middle image of a triplet)  ->  image N x 3 x height x width float32, values 0..255
 *                (normalize=False; gray replicated, alpha dropped = convert('RGB'))
 *   depth_raw    N x height x width uint16 (depth_bits 16) or uint8 (8)  ->  sparse_depth
 *                N x 1 x height x width float32 = value / 256
 * Either pair may be NULL.  kbn_preprocess_forward then derives the validity map, removes
 * outliers and normalises the image. */
int kbn_unpack_frames_forward(const unsigned char* image_u8, const void* depth_raw, float* image,
                              float* sparse_depth, int n, int height, int width, int raw_width,
                              int x_offset, int image_channels, int depth_bits, kbn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KBNET_HIP_H */
