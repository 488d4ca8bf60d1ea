// Shared device/host helpers for the KBNet gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/kbnet_hip.h"

#define KBN_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return KBN_ERR_LAUNCH;        \
    } while (0)

namespace kbn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

__device__ __forceinline__ float leaky_relu(float v, float slope) { return v > 0.f ? v : v * slope; }

// ---- per-frame activation statistics ("absmax slots") ------------------------------------------------
// A slot is one unsigned per frame: the bit pattern of max |a| over a tensor's frame (bit patterns of non-negative
// floats order like the floats).  The caller zeroes the slots; every conv kernel folds the values it stores into the
// slot of its output tensor (out_absmax) in its epilogue; the split-operand kernels read the slots of their inputs
// (kbn_conv_src.absmax) and place their fp16 window on them (sp_act_scale, conv_split.hip) -- the exponent follows the
// data of THIS forward, frame by frame, with no host round trip and no state between calls.
// `m` >= 0: this thread's maximum (NaNs never enter: fmaxf drops them).  One atomic per wave, and only when the wave
// would raise the slot (the plain load may be stale -- then the atomic is merely redundant).
// Bit pattern of the maximum of a NON-NEGATIVE float over the wave, wave-uniform (an SGPR).  DPP butterflies inside each row of 16
// lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: max is idempotent, any covering pattern does), then the four rows
// meet through v_readlane + s_max_u32.  No LDS crossbar: __shfl_xor is ds_bpermute_b32 with a per-lane address register for every
// distance -- six registers that hipcc computes in a kernel's prologue, and in a kernel at its register limit SPILLS, so that every
// step of the reduction became scratch_load + s_waitcnt vmcnt(0), i.e. a wait for all the stores the epilogue had just issued
// (kb1_front_kernel<.., NEXT>: + 330 us per 32 KITTI frames, round 5).
// PRECONDITION: the whole wave is active at the call (EXEC = all ones).  v_readlane of an inactive lane returns whatever its register
// holds, and lanes 0 / 16 / 32 / 48 are read unconditionally.  Every call site sits in wave-convergent code (after the epilogue's loops,
// with out-of-image lanes contributing 0 instead of leaving); a caller inside divergent control flow must reduce another way.
__device__ __forceinline__ unsigned wave_max_bits(float m) {
#define KBN_DPP_MAX(ctrl) m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), ctrl, 0xf, 0xf, true)))
    KBN_DPP_MAX(0xB1);    // quad_perm [1, 0, 3, 2]
    KBN_DPP_MAX(0x4E);    // quad_perm [2, 3, 0, 1]
    KBN_DPP_MAX(0x141);   // row_half_mirror
    KBN_DPP_MAX(0x140);   // row_mirror
#undef KBN_DPP_MAX
    const int v = __float_as_int(m);
    const unsigned a = (unsigned)__builtin_amdgcn_readlane(v, 0), b = (unsigned)__builtin_amdgcn_readlane(v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane(v, 32), d = (unsigned)__builtin_amdgcn_readlane(v, 48);
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;   // non-negative floats order like their bit patterns
}
// x + its three quad neighbours (lanes 4 q .. 4 q + 3), in every lane of the quad: two DPP quad_perm moves instead of two ds_bpermute
__device__ __forceinline__ float quad_sum(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));   // quad_perm [1, 0, 3, 2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));   // quad_perm [2, 3, 0, 1]
    return x;
}
__device__ __forceinline__ void absmax_commit(unsigned* slot, float m) {
    const unsigned b = wave_max_bits(m);
    if ((threadIdx.x & 63) == 0) {
        if (b > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, b);
    }
}
// max |x| of each of n frames of per_frame contiguous floats into slots[n] (csrc/conv_split.hip): the stand-alone pass
// for tensors whose producer did not fill a slot (inputs of a drop-in module call, outputs of the fallback kernels)
int absmax_frames_launch(const float* x, long long batch_stride, int n, long long per_frame, unsigned* slots, hipStream_t stream);

// ---- per-device one-time kernel setup -------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of (kernel, device): one bit per device
// ordinal of the calling thread's current device, so that a process driving several GPUs (the reference
// wraps its modules in torch.nn.DataParallel, src/kbnet_model.py:408-415) sets it on each of them.
struct DeviceOnce {
    std::atomic<unsigned long long> done[4] = {};   // 256 device ordinals
};
inline int set_max_dynamic_lds(DeviceOnce& once, const void* kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return KBN_ERR_LAUNCH;
    const unsigned long long bit = 1ull << (dev & 63);
    std::atomic<unsigned long long>& word = once.done[(dev >> 6) & 3];
    if (word.load(std::memory_order_relaxed) & bit) return KBN_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return KBN_ERR_LAUNCH;   // not a stream operation: legal while a stream is being captured
    word.fetch_or(bit, std::memory_order_relaxed);
    return KBN_OK;
}
int device_cu_count();   // abi.hip: compute units of the current device (cached per device), <= 0 on error

// ---- debug / experiment knobs ------------------------------------------------------------------------
// Read from the environment ONCE when the library is loaded (and again by kbn_reload_env(), which tests
// and the ablation tools call after changing a variable): no getenv on any launch path.
enum Knob {
    KNOB_DEBUG, KNOB_FORCE_MW, KNOB_FORCE_TWB, KNOB_EPI_LDS, KNOB_NO_WINO,
    KNOB_NO_UP2X9, KNOB_NO_UP2X3, KNOB_WINO_RT,
    KNOB_NO_KB_PAIR, KNOB_NO_KB_DEPTH_FUSION, KNOB_PAIR_CAND, KNOB_S2D_DEBUG, KNOB_AUTOTUNE,
    KNOB_NO_HEAD_FUSION, KNOB_NO_SPLIT,
    // switches of the host mirror (modules.py asks kbn_knob(): one reading of the environment for both sides)
    KNOB_NO_OVERLAP, KNOB_NO_PAIR, KNOB_NO_PAIR_MID, KNOB_NO_PAIR_ENC, KNOB_NO_PAIR_TAIL, KNOB_NO_DEPTH_FRONT_FUSION,
    KNOB_FP16_ONE_TERM, KNOB_DEPTH_FRONT_FUSION, KNOB_NO_FRONT_NEXT, KNOB_COUNT
};
struct KnobValue { int set, value; };
extern KnobValue g_knobs[KNOB_COUNT];
inline int knob(Knob k) { return g_knobs[k].value; }        // 0 when unset
inline bool knob_set(Knob k) { return g_knobs[k].set != 0; }

// XCD-aware block remap (bijective for any grid size): the dispatcher places block b on
// XCD b % 8; give every XCD one contiguous range of logical tiles so that neighbouring
// tiles (which share input halos / weight panels) hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int NX = 8;
    int xcd = bid % NX, idx = bid / NX;
    int q = nblocks / NX, r = nblocks % NX;
    int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// PyTorch's nearest-neighbour source index (aten/src/ATen/native/UpSample.h
// nearest_idx): identity when sizes match, >>1 for an exact 2x, else
// min(floor(dst * (float)in/out), in-1).
__device__ __forceinline__ int nearest_src_index(int dst, int in_size, int out_size) {
    if (in_size == out_size) return dst;
    if (out_size == 2 * in_size) return dst >> 1;
    float scale = (float)in_size / (float)out_size;
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

// conv_igemm.hip
int conv2d_launch(const kbn_conv_src* srcs, int n_src, const float* packed_weight, float* out,
                  long long out_batch_stride, int n, int out_channels, int kernel_size, int stride,
                  int in_height, int in_width, int resize, int apply_activation, float negative_slope,
                  unsigned* out_absmax, hipStream_t stream);

}  // namespace kbn
