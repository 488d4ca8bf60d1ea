// s2d_pools.h -- compile-time pool lists of the reference's shipped S2D presets, shared by csrc/s2d.hip (SparseToDensePool.forward)
// and csrc/front.hip's kb1_depth_front_kernel<pool preset> through csrc/s2d_stage.h (S2D -> conv0_depth -> KB1 depth branch in one launch).
#pragma once

namespace kbn {

struct S2DParams;

// Pool configuration: compile-time lists for the reference's shipped presets (the register-blocked passes
// unroll to straight-line code), a run-time list for anything else (same phases, plain loops).
template <int NMIN, int... KS>
struct StaticPools {
    static constexpr bool is_static = true;
    static constexpr int NP = sizeof...(KS);
    static constexpr int K[NP] = {KS...};
    static constexpr int cmax(int lo, int hi) {
        int m = 0;
        for (int i = lo; i < hi; ++i) m = (K[i] / 2 > m) ? K[i] / 2 : m;
        return m;
    }
    static constexpr int NMINP = NMIN;
    static constexpr int RMIN = cmax(0, NMIN), RMAX = cmax(NMIN, NP);
    static constexpr int RR = RMIN > RMAX ? RMIN : RMAX;
    static constexpr int RMAXZ = RR;          // radius the depth-tile register prefetch is sized for
    static constexpr int radius(int pi) { return K[pi] / 2; }
    __device__ static constexpr int R(const S2DParams&) { return RR; }
};
using KittiPools = StaticPools<5, 5, 7, 9, 11, 13, 15, 17>;   // bash/kitti/run_kbnet_kitti_validation.sh:15-16
using VoidPools = StaticPools<2, 15, 17, 23, 27, 29>;         // bash/void/run_kbnet_void1500.sh:15-16
using VoidTrainPools = StaticPools<3, 15, 17, 19, 23, 27>;    // bash/void/train_kbnet_void1500.sh:21-22

template <int N>
struct IntC { static constexpr int value = N; };
template <int I, int N, typename F>
__device__ __forceinline__ void s2d_for(F&& f) {
    if constexpr (I < N) {
        f(IntC<I>{});
        s2d_for<I + 1, N>(static_cast<F&&>(f));
    }
}

}  // namespace kbn
