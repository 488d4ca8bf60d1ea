// kb_pair.hip -- the three convs of a calibrated-backprojection block in ONE launch
// (kernel: kb_pair_impl.h; instantiations: kb_pair_nb3.hip / kb_pair_nb4.hip; this file: eligibility + launch).
//
// reference src/net_utils.py:1343-1371:
//     conv_image = act(conv3x3 s2 (image))
//     conv_depth = act(conv3x3 s2 (cat[depth, coordinates]))
//     conv_fused = act(conv1x1 s2 (cat[image, coordinates * act(proj_depth(depth)), fused]))
// Launched separately, the 1x1 stride-2 conv is HBM-bound on re-reading `image` (every other row of a
// tensor the 3x3 conv has just streamed: 329 MB of the 603 MB it moves at KB1, batch 8) and, on the
// coarse levels, on its own prologue.  But a 1x1 stride-2 conv reads exactly the pixels under the CENTRE
// TAP of a 3x3 stride-2 pad-1 conv.  So one workgroup keeps two sets of accumulators over the same pixel tile:
//
//   phase 1  K loop over the image channels (4 per chunk, tile + halo staged by LDS-DMA as in conv_dma):
//            9 taps x image weights -> acc_image;  centre tap x fused weights[image part] -> acc_fused
//            (the A fragment of the centre tap is read once and feeds both).
//   phase 2  K loop over the remaining conv_fused inputs, 8 per chunk, staged with the 1x1 stride-2
//            geometry (even rows only): xyz (3 channels) then `fused`.  xyz = K^-1 [x y 1]^T * act(proj . depth)
//            is evaluated in the prologue -- while the first DMAs are in flight -- at the pixels the conv
//            samples, parked in LDS and copied into its chunk.
//
//   phase 0  (KB1 only: a 16-filter conv_depth; workgroups whose n-tile index is below conv_depth's n-tile count)
//            conv_depth over the same pixel tile, before phase 1: depth planes by LDS-DMA, the three coordinate
//            planes K^-1 [x y 1]^T written by the staging waves (or DMA'd when the caller passes dense
//            coordinates), 9 taps -> acc_depth.  Launched on its own, KB1's conv_depth is HBM/latency-bound at
//            full resolution (92 us for 4.7 GFLOP); here its MFMAs ride along in a kernel that is already resident.
//
// All packed weight blobs are the ones kbn_conv2d_pack_weight makes for the two convs (the 1x1 blob is
// a sequence of 4-channel groups, so any multiple-of-4 channel offset is a valid slice): nothing new at
// pack time, and the three-launch path (unaligned shapes, other filter counts) reads the same blobs.
// Accumulation order per output = the order of the separate kernels (channels ascending, 4 per MFMA, taps
// inside), so both paths give bit-identical results (tests/test_hip_parity.py).
#include <stdlib.h>

#include "kb_pair_impl.h"

namespace kbn {
namespace {
inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
}  // namespace

int kb_pair_launch(const KbPairArgs& a, hipStream_t stream, bool* depth_done) {
    *depth_done = false;
    if (knob(KNOB_NO_KB_PAIR)) return KBN_ERR_UNSUPPORTED;
    // shapes: conv_image and conv_fused must pack with the same n-block count (3 or 4: 48 / 96 / 192 / 384 filters
    // and the like), image channels a whole number of 4-channel chunks, rows of every DMA source 16-byte aligned
    if (a.channels_image < 4 || (a.channels_image & 3) || (a.width & 3)) return KBN_ERR_UNSUPPORTED;
    const ConvPlan plI = make_plan(a.filters, a.channels_image, 3, 2);
    const ConvPlan plF = make_plan(a.filters, a.channels_image + 3 + a.channels_fused, 1, 2);
    if (plI.CK != 4 || plF.CK != 16 || plI.NB != plF.NB || plI.NB < 3 || plI.Cpad != a.channels_image)
        return KBN_ERR_UNSUPPORTED;
    if (!aligned16(a.image) || (a.image_bstride & 3)) return KBN_ERR_UNSUPPORTED;
    if (a.fused && (!aligned16(a.fused) || (a.fused_bstride & 3))) return KBN_ERR_UNSUPPORTED;
    if (!aligned16(a.wp_image) || !aligned16(a.wp_fused)) return KBN_ERR_UNSUPPORTED;
    if ((plF.Cpad - a.channels_image) % 8 != 0) return KBN_ERR_UNSUPPORTED;   // phase 2 runs in 8-channel chunks
    KbPairParams p{};
    p.a = a;
    p.CpadF = plF.Cpad;
    p.outH = ceil_div(a.height, 2);
    p.outW = ceil_div(a.width, 2);
    p.nTilesN = plI.nTilesN;
    // conv_depth rides along when it has one n-block (<= 16 filters: KB1, whose separate conv_depth is HBM-bound at full
    // resolution, 92 -> 51 us).  Measured for the wider ones (32 / 64 / 128 filters at KB2-4, the kernel takes any
    // NBD <= NB): +5 / +12 / -5 us -- the extra accumulators cost occupancy -- so those keep their own launch.
    int nbd = 0;
    if (a.wp_depth && a.out_depth && !knob(KNOB_NO_KB_DEPTH_FUSION)) {
        const ConvPlan plD = make_plan(a.filters_depth, a.channels_depth + 3, 3, 2);
        const bool fits = plI.NB == 3 && plD.NB == 1;
        if (plD.CK == 4 && fits && plD.nTilesN <= plI.nTilesN && aligned16(a.depth) && (a.depth_bstride & 3) == 0 &&
            aligned16(a.wp_depth) && (!a.coords || (aligned16(a.coords) && (a.coords_bstride & 3) == 0))) {
            nbd = plD.NB;
            p.CpadD = plD.Cpad;
            p.nTilesND = plD.nTilesN;
        }
    }
    auto launch = [&](int cand) -> int {
        KbPairParams q = p;
        return plI.NB == 3 ? kb_pair_dispatch_nb3(q, nbd, cand, stream) : kb_pair_dispatch_nb4(q, nbd, cand, stream);
    };
    // analytic choice: the largest tile that still gives every CU's two workgroup slots >= 2 rounds
    int model = 5;
    for (int c = 0; c < kPairCands; ++c) {
        if (kPairMW[c] * (2 * plI.NB + nbd) > 40) continue;      // not instantiated (register budget)
        const int th = 4 * kPairMW[c] / kPairTWB[c], tw = 16 * kPairTWB[c];
        const long long wgs = (long long)ceil_div(p.outW, tw) * ceil_div(p.outH, th) * a.n * p.nTilesN;
        if (wgs >= 1024) { model = c; break; }
    }
    int cand;
    const bool forced = knob_set(KNOB_PAIR_CAND) && knob(KNOB_PAIR_CAND) >= 0 && knob(KNOB_PAIR_CAND) < kPairCands;
    if (forced) {   // test hook: force a tile shape (tests/test_hip_parity.py)
        cand = knob(KNOB_PAIR_CAND);
    } else {
        cand = tune_pick(TuneKey{5, a.n, a.filters, a.channels_image, a.channels_fused, a.channels_depth, a.height,
                                 a.width, a.coords ? 1 : 0, nbd},
                         kPairCands, model, launch, stream);
    }
    int rc = launch(cand);
    if (rc == KBN_ERR_UNSUPPORTED && forced) rc = launch(model);   // forced shape not instantiated for this block
    if (rc == KBN_OK) *depth_done = nbd > 0;
    return rc;
}

}  // namespace kbn
