// conv_dma.hip -- the fast path of the fp32 implicit-GEMM conv: both operands reach LDS by
// LDS-DMA (`global_load_lds_dwordx4`), no VGPR round trip, no ds_write pass.
//
// Same GEMM decomposition, weight packing and epilogue as conv_igemm.hip; what differs is the
// staging of the input tile.  Eligible launches (conv2d_launch checks): tensor sources whose
// rows are 16-byte aligned (W % 4 == 0, aligned base, batch stride % 4 == 0), no resize.
//
//  * The staged tile of one channel is ROW-MAJOR [rows][colsV] with colsV a multiple of 4 that
//    starts at an x that is a multiple of 4, so every 16-byte LDS-DMA granule is a run of 4
//    in-image (or 4 out-of-image) pixels of one row.  A wave moves 64 granules = 1 KiB per DMA
//    instruction; lane l's granule lands at lds_base + 16*l, i.e. the tile is its own LDS image.
//  * Zero padding: both stage buffers are cleared once per workgroup.  Out-of-image granules
//    are simply never written (the lane is masked off), so they stay zero for every chunk --
//    the padding of a tile depends on its position, not on the channel.
//  * Stride-2 convs keep the row-major image (a DMA cannot de-interleave); their A fragments
//    read every other word, a 2-way bank conflict that the LDS has ample headroom for.
//  * KB-layer channels that are computed rather than loaded (K^-1 [x y 1]^T and its product
//    with z = act(proj . depth)) are written with ordinary ds_write by all threads of the chunk
//    that contains them (reference src/net_utils.py:1351-1360).
//  * Double buffered: chunk c+1's DMAs are issued before chunk c's MFMAs; one
//    s_waitcnt vmcnt(0) + barrier per chunk.
#include "conv_common.h"

namespace kbn {

template <int TWB>
int conv_dma_launch_twb(ConvParams& p, const ConvPlan& pl, int MW, int kernel_size, int stride, bool syn,
                        hipStream_t stream);   // conv_dma_t1/t2/t4.hip

int conv_dma_launch(ConvParams& p, const ConvPlan& pl, TileChoice tc, int kernel_size, int stride,
                    hipStream_t stream) {
    // eligibility: aligned tensor planes, no resize
    if (p.resize || (p.inW & 3)) return KBN_ERR_UNSUPPORTED;
    for (int s = 0; s < p.nsrc; ++s) {
        const SrcDev& d = p.src[s];
        if (d.kind == KBN_SRC_TENSOR) {
            if ((reinterpret_cast<uintptr_t>(d.data) & 15) || (d.bstride & 3)) return KBN_ERR_UNSUPPORTED;
        }
    }
    // generic staging (SYN) if a source is computed, or a chunk would straddle two sources
    bool syn = false;
    for (int s = 0; s < p.nsrc; ++s) {
        syn = syn || p.src[s].kind != KBN_SRC_TENSOR;
        if (s + 1 < p.nsrc && (p.src[s].cstart + p.src[s].C) % pl.CK != 0) syn = true;
    }
    switch (tc.TWB) {
        case 1: return conv_dma_launch_twb<1>(p, pl, tc.MW, kernel_size, stride, syn, stream);
        case 2: return conv_dma_launch_twb<2>(p, pl, tc.MW, kernel_size, stride, syn, stream);
        case 4: return conv_dma_launch_twb<4>(p, pl, tc.MW, kernel_size, stride, syn, stream);
        default: return KBN_ERR_UNSUPPORTED;
    }
}

}  // namespace kbn
