// kb_pair_nb3.hip -- fused KB block kernels for 48-filter n-tiles (KB1, KB2): without / with a 16-filter conv_depth.
#include "kb_pair_impl.h"

namespace kbn {
int kb_pair_dispatch_nb3(KbPairParams& p, int nbd, int cand, hipStream_t st) {
    switch (nbd) {
        case 0: return pair_dispatch<3, 0>(p, cand, st);
        case 1: return pair_dispatch<3, 1>(p, cand, st);
        default: return KBN_ERR_UNSUPPORTED;
    }
}
}  // namespace kbn
