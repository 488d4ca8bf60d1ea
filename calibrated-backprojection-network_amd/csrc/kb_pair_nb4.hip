// kb_pair_nb4.hip -- fused KB block kernels for 64-filter n-tiles (KB3, KB4).
#include "kb_pair_impl.h"

namespace kbn {
int kb_pair_dispatch_nb4(KbPairParams& p, int nbd, int cand, hipStream_t st) {
    switch (nbd) {
        case 0: return pair_dispatch<4, 0>(p, cand, st);
        default: return KBN_ERR_UNSUPPORTED;
    }
}
}  // namespace kbn
