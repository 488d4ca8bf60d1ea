// conv_dma_t1.hip -- the LDS-DMA conv kernels with tiles 16 pixels wide (see conv_dma.hip / conv_dma_impl.h).
#include "conv_dma_impl.h"

namespace kbn {
template int conv_dma_launch_twb<1>(ConvParams&, const ConvPlan&, int, int, int, bool, hipStream_t);
}
