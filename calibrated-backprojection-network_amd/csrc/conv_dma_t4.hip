// conv_dma_t4.hip -- the LDS-DMA conv kernels with tiles 64 pixels wide (see conv_dma.hip / conv_dma_impl.h).
#include "conv_dma_impl.h"

namespace kbn {
template int conv_dma_launch_twb<4>(ConvParams&, const ConvPlan&, int, int, int, bool, hipStream_t);
}
