// Instantiations of the LDS-DMA GEMM-tile forward kernel (conv_fwd_dma_kernel, conv_dma.h) for dtype=f16.
#include "conv_dma.h"

int fi_conv_fwd_dma_f16(int wgs_per_cu, const ConvArgs& a, hipStream_t st) { return launch_conv_fwd_dma<f16_t>(a, wgs_per_cu, st); }
