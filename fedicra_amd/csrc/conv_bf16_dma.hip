// Instantiations of the LDS-DMA GEMM-tile forward kernel (conv_fwd_dma_kernel, conv_dma.h) for dtype=bf16.
#include "conv_dma.h"

int fi_conv_fwd_dma_bf16(int wgs_per_cu, const ConvArgs& a, hipStream_t st) { return launch_conv_fwd_dma<bf16_t>(a, wgs_per_cu, st); }
