// Instantiations of the 64 x 64-wave-tile forward kernel (conv_fwd_ws2_kernel, conv_ws2.h) for dtype=f16.
#include "conv_ws2.h"

// tr = tile rows: 16 (x 128 output channels per workgroup) or 32 (x 64)
int fi_conv_fwd_ws2_f16(int tr, int wgs_per_cu, const ConvArgs& a, hipStream_t st) {
  if (tr == 16) return launch_conv_fwd_ws2<f16_t, 16, 128>(a, wgs_per_cu, st);
  if (tr == 32) return launch_conv_fwd_ws2<f16_t, 32, 64>(a, wgs_per_cu, st);
  return FI_ERR_UNSUPPORTED;
}
