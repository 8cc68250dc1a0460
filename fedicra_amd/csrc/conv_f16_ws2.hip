// Instantiations of the 64 x 64-wave-tile forward kernel (conv_fwd_ws2_kernel, conv_ws2.h) for dtype=f16.
#include "conv_ws2.h"

// form: 1 = 16-row tiles x 128 output channels per workgroup, 2 = 32 rows x 64; with the whole filter resident in LDS:
// 3 = 32 rows x 64 (Cout == 64), 4 = 32 rows x 32 (Cout == 32); 5 = 16 rows x 128 with the slabs walked inside a tile (Cin == 64)
int fi_conv_fwd_ws2_f16(int form, int wgs_per_cu, const ConvArgs& a, hipStream_t st) {
  if (form == 1) return launch_conv_fwd_ws2<f16_t, 16, 128, 0>(a, wgs_per_cu, st);
  if (form == 2) return launch_conv_fwd_ws2<f16_t, 32, 64, 0>(a, wgs_per_cu, st);
  if (form == 3) return launch_conv_fwd_ws2<f16_t, 32, 64, 1>(a, wgs_per_cu, st);
  if (form == 4) return launch_conv_fwd_ws2<f16_t, 32, 32, 1>(a, wgs_per_cu, st);
  if (form == 5) return launch_conv_fwd_ws2<f16_t, 16, 128, 2>(a, wgs_per_cu, st);
  return FI_ERR_UNSUPPORTED;
}
