// Instantiations of the row-streaming thin-layer filter gradient (conv_wgrad_rows_kernel, wgrad_rows.h) for dtype=f16.
#include "wgrad_rows.h"

int fi_conv_wgrad_rows_f16(int nci, int nco, const WgRowsArgs& a, int items, hipStream_t st) {
  if (nci == 1 && nco == 1) return launch_conv_wgrad_rows<f16_t, 1, 1>(a, items, st);
  if (nci == 2 && nco == 1) return launch_conv_wgrad_rows<f16_t, 2, 1>(a, items, st);
  if (nci == 1 && nco == 2) return launch_conv_wgrad_rows<f16_t, 1, 2>(a, items, st);
  if (nci == 2 && nco == 2) return launch_conv_wgrad_rows<f16_t, 2, 2>(a, items, st);
  return FI_ERR_UNSUPPORTED;
}

// (input blocks, gradient blocks) of a workgroup's tile; slices up to 128 wide for the 16-channel gradients, up to 64 for the 32-channel tiles
int fi_conv_wgrad_rows3d_f16(int nci, int nco, const WgRowsArgs& a, int items, hipStream_t st) {
  if (nco == 1 && nci == 1) return launch_conv_wgrad_rows3d<f16_t, 1, 1, 128>(a, items, st);
  if (nco == 1 && nci == 2) return launch_conv_wgrad_rows3d<f16_t, 2, 1, 128>(a, items, st);
  if (nco == 1 && nci == 3) return launch_conv_wgrad_rows3d<f16_t, 3, 1, 128>(a, items, st);
  if (nco == 2 && nci == 1) return launch_conv_wgrad_rows3d<f16_t, 1, 2, 64>(a, items, st);
  if (nco == 2 && nci == 2) return launch_conv_wgrad_rows3d<f16_t, 2, 2, 64>(a, items, st);
  return FI_ERR_UNSUPPORTED;
}

// narrow = 1: <= 4 input channels (first convolution), 2: <= 4 gradient channels (logits convolution); the other side 16
int fi_conv_wgrad_rows_narrow_f16(int narrow, const WgRowsArgs& a, int items, hipStream_t st) {
  if (narrow == 1) return launch_conv_wgrad_rows<f16_t, 1, 1, 1>(a, items, st);
  if (narrow == 2) return launch_conv_wgrad_rows<f16_t, 1, 1, 2>(a, items, st);
  return FI_ERR_UNSUPPORTED;
}

// channel-rich layers: (gradient, input) channel tile in 16-channel blocks -- (4, 4), (2, 4), (4, 2)
int fi_conv_wgrad_rows64_f16(int tco, int tci, const WgRowsArgs& a, int items, hipStream_t st) {
  if (tco == 4 && tci == 4) return launch_conv_wgrad_rows64<f16_t, 4, 4>(a, items, st);
  if (tco == 2 && tci == 4) return launch_conv_wgrad_rows64<f16_t, 2, 4>(a, items, st);
  if (tco == 4 && tci == 2) return launch_conv_wgrad_rows64<f16_t, 4, 2>(a, items, st);
  return FI_ERR_UNSUPPORTED;
}
