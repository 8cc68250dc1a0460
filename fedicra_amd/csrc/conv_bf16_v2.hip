// Instantiations of the persistent forward kernel (conv_fwd_v2_kernel, conv_impl.h) for dtype=bf16, ksize=3.
#include "conv_impl.h"
#include "conv_narrow.h"

#define V2_CASE(NF_, CK_) if (nf == NF_ && ck == CK_) return launch_conv_fwd_v2<bf16_t, 3, NF_, CK_>(a, wgs_per_cu, st);

int fi_conv_fwd_v2_bf16_k3(int nf, int ck, int wgs_per_cu, const ConvArgs& a, hipStream_t st) {
  V2_CASE(1, 16) V2_CASE(2, 16) V2_CASE(4, 16) V2_CASE(1, 32) V2_CASE(2, 32) V2_CASE(4, 32)
  return FI_ERR_UNSUPPORTED;
}

#define THIN_CASE(NF_, CK_) if (nf == NF_ && ck == CK_) return launch_conv_thin<bf16_t, NF_, CK_>(a, wgs_per_cu, st);

int fi_conv_thin_bf16(int nf, int ck, int wgs_per_cu, const ConvArgs& a, hipStream_t st) {
  THIN_CASE(1, 16) THIN_CASE(2, 16) THIN_CASE(1, 32) THIN_CASE(2, 32)
  return FI_ERR_UNSUPPORTED;
}

#define WS_CASE(NF_, CK_)                                                         \
  if (nf == NF_ && ck == CK_) {                                                   \
    if (pw == 44) return launch_conv_fwd_ws<bf16_t, NF_, CK_, 44>(a, wgs_per_cu, st); \
    if (pw == 8) return launch_conv_fwd_ws<bf16_t, NF_, CK_, 8>(a, wgs_per_cu, st);   \
    return launch_conv_fwd_ws<bf16_t, NF_, CK_, 4>(a, wgs_per_cu, st);                \
  }

// pw = producer waves per workgroup: 4 or 8 beside 4 consumer waves; 44 = two alternating teams of 4 beside 8 consumer waves
int fi_conv_fwd_ws_bf16(int nf, int ck, int pw, int wgs_per_cu, const ConvArgs& a, hipStream_t st) {
  WS_CASE(2, 16) WS_CASE(4, 16) WS_CASE(1, 32) WS_CASE(2, 32) WS_CASE(4, 32)
  return FI_ERR_UNSUPPORTED;
}

// the logits convolution (fp32 outputs, Cout <= 4) in the thin-layer form; the narrow-input layers (conv_narrow.h)
int fi_conv_thin_f32n_bf16(int ck, int wgs_per_cu, const ConvArgs& a, hipStream_t st) {
  if (ck == 16) return launch_conv_thin_f32n<bf16_t, 16>(a, wgs_per_cu, st);
  if (ck == 32) return launch_conv_thin_f32n<bf16_t, 32>(a, wgs_per_cu, st);
  return FI_ERR_UNSUPPORTED;
}
int fi_conv_narrow_in_bf16(const ConvArgs& a, hipStream_t st) { return launch_conv_narrow_in<bf16_t>(a, st); }
