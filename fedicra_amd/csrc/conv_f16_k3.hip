// Instantiations of the implicit-GEMM conv kernels for dtype=fp16, ksize=3 (see conv_impl.h).
#include "conv_impl.h"

#define FWD_CASE(TH_, NF_, CK_) \
  if (th == TH_ && nf == NF_ && ck == CK_) return launch_conv_fwd<f16_t, 3, TH_, NF_, CK_>(a, st);
#define FWD_CK(TH_, NF_) FWD_CASE(TH_, NF_, 8) FWD_CASE(TH_, NF_, 16) FWD_CASE(TH_, NF_, 32)
// deep layers: channel chunks twice as wide = half as many sequential staging round trips (narrow slabs only: LDS)
#define FWD_NF(TH_) FWD_CK(TH_, 1) FWD_CK(TH_, 2) FWD_CK(TH_, 4) FWD_CASE(TH_, 1, 64) FWD_CASE(TH_, 2, 64)

int fi_conv_fwd_f16_k3(int th, int nf, int ck, const ConvArgs& a, hipStream_t st) {
  FWD_NF(4) FWD_NF(8) FWD_NF(16)
  return FI_ERR_UNSUPPORTED;
}

#define WG_CASE(TH_, NFO_, NFI_) \
  if (th == TH_ && nfo == NFO_ && nfi == NFI_) return launch_conv_wgrad<f16_t, 3, TH_, NFO_, NFI_>(a, st);
#define WG_TH(TH_) WG_CASE(TH_, 1, 1) WG_CASE(TH_, 1, 2)

int fi_conv_wgrad_f16_k3(int th, int nfo, int nfi, const WgradArgs& a, hipStream_t st) {
  WG_TH(4) WG_TH(8) WG_TH(16)
  return FI_ERR_UNSUPPORTED;
}

#define QUAD_CASE(TH_) if (th == TH_) return launch_conv_wgrad_quad<f16_t, 3, TH_>(a, st);

int fi_conv_wgrad_quad_f16_k3(int th, const WgradArgs& a, hipStream_t st) {
  QUAD_CASE(4) QUAD_CASE(8) QUAD_CASE(16)
  return FI_ERR_UNSUPPORTED;
}
