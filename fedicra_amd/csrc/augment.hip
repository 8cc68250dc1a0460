// Training-time augmentation of the reference's data path (/root/reference/code/dataloaders/dataset.py:190-256) for a
// data set that lives in HBM: the whole FAZ set is 1332 x 256^2 floats = 350 MB, so instead of DataLoader workers
// rotating one numpy image at a time and a pinned-memory copy per batch, a batch is ONE gather out of the resident set.
//   RandomGenerator (:231-256): with probability 1/2 random_rot_flip (np.rot90 by k, then np.flip along one spatial
//   axis, :190-207), then with probability 1/2 random_rotate (scipy.ndimage.rotate by an integer angle in [-45,45),
//   order 0, reshape=False, constant padding: 0.8 / ignore label for faz, 0 / ignore label for odoc and polyp, :210-228).
// Output pixel (i,j) of sample b:
//   1. rotate: input coordinate r = ((0 + i*m00) + j*m01) + off0, c likewise -- fp64, the operations scipy's
//      NI_GeometricTransform performs in that order, NO fma contraction (fp contract(off) below); outside
//      [0, len-1] the pixel takes the constant, else the nearest sample floor(x + 0.5).
//   2. (r,c) indexes A = flip(rot90(img, k), axis); undo the flip, then the quarter turns, to reach the stored image.
// One thread per output pixel, all channels; writes are coalesced, reads follow a rotated line (cache-friendly at these
// angles).  HBM-bound: (C*4 + 1) bytes read + written per pixel.
#include "common.h"

__global__ __launch_bounds__(256) void augment2d_kernel(const float* __restrict__ src_img, const uint8_t* __restrict__ src_lab,
                                                        const int* __restrict__ ip, const double* __restrict__ dp,
                                                        float* __restrict__ out_img, uint8_t* __restrict__ out_lab, int C,
                                                        int H, int W, float img_cval, int lab_cval) {
#pragma clang fp contract(off)          // scipy's C evaluates mul, add separately: a fused multiply-add changes roundings
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= H * W) return;
  const int i = pix / W, j = pix - i * W;
  const int src = ip[b * 4 + 0], k = ip[b * 4 + 1], axis = ip[b * 4 + 2], rot = ip[b * 4 + 3];
  int r = i, c = j;
  bool constant = false;
  if (rot) {
    const double* m = dp + (size_t)b * 6;
    const double di = (double)i, dj = (double)j;
    const double rr = ((0.0 + di * m[0]) + dj * m[1]) + m[4];
    const double cc = ((0.0 + di * m[2]) + dj * m[3]) + m[5];
    constant = rr < 0.0 || rr > (double)(H - 1) || cc < 0.0 || cc > (double)(W - 1);
    r = (int)floor(rr + 0.5);
    c = (int)floor(cc + 0.5);
  }
  const size_t plane = (size_t)H * W;
  if (constant) {
    for (int ch = 0; ch < C; ++ch) out_img[((size_t)b * C + ch) * plane + pix] = img_cval;
    out_lab[(size_t)b * plane + pix] = (uint8_t)lab_cval;
    return;
  }
  if (k >= 0) {
    // A = flip(R, axis), R = rot90(img, k); R has the image's shape for even k and (W, H) for odd k (then H == W).
    const int Hr = (k & 1) ? W : H, Wr = (k & 1) ? H : W;
    if (axis == 0) r = Hr - 1 - r; else c = Wr - 1 - c;
    int si, sj;
    switch (k & 3) {
      case 0: si = r, sj = c; break;
      case 1: si = c, sj = W - 1 - r; break;           // R[r][c] = img[c][W-1-r]
      case 2: si = H - 1 - r, sj = W - 1 - c; break;
      default: si = H - 1 - c, sj = r; break;          // R[r][c] = img[H-1-c][r]
    }
    r = si, c = sj;
  }
  const size_t sp = (size_t)r * W + c;
  for (int ch = 0; ch < C; ++ch) out_img[((size_t)b * C + ch) * plane + pix] = src_img[((size_t)src * C + ch) * plane + sp];
  out_lab[(size_t)b * plane + pix] = src_lab[(size_t)src * plane + sp];
}

extern "C" int fi_augment2d(const float* src_img, const uint8_t* src_lab, const int* ip, const double* dp, float* out_img,
                            uint8_t* out_lab, int B, int C, int H, int W, float img_cval, int lab_cval, void* stream) {
  if (!src_img || !src_lab || !ip || !dp || !out_img || !out_lab) return FI_ERR_NULL;
  if (B < 1 || B > 65535 || C < 1 || H < 1 || W < 1 || (long)H * W > (1L << 30) || lab_cval < 0 || lab_cval > 255)
    return FI_ERR_SHAPE;
  hipLaunchKernelGGL(augment2d_kernel, dim3((unsigned)((H * W + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     src_img, src_lab, ip, dp, out_img, out_lab, C, H, W, img_cval, lab_cval);
  FI_CHECK_LAUNCH();
  return 0;
}
