// Gated CRF loss (Obukhov et al. 2019) as the reference's `_Ours` procedure calls it -- Potts compatibility, no masks
// (/root/reference/code/utils/gate_crf_loss.py:20-124; call site flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours.py:143-150
// with radius 5, xy sigma 6, rgb sigma 0.1).  The reference unfolds the prediction and the kernel features into
// N*C*(2r+1)^2*H*W tensors (12x2x121x256^2 floats = 762 MB); here one workgroup stages a (16+2r)^2 halo tile of the
// features and probabilities in LDS and evaluates the (2r+1)^2 - 1 Gaussian taps on the fly:
//     K(i,d)   = sum_k w_k exp(-0.5 |phi_k(i+d) - phi_k(i)|^2),   phi_k = (x, y)/sigma_xy,k  ++  sample/sigma_s,k
//     prod(i,c) = sum_d K(i,d) y(i+d,c)          (saved: the gradient is -2 prod / (N H W), K is symmetric)
//     acc[0] += sum K,  acc[1] += sum_c y(i,c) prod(i,c)
// A neighbour outside the image has phi = 0 and y = 0 (F.unfold's zero padding): it still contributes to sum K.
#include "common.h"

#define FI_CRF_MAXK 4
#define FI_CRF_MAXF 4
#define FI_CRF_MAXC 8
struct CrfArgs {
  const float* y;       // [N][H][W][C]
  const float* feat;    // [N][H][W][F]
  float* prod;          // [N][H][W][C]
  double* acc;          // [FI_CRF_SLOTS][2]
  int N, H, W, C, F, radius, nk;
  float w[FI_CRF_MAXK], inv_sxy[FI_CRF_MAXK], inv_ss[FI_CRF_MAXK];   // inverse sigmas; 0 = modality not in this kernel
};

__global__ __launch_bounds__(256) void gatedcrf_fwd_kernel(CrfArgs a) {
  extern __shared__ float sm[];
  const int R = a.radius, TW = 16 + 2 * R, C = a.C, F = a.F;
  float* sf = sm;                         // [TW*TW][F]   (zero outside the image)
  float* sy = sm + TW * TW * F;           // [TW*TW][C]
  float* sin = sy + TW * TW * C;          // [TW*TW]      1 inside the image, 0 outside
  const int tilesX = (a.W + 15) / 16, tilesY = (a.H + 15) / 16;
  int b = blockIdx.x;
  const int tx = b % tilesX;
  b /= tilesX;
  const int ty = b % tilesY, n = b / tilesY;
  for (int t = threadIdx.x; t < TW * TW; t += 256) {
    const int gy = ty * 16 + t / TW - R, gx = tx * 16 + t % TW - R;
    const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const size_t p = ((size_t)n * a.H + (in ? gy : 0)) * a.W + (in ? gx : 0);
    for (int f = 0; f < F; ++f) sf[t * F + f] = in ? a.feat[p * F + f] : 0.f;
    for (int c = 0; c < C; ++c) sy[t * C + c] = in ? a.y[p * C + c] : 0.f;
    sin[t] = in ? 1.f : 0.f;
  }
  __syncthreads();
  const int ly = threadIdx.x / 16, lx = threadIdx.x % 16;
  const int gy = ty * 16 + ly, gx = tx * 16 + lx;
  const bool live = gy < a.H && gx < a.W;
  const int ct = (ly + R) * TW + lx + R;
  float f0[FI_CRF_MAXF], y0[FI_CRF_MAXC], pr[FI_CRF_MAXC];
  for (int f = 0; f < F; ++f) f0[f] = sf[ct * F + f];
  for (int c = 0; c < C; ++c) {
    y0[c] = sy[ct * C + c];
    pr[c] = 0.f;
  }
  double ksum = 0.0;
  for (int dy = -R; dy <= R; ++dy) {
    float krow = 0.f;
    for (int dx = -R; dx <= R; ++dx) {
      if (dy == 0 && dx == 0) continue;
      const int t = ct + dy * TW + dx;
      float df2 = 0.f;
      for (int f = 0; f < F; ++f) {
        const float dd = sf[t * F + f] - f0[f];
        df2 += dd * dd;
      }
      // the mesh is zero-padded like every other feature: an outside neighbour sits at (0, 0)
      const float in = sin[t];
      const float ddx = in * (float)(gx + dx) - (float)gx, ddy = in * (float)(gy + dy) - (float)gy;
      const float dxy2 = ddx * ddx + ddy * ddy;
      float K = 0.f;
      for (int k = 0; k < a.nk; ++k)
        K += a.w[k] * expf(-0.5f * (dxy2 * a.inv_sxy[k] * a.inv_sxy[k] + df2 * a.inv_ss[k] * a.inv_ss[k]));
      krow += K;
      for (int c = 0; c < C; ++c) pr[c] += K * sy[t * C + c];
    }
    ksum += (double)krow;
  }
  double ysum = 0.0;
  if (live) {
    float* o = a.prod + (((size_t)n * a.H + gy) * a.W + gx) * C;
    for (int c = 0; c < C; ++c) {
      o[c] = pr[c];
      ysum += (double)(y0[c] * pr[c]);
    }
  } else {
    ksum = 0.0;
  }
  __shared__ double red[4][2];
  ksum = wave_sum(ksum);
  ysum = wave_sum(ysum);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6][0] = ksum;
    red[threadIdx.x >> 6][1] = ysum;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const double t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(&a.acc[(blockIdx.x & (FI_CRF_SLOTS - 1)) * 2 + threadIdx.x], t);
  }
}

extern "C" int fi_gatedcrf_fwd(const float* y, const float* feat, int N, int H, int W, int C, int F, int radius, int nk,
                               const float* weights, const float* sigma_xy, const float* sigma_sample, float* prod,
                               double* acc, void* stream) {
  if (!y || !feat || !weights || !sigma_xy || !sigma_sample || !prod || !acc) return FI_ERR_NULL;
  if (N < 1 || H < 1 || W < 1 || C < 1 || C > FI_CRF_MAXC || F < 1 || F > FI_CRF_MAXF || nk < 1 || nk > FI_CRF_MAXK ||
      radius < 1 || radius > 8)
    return FI_ERR_SHAPE;
  CrfArgs a;
  a.y = y, a.feat = feat, a.prod = prod, a.acc = acc;
  a.N = N, a.H = H, a.W = W, a.C = C, a.F = F, a.radius = radius, a.nk = nk;
  for (int k = 0; k < nk; ++k) {
    a.w[k] = weights[k];
    a.inv_sxy[k] = sigma_xy[k] > 0.f ? 1.0f / sigma_xy[k] : 0.f;
    a.inv_ss[k] = sigma_sample[k] > 0.f ? 1.0f / sigma_sample[k] : 0.f;
  }
  const int TW = 16 + 2 * radius;
  const size_t lds = (size_t)TW * TW * (F + C + 1) * sizeof(float);
  const long blocks = (long)N * ((H + 15) / 16) * ((W + 15) / 16);
  hipLaunchKernelGGL(gatedcrf_fwd_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
  FI_CHECK_LAUNCH();
  return 0;
}
