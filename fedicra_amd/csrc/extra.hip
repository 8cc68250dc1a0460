// Module-surface pieces that the reference defines but never runs: the transposed-convolution
// up-sampling branch of UpBlock (/root/reference/code/networks/unet.py:60-62; VNet's UpsamplingDeconvBlock,
// networks/vnet.py:94-118) and GroupNorm (vnet.py:5-31, `normalization='groupnorm'`).
//
// ConvTranspose(kernel 2, stride 2) has no overlap: output pixel (2i+a, 2j+b) is a 1x1 convolution of input pixel (i, j)
// with the (a, b) slice of the filter.  So it runs as ONE 1x1 implicit-GEMM launch with 4*Cout (8*Cout in 3D) output
// channels ordered [a][b][co] -- the existing MFMA kernel -- followed by a depth-to-space shuffle; backward is the inverse
// shuffle and the 1x1 dgrad / wgrad.  fi_depth_to_space2x is that shuffle (and its inverse), 2D or 3D.
//
// GroupNorm: one workgroup per (sample, group); statistics in fp64 over the group's (pixels x channels-per-group) slab,
// then normalise + affine (+ ReLU); backward from the saved mean / inverse std:
//   g = dz * gamma_c (masked by the ReLU);  dx = istd * (g - mean_grp(g) - xhat * mean_grp(g * xhat))
//   dgamma_c += sum_pix dz * xhat,  dbeta_c += sum_pix dz      (atomics over the samples)
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void depth_to_space_kernel(const T* __restrict__ src, T* __restrict__ dst, long npix, int D,
                                                             int H, int W, int C, int three_d, int inverse) {
  // packed: [N][D][H][W][P][C] with P = 4 (a, b) or 8 (c, a, b);  spatial: [N][sD][2H][2W][C], sD = 2D in 3D else D
  const int P = three_d ? 8 : 4;
  const long total = npix * P * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long q = i / C;
    const int p = (int)(q % P);
    q /= P;
    const int x = (int)(q % W);
    q /= W;
    const int y = (int)(q % H);
    q /= H;
    const int z = (int)(q % D);
    const long n = q / D;
    const int b = p & 1, a = (p >> 1) & 1, cz = three_d ? (p >> 2) : 0;
    const int sD = three_d ? 2 * D : D;
    const long o = ((((n * sD + (three_d ? 2 * z + cz : z)) * (2 * H) + (2 * y + a)) * (2L * W)) + (2 * x + b)) * C + c;
    if (inverse)
      dst[i] = src[o];
    else
      dst[o] = src[i];
  }
}

extern "C" int fi_depth_to_space2x(int dtype, const void* src, void* dst, int N, int D, int H, int W, int C, int three_d,
                                   int inverse, void* stream) {
  if (!src || !dst) return FI_ERR_NULL;
  if (N < 1 || D < 1 || H < 1 || W < 1 || C < 1 || (!three_d && D != 1)) return FI_ERR_SHAPE;
  const long npix = (long)N * D * H * W;
  const long total = npix * (three_d ? 8 : 4) * C;
  long blocks = (total + 1023) / 1024;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g((unsigned)blocks), b(256);
  if (dtype == FI_F32)
    hipLaunchKernelGGL(depth_to_space_kernel<float>, g, b, 0, st, (const float*)src, (float*)dst, npix, D, H, W, C, three_d, inverse);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(depth_to_space_kernel<bf16_t>, g, b, 0, st, (const bf16_t*)src, (bf16_t*)dst, npix, D, H, W, C, three_d,
                       inverse);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(depth_to_space_kernel<f16_t>, g, b, 0, st, (const f16_t*)src, (f16_t*)dst, npix, D, H, W, C, three_d,
                       inverse);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------ GroupNorm
__device__ __forceinline__ double block_sum(double v, double* sm) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sm[k];
  return t;
}

template <typename T>
__global__ __launch_bounds__(256) void groupnorm_fwd_kernel(const T* __restrict__ x, T* __restrict__ z,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ mean, float* __restrict__ invstd, long pixels,
                                                            int C, int G, float eps, int relu) {
  __shared__ double sm[4];
  const int n = blockIdx.x / G, g = blockIdx.x % G, cg = C / G;
  const T* xb = x + (size_t)n * pixels * C + g * cg;
  T* zb = z + (size_t)n * pixels * C + g * cg;
  const long cnt = pixels * cg;
  double s1 = 0.0, s2 = 0.0;
  for (long i = threadIdx.x; i < cnt; i += 256) {
    const double v = (double)to_f32(xb[(i / cg) * C + i % cg]);
    s1 += v;
    s2 += v * v;
  }
  s1 = block_sum(s1, sm);
  s2 = block_sum(s2, sm);
  const double m = s1 / (double)cnt;
  double var = s2 / (double)cnt - m * m;
  if (var < 0.0) var = 0.0;
  const float mu = (float)m, istd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    mean[blockIdx.x] = mu;
    invstd[blockIdx.x] = istd;
  }
  for (long i = threadIdx.x; i < cnt; i += 256) {
    const int c = (int)(i % cg);
    const size_t o = (i / cg) * C + c;
    float v = (to_f32(xb[o]) - mu) * istd * gamma[g * cg + c] + beta[g * cg + c];
    if (relu && v < 0.f) v = 0.f;
    zb[o] = from_f32<T>(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ x,
                                                            const T* __restrict__ z, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            T* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, long pixels, int C, int G, int relu) {
  __shared__ double sm[4];
  const int n = blockIdx.x / G, g = blockIdx.x % G, cg = C / G;
  const size_t base = (size_t)n * pixels * C + g * cg;
  const float mu = mean[blockIdx.x], istd = invstd[blockIdx.x];
  const long cnt = pixels * cg;
  double a1 = 0.0, a2 = 0.0;
  for (long i = threadIdx.x; i < cnt; i += 256) {
    const int c = (int)(i % cg);
    const size_t o = base + (i / cg) * C + c;
    float d = to_f32(dz[o]);
    if (relu && !(to_f32(z[o]) > 0.f)) d = 0.f;
    const float xh = (to_f32(x[o]) - mu) * istd;
    const double gd = (double)d * (double)gamma[g * cg + c];
    a1 += gd;
    a2 += gd * (double)xh;
  }
  a1 = block_sum(a1, sm);
  a2 = block_sum(a2, sm);
  const float m1 = (float)(a1 / (double)cnt), m2 = (float)(a2 / (double)cnt);
  // per-channel parameter gradients: thread t < cg owns channel t of this group (cg <= 256: host-checked)
  float pg = 0.f, pb = 0.f;
  const int tc = threadIdx.x;
  for (long i = threadIdx.x; i < cnt; i += 256) {
    const int c = (int)(i % cg);
    const size_t o = base + (i / cg) * C + c;
    float d = to_f32(dz[o]);
    if (relu && !(to_f32(z[o]) > 0.f)) d = 0.f;
    const float xh = (to_f32(x[o]) - mu) * istd;
    dx[o] = from_f32<T>(istd * (d * gamma[g * cg + c] - m1 - xh * m2));
  }
  if (dgamma || dbeta) {
    // channel c = tid % cg, pixel phase tid / cg: 256 / cg pixel streams per channel, folded through LDS
    __shared__ float sg[256], sb[256];
    const int R = 256 / cg, c = tc % cg, pr = tc / cg;
    if (pr < R) {
      for (long p = pr; p < pixels; p += R) {
        const size_t o = base + p * C + c;
        float d = to_f32(dz[o]);
        if (relu && !(to_f32(z[o]) > 0.f)) d = 0.f;
        pg += d * ((to_f32(x[o]) - mu) * istd);
        pb += d;
      }
    }
    sg[tc] = pg, sb[tc] = pb;
    __syncthreads();
    if (tc < cg) {
      for (int r = 1; r < R; ++r) pg += sg[r * cg + tc], pb += sb[r * cg + tc];
      if (dgamma) atomicAdd(&dgamma[g * cg + tc], pg);
      if (dbeta) atomicAdd(&dbeta[g * cg + tc], pb);
    }
  }
}

extern "C" int fi_groupnorm_fwd(int dtype, const void* x, void* z, const float* gamma, const float* beta, float* mean,
                                float* invstd, int N, long pixels, int C, int G, float eps, int relu, void* stream) {
  if (!x || !z || !gamma || !beta || !mean || !invstd) return FI_ERR_NULL;
  if (N < 1 || pixels < 1 || C < 1 || G < 1 || C % G || C / G > 256) return FI_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g((unsigned)(N * G)), b(256);
  if (dtype == FI_F32)
    hipLaunchKernelGGL(groupnorm_fwd_kernel<float>, g, b, 0, st, (const float*)x, (float*)z, gamma, beta, mean, invstd, pixels, C,
                       G, eps, relu);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(groupnorm_fwd_kernel<bf16_t>, g, b, 0, st, (const bf16_t*)x, (bf16_t*)z, gamma, beta, mean, invstd, pixels,
                       C, G, eps, relu);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(groupnorm_fwd_kernel<f16_t>, g, b, 0, st, (const f16_t*)x, (f16_t*)z, gamma, beta, mean, invstd, pixels, C,
                       G, eps, relu);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_groupnorm_bwd(int dtype, const void* dz, const void* x, const void* z, const float* gamma, const float* mean,
                                const float* invstd, void* dx, float* dgamma, float* dbeta, int N, long pixels, int C, int G,
                                int relu, void* stream) {
  if (!dz || !x || !z || !gamma || !mean || !invstd || !dx) return FI_ERR_NULL;
  if (N < 1 || pixels < 1 || C < 1 || G < 1 || C % G || C / G > 256) return FI_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g((unsigned)(N * G)), b(256);
  if (dtype == FI_F32)
    hipLaunchKernelGGL(groupnorm_bwd_kernel<float>, g, b, 0, st, (const float*)dz, (const float*)x, (const float*)z, gamma, mean,
                       invstd, (float*)dx, dgamma, dbeta, pixels, C, G, relu);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(groupnorm_bwd_kernel<bf16_t>, g, b, 0, st, (const bf16_t*)dz, (const bf16_t*)x, (const bf16_t*)z, gamma,
                       mean, invstd, (bf16_t*)dx, dgamma, dbeta, pixels, C, G, relu);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(groupnorm_bwd_kernel<f16_t>, g, b, 0, st, (const f16_t*)dz, (const f16_t*)x, (const f16_t*)z, gamma, mean,
                       invstd, (f16_t*)dx, dgamma, dbeta, pixels, C, G, relu);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------ per-channel statistics
// stats[slot][c][2] += (sum, sum of squares) of x[pix][c] -- the layout fi_bn_finalize / fi_bn_fused_fwd read.  The 2D path
// gets these sums from the convolution epilogue; BatchNorm3d behind the depth-sliced 3D convolution (VNet with
// normalization='batchnorm', networks/vnet.py:16-17) takes them from this pass over the finished volume.
template <typename T>
__global__ __launch_bounds__(256) void channel_stats_kernel(const T* __restrict__ x, double* __restrict__ stats, long pixels,
                                                            int C) {
  __shared__ double s1[256], s2[256];
  const int R = 256 / C, c = threadIdx.x % C, pr = threadIdx.x / C;
  double a = 0.0, b = 0.0;
  if (pr < R) {
    for (long p = (long)blockIdx.x * R + pr; p < pixels; p += (long)gridDim.x * R) {
      const double v = (double)to_f32(x[p * C + c]);
      a += v;
      b += v * v;
    }
  }
  s1[threadIdx.x] = a, s2[threadIdx.x] = b;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    for (int r = 1; r < R; ++r) a += s1[r * C + threadIdx.x], b += s2[r * C + threadIdx.x];
    const int slot = blockIdx.x & (FI_STATS_SLOTS - 1);
    atomicAdd(&stats[((size_t)slot * C + threadIdx.x) * 2 + 0], a);
    atomicAdd(&stats[((size_t)slot * C + threadIdx.x) * 2 + 1], b);
  }
}

extern "C" int fi_channel_stats(int dtype, const void* x, double* stats, long pixels, int C, void* stream) {
  if (!x || !stats) return FI_ERR_NULL;
  if (pixels < 1 || C < 1 || C > 256) return FI_ERR_SHAPE;
  long blocks = (pixels * C + 256 * 64 - 1) / (256 * 64);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g((unsigned)blocks), b(256);
  if (dtype == FI_F32)
    hipLaunchKernelGGL(channel_stats_kernel<float>, g, b, 0, st, (const float*)x, stats, pixels, C);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(channel_stats_kernel<bf16_t>, g, b, 0, st, (const bf16_t*)x, stats, pixels, C);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(channel_stats_kernel<f16_t>, g, b, 0, st, (const f16_t*)x, stats, pixels, C);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}
