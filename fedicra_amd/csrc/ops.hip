// Bandwidth-bound kernels of the FedICRA hot path for gfx950: BatchNorm finalize / apply /
// backward, LeakyReLU + dropout, 2x2 max-pool, bilinear x2 up-sampling, weight repack, layout
// and dtype conversion.  All are HBM-bound: 16-byte vectors per lane over dense NHWC, grid-stride
// loops capped at a few workgroups per CU, per-channel reductions as lane-private partials ->
// LDS -> one fp64 atomic per channel per workgroup.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// small vector helpers: VG elements of T <-> float[VG]
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load_vec(const T* p, float (&f)[DT<T>::VG]) {
  typedef typename DT<T>::vec_t vec_t;
  union {
    vec_t v;
    T e[DT<T>::VG];
  } u;
  u.v = *reinterpret_cast<const vec_t*>(p);
#pragma unroll
  for (int j = 0; j < DT<T>::VG; ++j) f[j] = to_f32(u.e[j]);
}
template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float (&f)[DT<T>::VG]) {
  typedef typename DT<T>::vec_t vec_t;
  union {
    vec_t v;
    T e[DT<T>::VG];
  } u;
#pragma unroll
  for (int j = 0; j < DT<T>::VG; ++j) u.e[j] = from_f32<T>(f[j]);
  *reinterpret_cast<vec_t*>(p) = u.v;
}

// packed 16-byte vector <-> float[VG]: what a kernel requests ahead of its prologue stays PACKED (4 VGPRs) until used
template <typename T>
__device__ __forceinline__ typename DT<T>::vec_t load_raw(const T* p) {
  return *reinterpret_cast<const typename DT<T>::vec_t*>(p);
}
template <typename T>
__device__ __forceinline__ void unpack(const typename DT<T>::vec_t& v, float (&f)[DT<T>::VG]) {
  union {
    typename DT<T>::vec_t v;
    T e[DT<T>::VG];
  } u;
  u.v = v;
#pragma unroll
  for (int j = 0; j < DT<T>::VG; ++j) f[j] = to_f32(u.e[j]);
}

static inline int grid_for(long work_items, int per_block) {
  long b = (work_items + per_block - 1) / per_block;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm finalize
// ------------------------------------------------------------------------------------------------
// running <- (1 - momentum) * running + momentum * batch, products rounded separately: the stand-alone finalize and the
// fused finalize+apply kernel must move the running statistics identically (ops.conv_bn_stats_only relies on it)
__device__ __forceinline__ void bn_running_update(float* rmean, float* rvar, int c, float momentum, float mu, float unb) {
  // products rounded separately, in every inlined copy: the hip __fmul_rn / __fadd_rn helpers are themselves compiled with
  // contraction on (hipcc fused one product into an fma in some kernels and not in others -- an ulp between kernels that
  // must agree), so this is plain arithmetic under an explicit contract(off)
#pragma clang fp contract(off)
  const float keep = 1.f - momentum;
  const float a0 = keep * rmean[c], b0 = momentum * mu;
  rmean[c] = a0 + b0;
  const float a1 = keep * rvar[c], b1 = momentum * unb;
  rvar[c] = a1 + b1;
}

__global__ void bn_finalize_kernel(const double* stats, double count, const float* gamma, const float* beta,
                                   float* rmean, float* rvar, int64_t* nbt, float momentum, float eps, int training,
                                   float* scale, float* shift, float* mean, float* invstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && training && nbt) nbt[0] += 1;
  if (c >= C) return;
  float mu, istd;
  if (training) {
    double s1 = 0.0, s2 = 0.0;
    for (int slot = 0; slot < FI_STATS_SLOTS; ++slot) {   // fixed order: deterministic given the slot contents
      s1 += stats[((size_t)slot * C + c) * 2];
      s2 += stats[((size_t)slot * C + c) * 2 + 1];
    }
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    mu = (float)m;
    istd = (float)(1.0 / sqrt(var + (double)eps));
    const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
    bn_running_update(rmean, rvar, c, momentum, mu, (float)unb);
  } else {
    mu = rmean[c];
    istd = 1.0f / sqrtf(rvar[c] + eps);
  }
  const float sc = gamma[c] * istd;
  scale[c] = sc;
  shift[c] = beta[c] - mu * sc;
  mean[c] = mu;
  invstd[c] = istd;
}

extern "C" int fi_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                              int training, float* scale, float* shift, float* mean, float* invstd, int C,
                              void* stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || !mean || !invstd) return FI_ERR_NULL;
  if (training && !stats) return FI_ERR_NULL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(fi_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, stats, count, gamma,
                     beta, running_mean, running_var, nbt, momentum, eps, training, scale, shift, mean, invstd, C);
  FI_CHECK_LAUNCH();
  return 0;
}

// G groups of one fused launch: coefficients per group, running statistics moved G times in group order
__global__ void bn_finalize_groups_kernel(const double* stats, long gstride, int G, double count, const float* gamma,
                                          const float* beta, float* rmean, float* rvar, int64_t* nbt, float momentum,
                                          float eps, float* coef, int C) {
  // rmean == NULL: coefficients only; coef == NULL: running statistics (and the counter) only -- the two halves of one call, with
  // the arithmetic of the whole: a caller that must ORDER the running-statistics update behind another stream's work makes it
  // later, on that stream, and still has the coefficients at once (ops.probe_conv_bn, flower_pCE_2D._iteration)
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) nbt[0] += G;
  if (c >= C) return;
  const float ga = gamma[c], be = beta[c];
  for (int g = 0; g < G; ++g) {
    const double* st = stats + (size_t)g * gstride;
    double s1 = 0.0, s2 = 0.0;
    for (int slot = 0; slot < FI_STATS_SLOTS; ++slot) {
      s1 += st[((size_t)slot * C + c) * 2];
      s2 += st[((size_t)slot * C + c) * 2 + 1];
    }
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const float mu = (float)m;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
    if (rmean) bn_running_update(rmean, rvar, c, momentum, mu, (float)unb);
    if (coef) {
      const float sc = ga * istd;
      coef[(size_t)g * C + c] = sc;
      coef[(size_t)(G + g) * C + c] = be - mu * sc;
    }
  }
}

extern "C" int fi_bn_finalize_groups(const double* stats, long stats_group_stride, int groups, double count,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     int64_t* nbt, float momentum, float eps, float* coef, int C, void* stream) {
  if (!stats || !gamma || !beta || (!running_mean != !running_var) || (!running_mean && !coef)) return FI_ERR_NULL;
  if (!running_mean) nbt = nullptr;                        // coefficients only: nothing of the module's state moves
  if (groups < 1 || C < 1) return FI_ERR_SHAPE;
  hipLaunchKernelGGL(bn_finalize_groups_kernel, dim3(fi_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, stats,
                     stats_group_stride, groups, count, gamma, beta, running_mean, running_var, nbt, momentum, eps, coef, C);
  FI_CHECK_LAUNCH();
  return 0;
}

// The running-statistics half of fi_bn_finalize_groups for SEVERAL BatchNorm layers in one launch: what the batched LC
// forwards leave to the training stream (ops._probe_finalize: 19 layers per iteration) would otherwise be 19 dependent launches
// of a few microseconds each right before the optimizer step.  Items travel by value in the kernel argument (no device table
// to build or keep alive under graph capture); blockIdx.y = item, the arithmetic and its order are bn_finalize_groups_kernel's.
struct FiBnRunPack {
  FiBnRunItem it[FI_BN_RUN_MAX];
};

__global__ void bn_running_groups_multi_kernel(FiBnRunPack p) {
  const FiBnRunItem& q = p.it[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && q.num_batches_tracked) q.num_batches_tracked[0] += q.groups;
  if (c >= q.C) return;
  for (int g = 0; g < q.groups; ++g) {
    const double* st = q.stats + (size_t)g * q.stats_group_stride;
    double s1 = 0.0, s2 = 0.0;
    for (int slot = 0; slot < FI_STATS_SLOTS; ++slot) {
      s1 += st[((size_t)slot * q.C + c) * 2];
      s2 += st[((size_t)slot * q.C + c) * 2 + 1];
    }
    const double m = s1 / q.count;
    double var = s2 / q.count - m * m;
    if (var < 0.0) var = 0.0;
    const double unb = q.count > 1.0 ? var * (q.count / (q.count - 1.0)) : var;
    bn_running_update(q.running_mean, q.running_var, c, q.momentum, (float)m, (float)unb);
  }
}

extern "C" int fi_bn_running_groups_multi(const FiBnRunItem* items, int n, void* stream) {
  if (!items) return FI_ERR_NULL;
  if (n < 0) return FI_ERR_SHAPE;
  for (int base = 0; base < n; base += FI_BN_RUN_MAX) {
    FiBnRunPack p;
    const int m = n - base < FI_BN_RUN_MAX ? n - base : FI_BN_RUN_MAX;
    int cmax = 1;
    for (int i = 0; i < m; ++i) {
      p.it[i] = items[base + i];
      if (!p.it[i].stats || !p.it[i].running_mean || !p.it[i].running_var) return FI_ERR_NULL;
      if (p.it[i].groups < 1 || p.it[i].C < 1) return FI_ERR_SHAPE;
      if (p.it[i].C > cmax) cmax = p.it[i].C;
    }
    hipLaunchKernelGGL(bn_running_groups_multi_kernel, dim3(fi_cdiv(cmax, 64), m), dim3(64), 0, (hipStream_t)stream, p);
    FI_CHECK_LAUNCH();
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// BN apply + activation + dropout
// ------------------------------------------------------------------------------------------------
struct DropSpec {
  int mode;
  uint32_t thresh;  // drop when rand32 < thresh
  float keep_scale;
  uint64_t seed;
  const uint8_t* mask;
  const int32_t* seed_offset;
  int C, hw;
};

__device__ __forceinline__ uint64_t drop_seed(const DropSpec& d) {
  uint64_t s = d.seed;
  if (d.seed_offset) s += 0xD1B54A32D192ED03ull * (uint64_t)(uint32_t)d.seed_offset[0];
  return s;
}

// `seed` is the value of drop_seed(d), hoisted out of the element loop by the callers
__device__ __forceinline__ float drop_factor(const DropSpec& d, uint64_t seed, size_t pixel, int c) {
  switch (d.mode) {
    case FI_DROP_MASK_ELEM:
      return d.mask[pixel * d.C + c] ? d.keep_scale : 0.f;
    case FI_DROP_RNG_ELEM: {
      const size_t e = pixel * d.C + c;
      uint32_t r[4];
      fi_rand32x4(seed, e >> 2, r);
      return r[e & 3] >= d.thresh ? d.keep_scale : 0.f;
    }
    case FI_DROP_MASK_CHAN:
      return d.mask[(pixel / d.hw) * d.C + c] ? d.keep_scale : 0.f;
    case FI_DROP_RNG_CHAN:
      return fi_keep(seed, (pixel / d.hw) * d.C + c, d.thresh) ? d.keep_scale : 0.f;
    default:
      return 1.f;
  }
}

// The VG factors of vector `vec` (= flat element index / VG of a dense NHWC tensor; pixel / c0 = its pixel and first
// channel).  The mode switch is taken once per vector; element-wise RNG draws come four at a time.
template <int VG>
__device__ __forceinline__ void drop_factors(const DropSpec& d, uint64_t seed, size_t vec, size_t pixel, int c0,
                                             float (&f)[VG]) {
  if (d.mode == FI_DROP_RNG_ELEM) {
#pragma unroll
    for (int g = 0; g < VG / 4; ++g) {
      uint32_t r[4];
      fi_rand32x4(seed, vec * (VG / 4) + g, r);
#pragma unroll
      for (int j = 0; j < 4; ++j) f[g * 4 + j] = r[j] >= d.thresh ? d.keep_scale : 0.f;
    }
  } else if (d.mode == FI_DROP_NONE) {
#pragma unroll
    for (int j = 0; j < VG; ++j) f[j] = 1.f;
  } else {
#pragma unroll
    for (int j = 0; j < VG; ++j) f[j] = drop_factor(d, seed, pixel, c0 + j);
  }
}

// Several SAMPLES of one launch (blockIdx.y): InstanceNorm3d is these kernels per sample -- its own statistics, coefficient rows and
// sums -- and was one launch per sample and pass; the small deep levels of the 3D U-Net are a few microseconds of work under a
// launch floor each.  n <= 1: one tensor, the pointers as they are.
struct BnBatch {
  int n;
  long tensor, stats, coef;        // strides between samples: elements of the activation, doubles, floats per coefficient ROW set
};

static DropSpec make_drop(const FiBnAct* d) {
  DropSpec s;
  s.mode = d->drop_p > 0.f ? d->drop_mode : FI_DROP_NONE;
  double t = (double)d->drop_p * 4294967296.0;
  s.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  s.keep_scale = d->drop_p < 1.f ? 1.0f / (float)(1.0 - (double)d->drop_p) : 0.f;
  s.seed = d->seed;
  s.mask = d->mask;
  s.seed_offset = d->seed_offset;
  s.C = d->C;
  s.hw = d->hw;
  return s;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, T* __restrict__ z,
                                                         long nvec, int C, float slope, DropSpec dr) {
  constexpr int VG = DT<T>::VG;
  // CV = C/VG divides the 256-thread block (host-checked): a thread keeps the SAME channel vector for its
  // whole grid-stride walk, so the per-channel coefficients are loaded once into registers and the loop
  // body is pure streaming (no per-element coefficient loads, no integer division).
  const unsigned CV = C / VG;
  const int c0 = (int)(threadIdx.x % CV) * VG;
  float sc[VG], sh[VG];
#pragma unroll
  for (int j = 0; j < VG; ++j) {
    sc[j] = scale[c0 + j];
    sh[j] = shift[c0 + j];
  }
  const uint64_t seed = drop_seed(dr);
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x, istep = gridDim.x * 256u;
  unsigned pixel = i0 / CV;
  const unsigned pstep = istep / CV;
  // MLP: 4 independent 16-byte loads per thread are issued before any is consumed (the plain grid-stride
  // loop exposed one HBM round trip per vector)
  for (long i = i0; i < nvec; i += 4L * istep, pixel += 4 * pstep) {
    float f[4][VG];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long iu = i + (long)u * istep;
      load_vec<T>(y + (iu < nvec ? iu : i) * VG, f[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long iu = i + (long)u * istep;
      if (iu < nvec) {
        float df[VG];
        drop_factors<VG>(dr, seed, (size_t)iu, pixel + u * pstep, c0, df);
#pragma unroll
        for (int j = 0; j < VG; ++j) {
          float v = f[u][j] * sc[j] + sh[j];
          v = v > 0.f ? v : v * slope;
          v *= df[j];
          f[u][j] = v;
        }
        store_vec<T>(z + iu * VG, f[u]);
      }
    }
  }
}

extern "C" int fi_bn_act_fwd(const FiBnAct* d, const void* y, const float* scale, const float* shift, void* z,
                             void* stream) {
  if (!d || !y || !scale || !shift || !z) return FI_ERR_NULL;
  if (d->dtype != FI_F32 && d->dtype != FI_BF16 && d->dtype != FI_F16) return FI_ERR_DTYPE;
  {
    const int vg = d->dtype == FI_F32 ? 4 : 8;
    if (d->C % vg || d->C / vg > 256 || 256 % (d->C / vg)) return FI_ERR_SHAPE;   // CV must divide the block
  }
  const DropSpec dr = make_drop(d);
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == FI_F32) {
    if (d->C % 4) return FI_ERR_SHAPE;
    const long nvec = d->pixels * (d->C / 4);
    if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
    hipLaunchKernelGGL(bn_act_fwd_kernel<float>, dim3(grid_for(nvec, 256 * 4)), dim3(256), 0, st, (const float*)y,
                       scale, shift, (float*)z, nvec, d->C, d->slope, dr);
  } else if (d->dtype == FI_BF16) {
    if (d->C % 8) return FI_ERR_SHAPE;
    const long nvec = d->pixels * (d->C / 8);
    if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
    hipLaunchKernelGGL(bn_act_fwd_kernel<bf16_t>, dim3(grid_for(nvec, 256 * 4)), dim3(256), 0, st, (const bf16_t*)y,
                       scale, shift, (bf16_t*)z, nvec, d->C, d->slope, dr);
  } else if (d->dtype == FI_F16) {
    if (d->C % 8) return FI_ERR_SHAPE;
    const long nvec = d->pixels * (d->C / 8);
    if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
    hipLaunchKernelGGL(bn_act_fwd_kernel<f16_t>, dim3(grid_for(nvec, 256 * 4)), dim3(256), 0, st, (const f16_t*)y,
                       scale, shift, (f16_t*)z, nvec, d->C, d->slope, dr);
  } else {
    return FI_ERR_DTYPE;
  }
  FI_CHECK_LAUNCH();
  return 0;
}

// g = dz through dropout and the activation, evaluated from the saved conv output y
template <typename T>
__device__ __forceinline__ void act_grad(const float (&dzv)[DT<T>::VG], const float (&yv)[DT<T>::VG],
                                         const float (&scale)[DT<T>::VG], const float (&shift)[DT<T>::VG], int c0,
                                         size_t pixel, size_t vec, float slope, const DropSpec& dr, uint64_t seed,
                                         float (&g)[DT<T>::VG]) {
  float df[DT<T>::VG];
  drop_factors<DT<T>::VG>(dr, seed, vec, pixel, c0, df);
#pragma unroll
  for (int j = 0; j < DT<T>::VG; ++j) {
    const float v = yv[j] * scale[j] + shift[j];
    float gg = dzv[j];
    gg *= df[j];
    g[j] = v > 0.f ? gg : gg * slope;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const T* __restrict__ dz, const T* __restrict__ y,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, double* sums,
                                                                long pixels, int C, float slope, DropSpec dr, BnBatch bb) {
  constexpr int VG = DT<T>::VG;
  if (bb.n > 1) {
    const long b = blockIdx.y;
    dz += b * bb.tensor, y += b * bb.tensor, sums += b * bb.stats;
    scale += b * bb.coef, shift += b * bb.coef, mean += b * bb.coef, invstd += b * bb.coef;
  }
  const int CV = C / VG;          // channel vectors per pixel; divides 256
  const int PS = 256 / CV;        // pixels handled concurrently by one workgroup
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  const int c0 = cv * VG;
  const uint64_t seed = drop_seed(dr);
  float sg[VG], sgx[VG], sc[VG], sh[VG], mu[VG], is[VG];
#pragma unroll
  for (int j = 0; j < VG; ++j) {
    sg[j] = sgx[j] = 0.f;
    sc[j] = scale[c0 + j];
    sh[j] = shift[c0 + j];
    mu[j] = mean[c0 + j];
    is[j] = invstd[c0 + j];
  }
  const long pstride = (long)gridDim.x * PS;
  for (long p = (long)blockIdx.x * PS + pl; p < pixels; p += 2 * pstride) {   // 2 pixels x (dz, y) in flight
    float dzv[2][VG], yv[2][VG];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long pu = p + u * pstride;
      const long pq = pu < pixels ? pu : p;
      load_vec<T>(dz + (pq * CV + cv) * VG, dzv[u]);
      load_vec<T>(y + (pq * CV + cv) * VG, yv[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long pu = p + u * pstride;
      if (pu < pixels) {
        float g[VG];
        act_grad<T>(dzv[u], yv[u], sc, sh, c0, (size_t)pu, (size_t)pu * CV + cv, slope, dr, seed, g);
#pragma unroll
        for (int j = 0; j < VG; ++j) {
          sg[j] += g[j];
          sgx[j] += g[j] * (yv[u][j] - mu[j]) * is[j];
        }
      }
    }
  }
  __shared__ float red[256][2 * VG + 1];
#pragma unroll
  for (int j = 0; j < VG; ++j) {
    red[threadIdx.x][j] = sg[j];
    red[threadIdx.x][VG + j] = sgx[j];
  }
  __syncthreads();
  if ((int)threadIdx.x < C * 2) {
    const int c = threadIdx.x >> 1, which = threadIdx.x & 1;
    const int tcv = c / VG, j = c % VG;
    double tot = 0.0;
    for (int q = 0; q < PS; ++q) tot += (double)red[q * CV + tcv][which * VG + j];
    atomicAdd(&sums[((size_t)(blockIdx.x & (FI_STATS_SLOTS - 1)) * C + c) * 2 + which], tot);   // slot: see conv epilogue
  }
  // C*2 can exceed 256 (C up to 512): remaining channels in further strides
  for (int t = threadIdx.x + 256; t < C * 2; t += 256) {
    const int c = t >> 1, which = t & 1;
    const int tcv = c / VG, j = c % VG;
    double tot = 0.0;
    for (int q = 0; q < PS; ++q) tot += (double)red[q * CV + tcv][which * VG + j];
    atomicAdd(&sums[((size_t)(blockIdx.x & (FI_STATS_SLOTS - 1)) * C + c) * 2 + which], tot);
  }
}

static int bn_act_bwd_reduce_impl(const FiBnAct* d, const void* dz, const void* y, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, double* sums, BnBatch bb, void* stream) {
  if (!d || !dz || !y || !scale || !shift || !mean || !invstd || !sums) return FI_ERR_NULL;
  const DropSpec dr = make_drop(d);
  hipStream_t st = (hipStream_t)stream;
  const int VG = d->dtype == FI_F32 ? 4 : 8;
  if (d->dtype != FI_F32 && d->dtype != FI_BF16 && d->dtype != FI_F16) return FI_ERR_DTYPE;
  if (d->C % VG) return FI_ERR_SHAPE;
  const int CV = d->C / VG;
  if (CV > 256 || 256 % CV) return FI_ERR_SHAPE;
  const int PS = 256 / CV;
  const int grid = grid_for(d->pixels, PS * 4);
  if (d->dtype == FI_F32)
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<float>, dim3(grid, bb.n), dim3(256), 0, st, (const float*)dz,
                       (const float*)y, scale, shift, mean, invstd, sums, d->pixels, d->C, d->slope, dr, bb);
  else if (d->dtype == FI_F16)
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<f16_t>, dim3(grid, bb.n), dim3(256), 0, st, (const f16_t*)dz,
                       (const f16_t*)y, scale, shift, mean, invstd, sums, d->pixels, d->C, d->slope, dr, bb);
  else
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<bf16_t>, dim3(grid, bb.n), dim3(256), 0, st, (const bf16_t*)dz,
                       (const bf16_t*)y, scale, shift, mean, invstd, sums, d->pixels, d->C, d->slope, dr, bb);
  FI_CHECK_LAUNCH();
  return 0;
}
extern "C" int fi_bn_act_bwd_reduce(const FiBnAct* d, const void* dz, const void* y, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, double* sums,
                                    void* stream) {
  return bn_act_bwd_reduce_impl(d, dz, y, scale, shift, mean, invstd, sums, BnBatch{1, 0, 0, 0}, stream);
}
static int bn_batch_ok(const FiBnAct* d, int nbatch, long tensor_stride, long stats_stride, long coef_stride) {
  if (!d) return FI_ERR_NULL;
  if (nbatch < 1 || nbatch > 65535 || tensor_stride < d->pixels * d->C || stats_stride < 0 || coef_stride < d->C) return FI_ERR_SHAPE;
  if (d->drop_p > 0.f) return FI_ERR_UNSUPPORTED;              // (mask indices / seeds are per tensor: not batched)
  return 0;
}
extern "C" int fi_bn_act_bwd_reduce_batched(const FiBnAct* d, int nbatch, long tensor_stride, long stats_stride, long coef_stride,
                                            const void* dz, const void* y, const float* scale, const float* shift,
                                            const float* mean, const float* invstd, double* sums, void* stream) {
  const int rc = bn_batch_ok(d, nbatch, tensor_stride, stats_stride, coef_stride);
  if (rc) return rc;
  return bn_act_bwd_reduce_impl(d, dz, y, scale, shift, mean, invstd, sums, BnBatch{nbatch, tensor_stride, stats_stride, coef_stride}, stream);
}

template <typename T, bool HOIST>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ y,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const double* __restrict__ sums, int training,
                                                               T* __restrict__ dy, float* dgamma, float* dbeta,
                                                               int accumulate_param, long nvec, long pixels, int C,
                                                               float slope, DropSpec dr, BnBatch bb) {
  constexpr int VG = DT<T>::VG;
  typedef typename DT<T>::vec_t vec_t;
  if (bb.n > 1) {
    const long b = blockIdx.y;
    dz += b * bb.tensor, y += b * bb.tensor, sums += b * bb.stats;
    if (dy) dy += b * bb.tensor;
    scale += b * bb.coef, shift += b * bb.coef, mean += b * bb.coef, invstd += b * bb.coef;
  }
  const int CV = C / VG;
  const unsigned CVu = CV;
  const int c0 = (int)(threadIdx.x % CVu) * VG;
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x, istep = gridDim.x * 256u;
  // HOIST (small maps, where the kernel is one dependent chain): three independent round trips -- the first batch of
  // (dz, y), the per-channel coefficients, the partial sums -- are all requested here, ahead of the barrier of the
  // fold; issued one after the other they put a floor of 7.6 us under this kernel (now 5.9-6.5).  On the large maps
  // the extra live registers cost occupancy instead (17.9 -> 19.7 us at 256^2), so those keep the plain order.
  vec_t r0dz[2], r0y[2];
  if (HOIST && dy) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long iu = (long)i0 + (long)u * istep;
      const long il = iu < nvec ? iu : (i0 < nvec ? (long)i0 : 0L);
      r0dz[u] = load_raw<T>(dz + il * VG);
      r0y[u] = load_raw<T>(y + il * VG);
    }
  }
  float sc[VG], sh[VG], mu[VG], is[VG];
#pragma unroll
  for (int j = 0; j < VG; ++j) {
    sc[j] = scale[c0 + j];
    sh[j] = shift[c0 + j];
    mu[j] = mean[c0 + j];
    is[j] = invstd[c0 + j];
  }
  // sums arrives as FI_STATS_SLOTS partial accumulators: fold them once per workgroup into LDS
  __shared__ float ssum[2 * 512];
  for (int t = threadIdx.x; t < 2 * C; t += blockDim.x) {
    double tot = 0.0;
#pragma unroll
    for (int slot = 0; slot < FI_STATS_SLOTS; ++slot) tot += sums[(size_t)slot * 2 * C + t];
    ssum[t] = (float)tot;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float dg = ssum[2 * c + 1], db = ssum[2 * c];
      if (dgamma) dgamma[c] = accumulate_param ? dgamma[c] + dg : dg;
      if (dbeta) dbeta[c] = accumulate_param ? dbeta[c] + db : db;
    }
  }
  if (!dy) return;
  const uint64_t seed = drop_seed(dr);
  const float invM = (float)(1.0 / (double)pixels);
  // per-thread channel constants (CV divides 256, see bn_act_fwd_kernel):
  //   dy = sc*g - k0 - (y - mu)*k1   with k0 = sc*sum_g/M, k1 = sc*invstd*sum_gx/M   (training)
  float k0[VG], k1[VG];
#pragma unroll
  for (int j = 0; j < VG; ++j) {
    const int c = c0 + j;
    k0[j] = training ? sc[j] * (ssum[2 * c] * invM) : 0.f;
    k1[j] = training ? sc[j] * is[j] * (ssum[2 * c + 1] * invM) : 0.f;
  }
  unsigned pixel = i0 / CVu;
  const unsigned pstep = istep / CVu;
  for (long i = i0; i < nvec; i += 2L * istep, pixel += 2 * pstep) {   // 2 x (dz, y) = 4 loads in flight
    float dzv[2][VG], yv[2][VG];
    if (HOIST && i == (long)i0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        unpack<T>(r0dz[u], dzv[u]);
        unpack<T>(r0y[u], yv[u]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long iu = i + (long)u * istep;
        const long il = iu < nvec ? iu : i;
        load_vec<T>(dz + il * VG, dzv[u]);
        load_vec<T>(y + il * VG, yv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long iu = i + (long)u * istep;
      if (iu < nvec) {
        float g[VG], out[VG];
        act_grad<T>(dzv[u], yv[u], sc, sh, c0, pixel + u * pstep, (size_t)iu, slope, dr, seed, g);
#pragma unroll
        for (int j = 0; j < VG; ++j) out[j] = sc[j] * g[j] - k0[j] - (yv[u][j] - mu[j]) * k1[j];
        store_vec<T>(dy + iu * VG, out);
      }
    }
  }
}

static int bn_act_bwd_apply_impl(const FiBnAct* d, const void* dz, const void* y, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, const double* sums, int training, void* dy, float* dgamma,
                                 float* dbeta, int accumulate_param, BnBatch bb, void* stream) {
  if (!d || !dz || !y || !scale || !shift || !mean || !invstd || !sums) return FI_ERR_NULL;
  if (d->dtype != FI_F32 && d->dtype != FI_BF16 && d->dtype != FI_F16) return FI_ERR_DTYPE;
  {
    const int vg = d->dtype == FI_F32 ? 4 : 8;
    if (d->C % vg || d->C / vg > 256 || 256 % (d->C / vg) || d->C > 512) return FI_ERR_SHAPE;   // CV divides the block; LDS fold holds 512 channels
  }
  const DropSpec dr = make_drop(d);
  hipStream_t st = (hipStream_t)stream;
  const int vgl = d->dtype == FI_F32 ? 4 : 8;
  const long nvec = d->pixels * (d->C / vgl);
  if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
  const bool hoist = nvec <= 512L * 1024;             // the 64^2 level and below of the U-Net (measured crossover)
  const dim3 g(grid_for(nvec, 256 * 4), bb.n), b(256);
#define FI_APPLY(T_, H_)                                                                                            \
  hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T_, H_>), g, b, 0, st, (const T_*)dz, (const T_*)y, scale, shift, mean, \
                     invstd, sums, training, (T_*)dy, dgamma, dbeta, accumulate_param, nvec, d->pixels, d->C,       \
                     d->slope, dr, bb)
  if (d->dtype == FI_F32) {
    if (hoist) FI_APPLY(float, true); else FI_APPLY(float, false);
  } else if (d->dtype == FI_F16) {
    if (hoist) FI_APPLY(f16_t, true); else FI_APPLY(f16_t, false);
  } else {
    if (hoist) FI_APPLY(bf16_t, true); else FI_APPLY(bf16_t, false);
  }
#undef FI_APPLY
  FI_CHECK_LAUNCH();
  return 0;
}
extern "C" int fi_bn_act_bwd_apply(const FiBnAct* d, const void* dz, const void* y, const float* scale,
                                   const float* shift, const float* mean, const float* invstd, const double* sums,
                                   int training, void* dy, float* dgamma, float* dbeta, int accumulate_param,
                                   void* stream) {
  return bn_act_bwd_apply_impl(d, dz, y, scale, shift, mean, invstd, sums, training, dy, dgamma, dbeta, accumulate_param,
                               BnBatch{1, 0, 0, 0}, stream);
}
extern "C" int fi_bn_act_bwd_apply_batched(const FiBnAct* d, int nbatch, long tensor_stride, long stats_stride, long coef_stride,
                                           const void* dz, const void* y, const float* scale, const float* shift,
                                           const float* mean, const float* invstd, const double* sums, int training, void* dy,
                                           void* stream) {
  const int rc = bn_batch_ok(d, nbatch, tensor_stride, stats_stride, coef_stride);
  if (rc) return rc;
  return bn_act_bwd_apply_impl(d, dz, y, scale, shift, mean, invstd, sums, training, dy, nullptr, nullptr, 0,
                               BnBatch{nbatch, tensor_stride, stats_stride, coef_stride}, stream);
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(2)
// ------------------------------------------------------------------------------------------------
// z = maxpool2x2(act(scale_g[c] * y + shift_g[c])) of a RAW convolution output y [N][2H][2W][C] with one coefficient row pair
// per statistics group (g = n / gimages): what the pooling loader of the fused forward evaluates (conv_impl.h XF == 2: each of the
// four values rounded to the storage type first, then the scan-order strict maximum), written out ONCE.  The deep DownBlocks of
// the batched LC forwards stage every input tile once per 64 / 128-channel output slab; with the pooled activation in memory
// they take the wave-specialised kernels' plain loader instead of the one-tile kernel's four-vectors-per-staged-one.
// (POOL = false: the same without the pooling -- act(BN(y)) of every group in ONE launch: the frozen encoder's feature maps of
// a whole ALA epoch, fi_bn_act_fwd's arithmetic)
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void bn_act_pool_groups_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, float slope, T* __restrict__ z,
                                                                 int N, int Ho, int Wo, int C, int gimages) {
  constexpr int VG = DT<T>::VG;
  const int CV = C / VG, W = 2 * Wo;
  const long nvec = (long)N * Ho * Wo * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)((unsigned)i % (unsigned)CV);
    unsigned p = (unsigned)i / (unsigned)CV;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const int g = gimages > 0 ? n / gimages : 0;
    const float* sc = scale + (size_t)g * C + cv * VG;
    const float* sh = shift + (size_t)g * C + cv * VG;
    float m[VG];
    if constexpr (POOL) {
      const T* base = y + ((((size_t)n * 2 * Ho + 2 * oy) * W + 2 * ox) * C + cv * VG);
      float q[4][VG];
      load_vec<T>(base, q[0]);
      load_vec<T>(base + C, q[1]);
      load_vec<T>(base + (size_t)W * C, q[2]);
      load_vec<T>(base + (size_t)W * C + C, q[3]);
#pragma unroll
      for (int j = 0; j < VG; ++j) {
        float best = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float t = q[k][j] * sc[j] + sh[j];
          const float v = to_f32(from_f32<T>(fmaxf(t, t * slope)));      // rounded like fi_bn_act_fwd, THEN compared
          best = (k == 0 || v > best) ? v : best;
        }
        m[j] = best;
      }
    } else {
      float q[VG];
      load_vec<T>(y + i * VG, q);
#pragma unroll
      for (int j = 0; j < VG; ++j) {
        const float t = q[j] * sc[j] + sh[j];
        m[j] = fmaxf(t, t * slope);
      }
    }
    store_vec<T>(z + i * VG, m);
  }
}

extern "C" int fi_bn_act_pool_groups(int dtype, const void* y, const float* scale, const float* shift, float slope, void* z, int N,
                                     int Ho, int Wo, int C, int group_images, int pool, void* stream) {
  if (!y || !scale || !shift || !z) return FI_ERR_NULL;
  if (N < 1 || Ho < 1 || Wo < 1 || C < 1 || group_images < 0 || (group_images > 0 && N % group_images)) return FI_ERR_SHAPE;
  if (slope < 0.f || slope > 1.f) return FI_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int vg = dtype == FI_F32 ? 4 : 8;
  if (C % vg) return FI_ERR_SHAPE;
  const long nvec = (long)N * Ho * Wo * (C / vg);
  if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;
  const dim3 grid(grid_for(nvec, 256 * 2)), blk(256);
#define FI_BAP(TT, PP) hipLaunchKernelGGL((bn_act_pool_groups_kernel<TT, PP>), grid, blk, 0, st, (const TT*)y, scale, shift, slope, (TT*)z, N, Ho, Wo, C, group_images)
  if (dtype == FI_F32) {
    if (pool) FI_BAP(float, true); else FI_BAP(float, false);
  } else if (dtype == FI_BF16) {
    if (pool) FI_BAP(bf16_t, true); else FI_BAP(bf16_t, false);
  } else if (dtype == FI_F16) {
    if (pool) FI_BAP(f16_t, true); else FI_BAP(f16_t, false);
  } else {
    return FI_ERR_DTYPE;
  }
#undef FI_BAP
  FI_CHECK_LAUNCH();
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H,
                                                          int W, int C) {
  constexpr int VG = DT<T>::VG;
  const int CV = C / VG, Ho = H / 2, Wo = W / 2;
  const long nvec = (long)N * Ho * Wo * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)((unsigned)i % (unsigned)CV);
    unsigned p = (unsigned)i / (unsigned)CV;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const T* base = x + ((((size_t)n * H + 2 * oy) * W + 2 * ox) * C + cv * VG);
    float a[VG], b[VG], c[VG], d[VG], m[VG];
    load_vec<T>(base, a);
    load_vec<T>(base + C, b);
    load_vec<T>(base + (size_t)W * C, c);
    load_vec<T>(base + (size_t)W * C + C, d);
#pragma unroll
    for (int j = 0; j < VG; ++j) {
      float mm = a[j];
      if (b[j] > mm) mm = b[j];
      if (c[j] > mm) mm = c[j];
      if (d[j] > mm) mm = d[j];
      m[j] = mm;
    }
    store_vec<T>(y + i * VG, m);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const T* add, T* dx, int N, int H, int W, int C) {
  constexpr int VG = DT<T>::VG;
  const int CV = C / VG, Ho = H / 2, Wo = W / 2;
  const long nvec = (long)N * Ho * Wo * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)((unsigned)i % (unsigned)CV);
    unsigned p = (unsigned)i / (unsigned)CV;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const size_t o00 = (((size_t)n * H + 2 * oy) * W + 2 * ox) * C + cv * VG;
    const size_t offs[4] = {o00, o00 + C, o00 + (size_t)W * C, o00 + (size_t)W * C + C};
    float v[4][VG], g[VG], out[4][VG];
#pragma unroll
    for (int q = 0; q < 4; ++q) load_vec<T>(x + offs[q], v[q]);
    load_vec<T>(dy + i * VG, g);
#pragma unroll
    for (int j = 0; j < VG; ++j) {
      int best = 0;
      float mm = v[0][j];
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (v[q][j] > mm) {
          mm = v[q][j];
          best = q;
        }
#pragma unroll
      for (int q = 0; q < 4; ++q) out[q][j] = (q == best) ? g[j] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (add) {   // dx = add + scatter(dy): the other consumer's gradient (skip connection) joins in the same pass
        float old[VG];
        load_vec<T>(add + offs[q], old);
#pragma unroll
        for (int j = 0; j < VG; ++j) out[q][j] += old[j];
      }
      store_vec<T>(dx + offs[q], out[q]);
    }
  }
}

extern "C" int fi_maxpool2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream) {
  if (!x || !y) return FI_ERR_NULL;
  if ((H & 1) || (W & 1)) return FI_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FI_F32) {
    if (C % 4) return FI_ERR_SHAPE;
    const long nvec = (long)N * (H / 2) * (W / 2) * (C / 4);
    if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(nvec, 256 * 2)), dim3(256), 0, st, (const float*)x,
                       (float*)y, N, H, W, C);
  } else if (dtype == FI_BF16) {
    if (C % 8) return FI_ERR_SHAPE;
    const long nvec = (long)N * (H / 2) * (W / 2) * (C / 8);
    if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid_for(nvec, 256 * 2)), dim3(256), 0, st, (const bf16_t*)x,
                       (bf16_t*)y, N, H, W, C);
  } else if (dtype == FI_F16) {
    if (C % 8) return FI_ERR_SHAPE;
    const long nvec = (long)N * (H / 2) * (W / 2) * (C / 8);
    if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
    hipLaunchKernelGGL(maxpool_fwd_kernel<f16_t>, dim3(grid_for(nvec, 256 * 2)), dim3(256), 0, st, (const f16_t*)x,
                       (f16_t*)y, N, H, W, C);
  } else {
    return FI_ERR_DTYPE;
  }
  FI_CHECK_LAUNCH();
  return 0;
}

static int maxpool_bwd_impl(int dtype, const void* x, const void* dy, const void* add, void* dx, int N, int H, int W,
                            int C, void* stream) {
  if (!x || !dy || !dx) return FI_ERR_NULL;
  if ((H & 1) || (W & 1)) return FI_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FI_F32) {
    if (C % 4) return FI_ERR_SHAPE;
    const long nvec = (long)N * (H / 2) * (W / 2) * (C / 4);
    if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for(nvec, 256 * 2)), dim3(256), 0, st, (const float*)x,
                       (const float*)dy, (const float*)add, (float*)dx, N, H, W, C);
  } else if (dtype == FI_BF16) {
    if (C % 8) return FI_ERR_SHAPE;
    const long nvec = (long)N * (H / 2) * (W / 2) * (C / 8);
    if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid_for(nvec, 256 * 2)), dim3(256), 0, st,
                       (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)add, (bf16_t*)dx, N, H, W, C);
  } else if (dtype == FI_F16) {
    if (C % 8) return FI_ERR_SHAPE;
    const long nvec = (long)N * (H / 2) * (W / 2) * (C / 8);
    if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
    hipLaunchKernelGGL(maxpool_bwd_kernel<f16_t>, dim3(grid_for(nvec, 256 * 2)), dim3(256), 0, st,
                       (const f16_t*)x, (const f16_t*)dy, (const f16_t*)add, (f16_t*)dx, N, H, W, C);
  } else {
    return FI_ERR_DTYPE;
  }
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_maxpool2_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int C,
                               int accumulate, void* stream) {
  return maxpool_bwd_impl(dtype, x, dy, accumulate ? dx : nullptr, dx, N, H, W, C, stream);
}

extern "C" int fi_maxpool2_bwd_add(int dtype, const void* x, const void* dy, const void* add, void* dx, int N, int H,
                                   int W, int C, void* stream) {
  if (!add) return FI_ERR_NULL;
  return maxpool_bwd_impl(dtype, x, dy, add, dx, N, H, W, C, stream);
}

// ------------------------------------------------------------------------------------------------
// bilinear x2, align_corners=True
//   src = dst * (in-1)/(out-1); i0 = floor(src) clamped; i1 = min(i0+1, in-1); l1 = src - i0.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lin_coord(int o, int in, float sc, int& i0, int& i1, float& l0, float& l1) {
  const float src = sc * (float)o;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + 1 < in ? i0 + 1 : in - 1;
  l1 = src - (float)i0;
  if (l1 < 0.f) l1 = 0.f;
  if (l1 > 1.f) l1 = 1.f;
  l0 = 1.f - l1;
}

// (a one-wavefront-per-output-row form of this kernel, without the div/mod, measured slower: 13.9 vs 11.6 us at 256^2)
template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int h,
                                                           int w, int C, float sh, float sw) {
  constexpr int VG = DT<T>::VG;
  const int CV = C / VG, Ho = 2 * h, Wo = 2 * w;
  const long nvec = (long)N * Ho * Wo * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)((unsigned)i % (unsigned)CV);
    unsigned p = (unsigned)i / (unsigned)CV;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    lin_coord(oy, h, sh, y0, y1, ly0, ly1);
    lin_coord(ox, w, sw, x0, x1, lx0, lx1);
    const T* b = x + (size_t)n * h * w * C + cv * VG;
    float a00[VG], a01[VG], a10[VG], a11[VG], o[VG];
    VecWords<T>::unpack(*reinterpret_cast<const typename DT<T>::vec_t*>(b + ((size_t)y0 * w + x0) * C), a00);
    VecWords<T>::unpack(*reinterpret_cast<const typename DT<T>::vec_t*>(b + ((size_t)y0 * w + x1) * C), a01);
    VecWords<T>::unpack(*reinterpret_cast<const typename DT<T>::vec_t*>(b + ((size_t)y1 * w + x0) * C), a10);
    VecWords<T>::unpack(*reinterpret_cast<const typename DT<T>::vec_t*>(b + ((size_t)y1 * w + x1) * C), a11);
#pragma unroll
    for (int j = 0; j < VG; ++j)
      o[j] = fi_lerp2(ly0, fi_lerp2(lx0, a00[j], lx1, a01[j]), ly1, fi_lerp2(lx0, a10[j], lx1, a11[j]));
    *reinterpret_cast<typename DT<T>::vec_t*>(y + i * VG) = VecWords<T>::pack(o);
  }
}

// Row form for the big launches (the batched LC forwards up-sample 84 images at a time): the flat form above spends ~250
// vector instructions per 16-byte output vector, most of them three 32-bit divisions and the row's interpolation weights --
// it is VALU-bound at 3.2 TB/s.  Here a workgroup owns output rows (row index and row weights are wave-uniform scalars),
// a thread walks the row's vectors with a shift for the channel-vector split (C / VG a power of two).  The interpolation
// expression is the flat form's, operand for operand: the two give the same bits.
template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int h,
                                                                int w, int C, float sh, float sw, int cv_shift) {
  constexpr int VG = DT<T>::VG;
  const int Ho = 2 * h, Wo = 2 * w, rowv = Wo << cv_shift;       // vectors per output row
  const int nrows = N * Ho;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int n = row / Ho, oy = row - n * Ho;
    int y0, y1;
    float ly0, ly1;
    lin_coord(oy, h, sh, y0, y1, ly0, ly1);
    const T* const r0 = x + ((size_t)n * h + y0) * w * C;
    const T* const r1 = x + ((size_t)n * h + y1) * w * C;
    T* const yo = y + (size_t)row * Wo * C;
    for (int i = threadIdx.x; i < rowv; i += 256) {
      const int ox = i >> cv_shift, cv = i - (ox << cv_shift);
      int x0, x1;
      float lx0, lx1;
      lin_coord(ox, w, sw, x0, x1, lx0, lx1);
      const int o0 = x0 * C + cv * VG, o1 = x1 * C + cv * VG;
      float a00[VG], a01[VG], a10[VG], a11[VG], o[VG];
      VecWords<T>::unpack(*reinterpret_cast<const typename DT<T>::vec_t*>(r0 + o0), a00);
      VecWords<T>::unpack(*reinterpret_cast<const typename DT<T>::vec_t*>(r0 + o1), a01);
      VecWords<T>::unpack(*reinterpret_cast<const typename DT<T>::vec_t*>(r1 + o0), a10);
      VecWords<T>::unpack(*reinterpret_cast<const typename DT<T>::vec_t*>(r1 + o1), a11);
#pragma unroll
      for (int j = 0; j < VG; ++j)
        o[j] = fi_lerp2(ly0, fi_lerp2(lx0, a00[j], lx1, a01[j]), ly1, fi_lerp2(lx0, a10[j], lx1, a11[j]));
      *reinterpret_cast<typename DT<T>::vec_t*>(yo + (size_t)i * VG) = VecWords<T>::pack(o);
    }
  }
}

// backward as a gather: for every input pixel collect the outputs whose 2x2 footprint touches it.
// Along one axis the outputs touching input coordinate i are a contiguous run of at most 5 (the open interval
// (i-1, i+1) / scale has length 4 + 2/(in-1)): `lo` = its first output, wt[k] = the weight output lo+k gives to i
// (0 beyond the run / the tensor).  The weights come from lin_coord, i.e. they are exactly the forward's.
__device__ __forceinline__ void up_taps(int i, int in, int out, float sc, int& lo, float (&wt)[5]) {
  auto weight = [&](int o) {
    if (o > out - 1) return 0.f;
    int i0, i1;
    float l0, l1;
    lin_coord(o, in, sc, i0, i1, l0, l1);
    float wv = 0.f;
    if (i0 == i) wv += l0;
    if (i1 == i) wv += l1;
    return wv;
  };
  int c = 0;
  if (sc > 0.f) {
    c = (int)floorf((float)(i - 1) / sc) - 1;   // at most 2 (+ rounding) before the first contributing output
    if (c < 0) c = 0;
  }
  for (int t = 0; t < 4 && weight(c) == 0.f; ++t) ++c;
  lo = c;
#pragma unroll
  for (int k = 0; k < 5; ++k) wt[k] = weight(c + k);
}

// One wavefront per input row (n, iy): the row taps are the same for all its lanes, the lanes run over
// (ix, channel vector).  All 5 x 5 candidate vectors are requested unconditionally (clamped address, zero weight) before
// any is used: the previous form walked the candidates with loads under data-dependent branches, i.e. one memory round
// trip per tap (19 us for the 256^2 level, 9 us even for a 1.5 MB tensor).
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int h,
                                                           int w, int C, float sh, float sw, int accumulate) {
  constexpr int VG = DT<T>::VG;
  typedef typename DT<T>::vec_t vec_t;
  const int CV = C / VG, Ho = 2 * h, Wo = 2 * w;
  const int per_row = w * CV, chunks = (per_row + 63) / 64;
  const long total = (long)N * h * chunks;
  const int lane = threadIdx.x & 63;
  // column taps depend on ix only: tabulated once per workgroup (the tap search is ~20 lin_coord evaluations)
  constexpr int TABLE = 512;
  __shared__ float s_wx[TABLE][5];
  __shared__ int s_lo[TABLE];
  const bool tabled = w <= TABLE;
  if (tabled) {
    for (int i = threadIdx.x; i < w; i += 256) {
      int lo;
      float wt[5];
      up_taps(i, w, Wo, sw, lo, wt);
      s_lo[i] = lo;
#pragma unroll
      for (int k = 0; k < 5; ++k) s_wx[i][k] = wt[k];
    }
    __syncthreads();
  }
  for (long wvi = (long)blockIdx.x * 4 + (threadIdx.x >> 6); wvi < total; wvi += (long)gridDim.x * 4) {
    const int chunk = (int)(wvi % chunks);
    const long row = wvi / chunks;
    const int iy = (int)(row % h), n = (int)(row / h);
    const int e = chunk * 64 + lane;
    const bool active = e < per_row;
    const int ee = active ? e : 0;
    const int ix = ee / CV, cv = ee % CV;
    int oy_lo, ox_lo;
    float wy[5], wx[5];
    up_taps(iy, h, Ho, sh, oy_lo, wy);
    if (tabled) {
      ox_lo = s_lo[ix];
#pragma unroll
      for (int k = 0; k < 5; ++k) wx[k] = s_wx[ix][k];
    } else {
      up_taps(ix, w, Wo, sw, ox_lo, wx);
    }
    const T* b = dy + (size_t)n * Ho * Wo * C + cv * VG;
    vec_t v[5][5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int oy = min(oy_lo + k, Ho - 1);
#pragma unroll
      for (int l = 0; l < 5; ++l) {
        const int ox = min(ox_lo + l, Wo - 1);
        v[k][l] = load_raw<T>(b + ((size_t)oy * Wo + ox) * C);
      }
    }
    float acc[VG];
#pragma unroll
    for (int j = 0; j < VG; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
      for (int l = 0; l < 5; ++l) {
        float g[VG];
        unpack<T>(v[k][l], g);
        const float wgt = wy[k] * wx[l];
#pragma unroll
        for (int j = 0; j < VG; ++j) acc[j] += wgt * g[j];
      }
    if (active) {
      T* o = dx + (((size_t)n * h + iy) * w + ix) * C + cv * VG;
      if (accumulate) {
        float old[VG];
        load_vec<T>(o, old);
#pragma unroll
        for (int j = 0; j < VG; ++j) acc[j] += old[j];
      }
      store_vec<T>(o, acc);
    }
  }
}

// Separable row form of the backward for the launches that matter (the four UpBlocks of a 12-image backward pass: 128 us at 0.15-0.25
// of their HBM floor in the gather form above, which requests 25 vectors and spends ~750 vector instructions per 16 bytes of dx).
// The interpolation is a tensor product, so its transpose is too:  dx[iy][ix] = sum_l wx[l] * ( sum_k wy[k] * dy[oy_lo+k][ox_lo+l] ).
// A workgroup owns a run of input rows of one launch: per input row the inner sums over the (at most five) output rows become ONE
// fp32 row of 2w x C in LDS -- five coalesced loads per vector, row weights wave-uniform -- and the outer sums read it back five
// times per dx vector.  10 global vector requests per dx vector instead of 25, ~200 instructions instead of ~750; the column taps
// are tabulated once per workgroup, the row taps of its run by its first lanes.  Weights are lin_coord's, i.e. the forward's.
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_rows_kernel(const T* __restrict__ dy, T* __restrict__ dx, int nrows, int h, int w,
                                                                int C, float sh, float sw, int accumulate, int rpw, int cv_shift) {
  constexpr int VG = DT<T>::VG;
  constexpr int TABLE = 512, RUN = 16;
  extern __shared__ __attribute__((aligned(16))) float4 s_t[];                // [2w][C] fp32: the row-combined gradient of the current input row
  constexpr int Q = VG / 4;                         // float4s per vector
  __shared__ float s_wx[TABLE][5], s_wy[RUN][5];
  __shared__ int s_lox[TABLE], s_loy[RUN];
  const int Ho = 2 * h, Wo = 2 * w, CV = 1 << cv_shift;
  // workgroups are dealt round-robin to the 8 XCDs, each with an L2 of its own: runs that share halo rows (neighbours) go to the SAME
  // XCD -- workgroup b takes run (b % 8) * (gridDim.x / 8) + b / 8 (the grid is a multiple of 8; runs past the end do nothing)
  const int run = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  const int row0 = run * rpw, row1 = min(row0 + rpw, nrows);
  if (row0 >= nrows) return;
  for (int i = threadIdx.x; i < w; i += 256) {
    int lo;
    float wt[5];
    up_taps(i, w, Wo, sw, lo, wt);
    s_lox[i] = lo;
#pragma unroll
    for (int k = 0; k < 5; ++k) s_wx[i][k] = wt[k];
  }
  if ((int)threadIdx.x < row1 - row0) {
    int lo;
    float wt[5];
    up_taps((row0 + (int)threadIdx.x) % h, h, Ho, sh, lo, wt);
    s_loy[threadIdx.x] = lo;
#pragma unroll
    for (int k = 0; k < 5; ++k) s_wy[threadIdx.x][k] = wt[k];
  }
  __syncthreads();
  const int rowv = Wo << cv_shift, inv = w << cv_shift;      // vectors of an output row / of an input row
  for (int row = row0; row < row1; ++row) {
    const int n = row / h, r = row - row0;
    const int oy_lo = s_loy[r];
    float wy[5];
    const T* src[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      wy[k] = s_wy[r][k];
      src[k] = dy + ((size_t)n * Ho + min(oy_lo + k, Ho - 1)) * Wo * C;
    }
#pragma clang loop vectorize(disable) interleave(disable)   // (else: packed FMAs over lanes of TWO vectors, the row stored word by word)
    for (int i = threadIdx.x; i < rowv; i += 256) {
      typename DT<T>::vec_t v[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) v[k] = load_raw<T>(src[k] + (size_t)i * VG);
      float acc[VG];
#pragma unroll
      for (int j = 0; j < VG; ++j) acc[j] = 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        float g[VG];
        unpack<T>(v[k], g);
#pragma unroll
        for (int j = 0; j < VG; ++j) acc[j] += wy[k] * g[j];
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) s_t[i * Q + q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    __syncthreads();
    T* const out = dx + (size_t)row * w * C;
    for (int e = threadIdx.x; e < inv; e += 256) {
      const int ix = e >> cv_shift, cv = e - (ix << cv_shift);
      const int lo = s_lox[ix];
      float acc[VG];
#pragma unroll
      for (int j = 0; j < VG; ++j) acc[j] = 0.f;
#pragma unroll
      for (int l = 0; l < 5; ++l) {
        const float wl = s_wx[ix][l];
        const float4* t = s_t + (((min(lo + l, Wo - 1)) << cv_shift) + cv) * Q;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const float4 tv = t[q];
          acc[4 * q] += wl * tv.x, acc[4 * q + 1] += wl * tv.y, acc[4 * q + 2] += wl * tv.z, acc[4 * q + 3] += wl * tv.w;
        }
      }
      T* o = out + (size_t)e * VG;
      if (accumulate) {
        float old[VG];
        load_vec<T>(o, old);
#pragma unroll
        for (int j = 0; j < VG; ++j) acc[j] += old[j];
      }
      store_vec<T>(o, acc);
    }
    __syncthreads();                                 // s_t is rewritten by the next row
  }
}

static inline float up_scale(int in) { return in > 1 ? (float)(in - 1) / (float)(2 * in - 1) : 0.f; }

// gather form for small launches, separable row form where a launch has rows enough to fill the chip (FI_UPBWD_ROWS=0: gather form
// everywhere, A/B runs; FI_UPBWD_WGS: workgroups of the row form -- default one per input row).  tools/upbench.py --bwd --images 12,
// the four levels of a 12-image backward pass, us: gather form 10.4 / 15.4 / 27.8 / 50.1 = 103.6; row form with its runs dealt
// round-robin to the XCDs like the workgroups are 8.3 / 15.2 / 24.7 / 46.1 = 94.3 (runs of 4 rows; 8 rows: 88.9 -- the halo rows of
// neighbouring runs were fetched once per XCD); with neighbouring runs on one XCD 7.6 / 9.4 / 19.7 / 36.7 = 73.4 at one row per
// workgroup (runs of 2: 78.6, of 4: 83.7, of 16: 104.6).
template <typename T>
static int launch_upsample_bwd(const void* dy, void* dx, int N, int h, int w, int C, int accumulate, hipStream_t st) {
  constexpr int VG = DT<T>::VG;
  if (C % VG) return FI_ERR_SHAPE;
  const int CV = C / VG;
  const long nvec = (long)N * h * w * CV;
  if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
  static const long rows_on = [] {
    const char* v = getenv("FI_UPBWD_ROWS");
    return v ? atol(v) : 1L;
  }();
  static const long wgs = [] {
    const char* v = getenv("FI_UPBWD_WGS");
    return v && atol(v) > 0 ? atol(v) : (1L << 20);
  }();
  int shift = 0;
  while ((1 << shift) < CV) ++shift;
  const long nrows = (long)N * h, lds = (long)2 * w * C * (long)sizeof(float);
  if (rows_on && (1 << shift) == CV && w <= 512 && lds <= 48 * 1024 && nrows >= 64 && nrows < (1L << 31) && (long)w * CV >= 64) {
    long rpw = (nrows + wgs - 1) / wgs;
    if (rpw > 16) rpw = 16;
    const long blocks = ((nrows + rpw - 1) / rpw + 7) / 8 * 8;
    hipLaunchKernelGGL(upsample_bwd_rows_kernel<T>, dim3((unsigned)blocks), dim3(256), (size_t)lds, st, (const T*)dy, (T*)dx, (int)nrows,
                       h, w, C, up_scale(h), up_scale(w), accumulate, (int)rpw, shift);
  } else {
    hipLaunchKernelGGL(upsample_bwd_kernel<T>, dim3(grid_for((long)N * h * ((w * CV + 63) / 64), 4)), dim3(256), 0, st, (const T*)dy,
                       (T*)dx, N, h, w, C, up_scale(h), up_scale(w), accumulate);
  }
  FI_CHECK_LAUNCH();
  return 0;
}

// launches big enough to be VALU-bound in the flat form take the row form (FI_UP_ROWS=0 keeps the flat form: A/B runs)
template <typename T>
static int launch_upsample_fwd(const void* x, void* y, int N, int h, int w, int C, hipStream_t st) {
  constexpr int VG = DT<T>::VG;
  if (C % VG) return FI_ERR_SHAPE;
  const int CV = C / VG;
  const long nvec = (long)N * 4 * h * w * CV;
  if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;  // kernels index vectors with 32 bits
  static const long rows_min = [] {
    const char* v = getenv("FI_UP_ROWS_MIN");
    return v ? atol(v) : (1L << 21);
  }();
  int shift = 0;
  while ((1 << shift) < CV) ++shift;
  const long rowv = (long)2 * w * CV;
  if ((1 << shift) == CV && nvec >= rows_min && rowv >= 256 && (long)N * 2 * h < (1L << 31)) {
    long blocks = (long)N * 2 * h;
    // one workgroup per output row (tools/upbench.py, 84 x 256^2 x 16 -> 512^2: 4096 persistent workgroups 270 us, one per row 253;
    // unrolling the row walk 2x / 4x: 275 / 280).  3.3-4.1 TB/s of read-once + write-once traffic at every level.
    if (blocks > (1L << 20)) blocks = 1L << 20;
    hipLaunchKernelGGL(upsample_fwd_rows_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (T*)y, N, h, w, C,
                       up_scale(h), up_scale(w), shift);
  } else {
    hipLaunchKernelGGL(upsample_fwd_kernel<T>, dim3(grid_for(nvec, 256 * 2)), dim3(256), 0, st, (const T*)x, (T*)y, N, h, w, C,
                       up_scale(h), up_scale(w));
  }
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_upsample2x_fwd(int dtype, const void* x, void* y, int N, int h, int w, int C, void* stream) {
  if (!x || !y) return FI_ERR_NULL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FI_F32) return launch_upsample_fwd<float>(x, y, N, h, w, C, st);
  if (dtype == FI_BF16) return launch_upsample_fwd<bf16_t>(x, y, N, h, w, C, st);
  if (dtype == FI_F16) return launch_upsample_fwd<f16_t>(x, y, N, h, w, C, st);
  return FI_ERR_DTYPE;
}

extern "C" int fi_upsample2x_bwd(int dtype, const void* dy, void* dx, int N, int h, int w, int C, int accumulate,
                                 void* stream) {
  if (!dy || !dx) return FI_ERR_NULL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FI_F32) return launch_upsample_bwd<float>(dy, dx, N, h, w, C, accumulate, st);
  if (dtype == FI_BF16) return launch_upsample_bwd<bf16_t>(dy, dx, N, h, w, C, accumulate, st);
  if (dtype == FI_F16) return launch_upsample_bwd<f16_t>(dy, dx, N, h, w, C, accumulate, st);
  return FI_ERR_DTYPE;
}

// ------------------------------------------------------------------------------------------------
// weight repack / casts / layout
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ src, T* __restrict__ dst, int cout, int kk, int cin,
                                    int mode) {
  const long n = (long)cout * kk * cin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (mode == 0) {
      dst[i] = from_f32<T>(src[i]);
    } else {
      const int ci = (int)(i % cin);
      const int t = (int)((i / cin) % kk);
      const int co = (int)(i / ((long)cin * kk));
      if (mode == 1)
        dst[((size_t)ci * kk + (kk - 1 - t)) * cout + co] = from_f32<T>(src[i]);
      else if (mode == 2)
        dst[(((size_t)(ci >> 4) * cout + co) * kk + t) * 16 + (ci & 15)] = from_f32<T>(src[i]);
      else
        dst[(((size_t)(co >> 4) * cin + ci) * kk + (kk - 1 - t)) * 16 + (co & 15)] = from_f32<T>(src[i]);
    }
  }
}

extern "C" int fi_pack_weights(const float* src, void* dst, int cout, int kk, int cin, int mode, int dtype,
                               void* stream) {
  if (!src || !dst) return FI_ERR_NULL;
  if (mode < 0 || mode > 3) return FI_ERR_UNSUPPORTED;
  if ((mode == 2 && cin % 16) || (mode == 3 && cout % 16)) return FI_ERR_SHAPE;
  const long n = (long)cout * kk * cin;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FI_F32)
    hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, st, src, (float*)dst, cout,
                       kk, cin, mode);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(grid_for(n, 256)), dim3(256), 0, st, src, (bf16_t*)dst, cout,
                       kk, cin, mode);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(pack_weights_kernel<f16_t>, dim3(grid_for(n, 256)), dim3(256), 0, st, src, (f16_t*)dst, cout,
                       kk, cin, mode);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ s, D* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    d[i] = from_f32<D>(to_f32(s[i]));
}

extern "C" int fi_cast(const void* src, int sd, void* dst, int dd, long n, void* stream) {
  if (!src || !dst) return FI_ERR_NULL;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid_for(n, 256)), b(256);
  if (sd == FI_F32 && dd == FI_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, st, (const float*)src, (bf16_t*)dst, n);
  else if (sd == FI_BF16 && dd == FI_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t*)src, (float*)dst, n);
  else if (sd == FI_F32 && dd == FI_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, (const float*)src, (float*)dst, n);
  else if (sd == FI_BF16 && dd == FI_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
  else if (sd == FI_F32 && dd == FI_F16)
    hipLaunchKernelGGL((cast_kernel<float, f16_t>), g, b, 0, st, (const float*)src, (f16_t*)dst, n);
  else if (sd == FI_F16 && dd == FI_F32)
    hipLaunchKernelGGL((cast_kernel<f16_t, float>), g, b, 0, st, (const f16_t*)src, (float*)dst, n);
  else if (sd == FI_F16 && dd == FI_F16)
    hipLaunchKernelGGL((cast_kernel<f16_t, f16_t>), g, b, 0, st, (const f16_t*)src, (f16_t*)dst, n);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ s, T* __restrict__ d, int N, int C, int H, int W) {
  const long n = (long)N * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long p = i / C;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    d[i] = from_f32<T>(s[(((size_t)b * C + c) * H + y) * W + x]);
  }
}
// C <= 4 planes (the network input, 1 or 3 channels): a thread converts TWO neighbouring pixels -- one 8-byte load per plane
// (coalesced along the row), 2 C consecutive 16-bit results.  The element-wise form above spends three 64-bit divisions per
// element: 30 us for 12 x 3 x 512^2 (56 MB of traffic).
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_narrow_kernel(const float* __restrict__ s, T* __restrict__ d, int N, int C, int HW) {
  const int pairs = HW / 2;
  const long total = (long)N * pairs;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / pairs), p = (int)(i - (long)b * pairs) * 2;
    const float* src = s + (size_t)b * C * HW + p;
    T* dst = d + ((size_t)b * HW + p) * C;
    float v[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < C) {
        const float2 t = *reinterpret_cast<const float2*>(src + (size_t)c * HW);
        v[0][c] = t.x, v[1][c] = t.y;
      }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < C) dst[q * C + c] = from_f32<T>(v[q][c]);
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ s, float* __restrict__ d, int N, int C, int H, int W) {
  const long n = (long)N * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long p = i / W;
    const int y = (int)(p % H);
    p /= H;
    const int c = (int)(p % C);
    const int b = (int)(p / C);
    d[i] = to_f32(s[(((size_t)b * H + y) * W + x) * C + c]);
  }
}

extern "C" int fi_nchw_to_nhwc(const float* src, void* dst, int dtype, int N, int C, int H, int W, void* stream) {
  if (!src || !dst) return FI_ERR_NULL;
  const long n = (long)N * C * H * W;
  hipStream_t st = (hipStream_t)stream;
  const long hw = (long)H * W;
  if (C <= 4 && hw % 2 == 0 && hw < (1L << 30) && dtype != FI_F32) {
    const unsigned g = grid_for((long)N * (hw / 2), 256);
    if (dtype == FI_BF16)
      hipLaunchKernelGGL(nchw_to_nhwc_narrow_kernel<bf16_t>, dim3(g), dim3(256), 0, st, src, (bf16_t*)dst, N, C, (int)hw);
    else if (dtype == FI_F16)
      hipLaunchKernelGGL(nchw_to_nhwc_narrow_kernel<f16_t>, dim3(g), dim3(256), 0, st, src, (f16_t*)dst, N, C, (int)hw);
    else
      return FI_ERR_DTYPE;
    FI_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == FI_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, st, src, (float*)dst, N, C,
                       H, W);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(n, 256)), dim3(256), 0, st, src, (bf16_t*)dst, N, C,
                       H, W);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16_t>, dim3(grid_for(n, 256)), dim3(256), 0, st, src, (f16_t*)dst, N, C,
                       H, W);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_nhwc_to_nchw(const void* src, int dtype, float* dst, int N, int C, int H, int W, void* stream) {
  if (!src || !dst) return FI_ERR_NULL;
  const long n = (long)N * C * H * W;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FI_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const float*)src, dst,
                       N, C, H, W);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const bf16_t*)src, dst,
                       N, C, H, W);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16_t>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const f16_t*)src, dst,
                       N, C, H, W);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// multi-tensor weight repack: ONE launch for every conv weight of a model (the per-tensor launches
// were ~45 x 5 us per training step).  table[t] = {src, dst_fwd, dst_dgrad, cout, kk, cin, dst_fwd16, dst_dgrad16}
// (int64 x FI_PACK_ROW; the last two are the chunk-major forms of conv_fwd_ws2_kernel, or 0).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const long long* __restrict__ table) {
  // 32 (cout) x 32 (cin) tiles per filter tap, transposed through LDS: reads are coalesced along cin, the dgrad
  // operand's writes along cout (the element-wise scatter this replaces took 44 us for the U-Net's 1.8 M weights)
  __shared__ float tile[32][33];
  const long long* d = table + (size_t)blockIdx.y * FI_PACK_ROW;
  const float* src = reinterpret_cast<const float*>(d[0]);
  T* dst0 = reinterpret_cast<T*>(d[1]);
  T* dst1 = reinterpret_cast<T*>(d[2]);
  T* dst2 = reinterpret_cast<T*>(d[6]);        // chunk-major forms (16-channel chunks of the contraction), or NULL
  T* dst3 = reinterpret_cast<T*>(d[7]);
  const int cout = (int)d[3], kk = (int)d[4], cin = (int)d[5];
  const int tco = (cout + 31) / 32, tci = (cin + 31) / 32;
  const int ntiles = tco * tci * kk;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int t = tl % kk, ci0 = ((tl / kk) % tci) * 32, co0 = (tl / (kk * tci)) * 32;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + ty + r * 8, ci = ci0 + tx;
      float v = 0.f;
      if (co < cout && ci < cin) {
        const size_t i = ((size_t)co * kk + t) * cin + ci;
        v = src[i];
        if (dst0) dst0[i] = from_f32<T>(v);
        if (dst2) dst2[(((size_t)(ci >> 4) * cout + co) * kk + t) * 16 + (ci & 15)] = from_f32<T>(v);
      }
      tile[ty + r * 8][tx] = v;
    }
    __syncthreads();
    if (dst1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + ty + r * 8, co = co0 + tx;
        if (co < cout && ci < cin) dst1[((size_t)ci * kk + (kk - 1 - t)) * cout + co] = from_f32<T>(tile[tx][ty + r * 8]);
      }
    }
    if (dst3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + ty + r * 8, co = co0 + tx;
        if (co < cout && ci < cin)
          dst3[(((size_t)(co >> 4) * cin + ci) * kk + (kk - 1 - t)) * 16 + (co & 15)] = from_f32<T>(tile[tx][ty + r * 8]);
      }
    }
  }
}

extern "C" int fi_pack_weights_multi(const long long* table, int ntensors, int dtype, void* stream) {
  if (!table) return FI_ERR_NULL;
  if (ntensors <= 0) return 0;
  // one workgroup per 32x32 tile of the LARGEST tensor (a 256x256 3x3 filter has 576); the surplus workgroups of the
  // smaller tensors exit at once.  64 workgroups per tensor walked 9 tiles each: 21.0 us, 576: 9.0 us.
  static const long xblocks = [] { const char* e = getenv("FI_PACK_BLOCKS"); return e && *e ? atol(e) : 576L; }();
  const dim3 g((unsigned)xblocks, ntensors), b(256);
  if (dtype == FI_F32)
    hipLaunchKernelGGL(pack_weights_multi_kernel<float>, g, b, 0, (hipStream_t)stream, table);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(pack_weights_multi_kernel<bf16_t>, g, b, 0, (hipStream_t)stream, table);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(pack_weights_multi_kernel<f16_t>, g, b, 0, (hipStream_t)stream, table);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// fused BN finalize + apply + activation + dropout (forward): every workgroup folds the statistic slots of
// ALL channels into LDS (2C doubles x 32 slots, L2-resident), derives scale/shift itself, and streams;
// workgroup 0 additionally publishes scale/shift/mean/invstd for backward and updates the running
// statistics.  Removes one ~5 us launch per BatchNorm per forward.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_fused_fwd_kernel(const T* __restrict__ y, T* __restrict__ z,
                                                           const double* __restrict__ stats, double count,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* rmean, float* rvar,
                                                           int64_t* nbt, float momentum, float eps, int training,
                                                           float* __restrict__ coef, long nvec, int C, float slope,
                                                           DropSpec dr, BnBatch bb) {
  constexpr int VG = DT<T>::VG;
  if (bb.n > 1) {
    const long b = blockIdx.y;
    y += b * bb.tensor, z += b * bb.tensor, coef += b * bb.coef;
    if (stats) stats += b * bb.stats;
  }
  __shared__ float s_scale[512], s_shift[512];
  // (requesting the first batch of y ahead of this prologue was measured: no gain on the small maps, 12.7 -> 14.7 us
  // on the 256^2 level -- the extra live registers cost more than the overlapped round trip saves)
  for (int c = threadIdx.x; c < C; c += 256) {
    float mu, istd;
    double unb = 0.0;
    if (training) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int slot = 0; slot < FI_STATS_SLOTS; ++slot) {
        s1 += stats[((size_t)slot * C + c) * 2];
        s2 += stats[((size_t)slot * C + c) * 2 + 1];
      }
      const double m = s1 / count;
      double var = s2 / count - m * m;
      if (var < 0.0) var = 0.0;
      mu = (float)m;
      istd = (float)(1.0 / sqrt(var + (double)eps));
      unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
    } else {
      mu = rmean[c];
      istd = 1.0f / sqrtf(rvar[c] + eps);
    }
    const float sc = gamma[c] * istd, sh = beta[c] - mu * sc;
    s_scale[c] = sc;
    s_shift[c] = sh;
    if (blockIdx.x == 0) {
      coef[c] = sc;
      coef[C + c] = sh;
      coef[2 * C + c] = mu;
      coef[3 * C + c] = istd;
      if (training) {
        bn_running_update(rmean, rvar, c, momentum, mu, (float)unb);
        if (c == 0 && nbt) nbt[0] += 1;
      }
    }
  }
  __syncthreads();
  const unsigned CV = C / VG;
  const int c0 = (int)(threadIdx.x % CV) * VG;
  float sc[VG], sh[VG];
#pragma unroll
  for (int j = 0; j < VG; ++j) {
    sc[j] = s_scale[c0 + j];
    sh[j] = s_shift[c0 + j];
  }
  const uint64_t seed = drop_seed(dr);
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x, istep = gridDim.x * 256u;
  unsigned pixel = i0 / CV;
  const unsigned pstep = istep / CV;
  // MLP: 4 independent 16-byte loads per thread are issued before any is consumed (the plain grid-stride
  // loop exposed one HBM round trip per vector)
  for (long i = i0; i < nvec; i += 4L * istep, pixel += 4 * pstep) {
    float f[4][VG];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long iu = i + (long)u * istep;
      load_vec<T>(y + (iu < nvec ? iu : i) * VG, f[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long iu = i + (long)u * istep;
      if (iu < nvec) {
        float df[VG];
        drop_factors<VG>(dr, seed, (size_t)iu, pixel + u * pstep, c0, df);
#pragma unroll
        for (int j = 0; j < VG; ++j) {
          float v = f[u][j] * sc[j] + sh[j];
          v = v > 0.f ? v : v * slope;
          v *= df[j];
          f[u][j] = v;
        }
        store_vec<T>(z + iu * VG, f[u]);
      }
    }
  }
}

static int bn_fused_fwd_impl(const FiBnAct* d, const void* y, void* z, const double* stats, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps, int training,
                             float* coef, BnBatch bb, void* stream) {
  if (!d || !y || !z || !gamma || !beta || !running_mean || !running_var || !coef) return FI_ERR_NULL;
  if (training && !stats) return FI_ERR_NULL;
  if (d->dtype != FI_F32 && d->dtype != FI_BF16 && d->dtype != FI_F16) return FI_ERR_DTYPE;
  const int vg = d->dtype == FI_F32 ? 4 : 8;
  if (d->C % vg || d->C > 512 || 256 % (d->C / vg)) return FI_ERR_SHAPE;
  const long nvec = d->pixels * (d->C / vg);
  if (nvec >= (1L << 32)) return FI_ERR_UNSUPPORTED;
  const DropSpec dr = make_drop(d);
  hipStream_t st = (hipStream_t)stream;
  // race note: workgroup 0 updates running_mean/var while other workgroups read them only in eval mode, where
  // nothing is written; in training mode nobody reads them.
  if (d->dtype == FI_F32)
    hipLaunchKernelGGL(bn_fused_fwd_kernel<float>, dim3(grid_for(nvec, 256 * 4), bb.n), dim3(256), 0, st, (const float*)y,
                       (float*)z, stats, (double)d->pixels, gamma, beta, running_mean, running_var, nbt, momentum,
                       eps, training, coef, nvec, d->C, d->slope, dr, bb);
  else if (d->dtype == FI_F16)
    hipLaunchKernelGGL(bn_fused_fwd_kernel<f16_t>, dim3(grid_for(nvec, 256 * 4), bb.n), dim3(256), 0, st,
                       (const f16_t*)y, (f16_t*)z, stats, (double)d->pixels, gamma, beta, running_mean,
                       running_var, nbt, momentum, eps, training, coef, nvec, d->C, d->slope, dr, bb);
  else
    hipLaunchKernelGGL(bn_fused_fwd_kernel<bf16_t>, dim3(grid_for(nvec, 256 * 4), bb.n), dim3(256), 0, st,
                       (const bf16_t*)y, (bf16_t*)z, stats, (double)d->pixels, gamma, beta, running_mean,
                       running_var, nbt, momentum, eps, training, coef, nvec, d->C, d->slope, dr, bb);
  FI_CHECK_LAUNCH();
  return 0;
}
extern "C" int fi_bn_fused_fwd(const FiBnAct* d, const void* y, void* z, const double* stats, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                               float momentum, float eps, int training, float* coef, void* stream) {
  return bn_fused_fwd_impl(d, y, z, stats, gamma, beta, running_mean, running_var, nbt, momentum, eps, training, coef,
                           BnBatch{1, 0, 0, 0}, stream);
}
// `nbatch` samples of one launch, each with its own statistics (stats_stride doubles apart), coefficient rows (coef [nbatch][4][C]:
// coef_stride = 4 * C floats) and slice of y / z (tensor_stride elements apart): InstanceNorm3d(affine=False) + ReLU per sample
// (/root/reference/code/networks/utils.py:106-110) -- gamma / beta the constant (1, 0) rows, momentum 0 (the running statistics
// are scratch and left as they are), no dropout.
extern "C" int fi_bn_fused_fwd_batched(const FiBnAct* d, int nbatch, long tensor_stride, long stats_stride, long coef_stride,
                                       const void* y, void* z, const double* stats, const float* gamma, const float* beta,
                                       float* running_scratch_mean, float* running_scratch_var, float eps, float* coef,
                                       void* stream) {
  const int rc = bn_batch_ok(d, nbatch, tensor_stride, stats_stride, coef_stride);
  if (rc) return rc;
  return bn_fused_fwd_impl(d, y, z, stats, gamma, beta, running_scratch_mean, running_scratch_var, nullptr, 0.f, eps, 1, coef,
                           BnBatch{nbatch, tensor_stride, stats_stride, coef_stride}, stream);
}
