// Loss, metric, optimizer, aggregation and PCS helper kernels (gfx950).
// Reductions: lane-private partials -> wavefront xor-shuffle (64 lanes) -> LDS across the
// workgroup's waves -> one fp64 / int64 atomic per workgroup.
#include "common.h"

static inline int grid_for(long work_items, int per_block) {
  long b = (work_items + per_block - 1) / per_block;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}

#define FI_MAX_CLASSES 8

// block-wide sum of a double; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wv] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
  return t;
}

// ------------------------------------------------------------------------------------------------
// partial cross-entropy
// ------------------------------------------------------------------------------------------------
// acc is FI_CE_SLOTS x {loss sum, count}: 768 workgroups adding to the same two fp64 addresses serialise at the
// fabric (the kernel took 26 us for 7 MB of input); workgroup b adds to slot b % FI_CE_SLOTS and the readers fold.
__device__ __forceinline__ void ce_pixel(const float* z, int C, int lb, int ignore, double& loss, double& cnt) {
  if (lb == ignore) return;
  float mx = z[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
  float se = 0.f;
  for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
  const float zl = (lb >= 0 && lb < C) ? z[lb] : 0.f;
  loss += (double)(logf(se) + mx - zl);
  cnt += 1.0;
}

// Four consecutive pixels per thread wherever M % 4 == 0 and C is 2 or 3 (the segmentation heads): C 16-byte logit loads and one
// 4-byte label load, all issued before the first use -- the pixel-strided loop below it ran 8 dependent round trips per thread
// (36 us for 12 x 512^2 x 3 logits, 38 MB; 0.13 of the HBM roofline).
template <int C>
__device__ __forceinline__ void ce_load4(const float* __restrict__ logits, const uint8_t* __restrict__ labels, long q, float (&z)[4 * C],
                                         uint32_t& lw) {
  const float4* p = reinterpret_cast<const float4*>(logits) + (size_t)C * q;
#pragma unroll
  for (int v = 0; v < C; ++v) {
    const float4 a = p[v];
    z[4 * v] = a.x, z[4 * v + 1] = a.y, z[4 * v + 2] = a.z, z[4 * v + 3] = a.w;
  }
  lw = reinterpret_cast<const uint32_t*>(labels)[q];
}

template <int C>
__device__ __forceinline__ void ce_fwd_vec(const float* __restrict__ logits, const uint8_t* __restrict__ labels, long M4, int ignore, long tid,
                                           long nth, double& loss, double& cnt) {
  for (long q = tid; q < M4; q += nth) {
    float z[4 * C];
    uint32_t lw;
    ce_load4<C>(logits, labels, q, z, lw);
#pragma unroll
    for (int k = 0; k < 4; ++k) ce_pixel(z + C * k, C, (int)((lw >> (8 * k)) & 0xFFu), ignore, loss, cnt);
  }
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits,
                                                     const uint8_t* __restrict__ labels, long M, int C, int ignore,
                                                     double* acc) {
  __shared__ double sm[4];
  double loss = 0.0, cnt = 0.0;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  if (C == 2 && (M & 3) == 0) {
    ce_fwd_vec<2>(logits, labels, M >> 2, ignore, tid, nth, loss, cnt);
  } else if (C == 3 && (M & 3) == 0) {
    ce_fwd_vec<3>(logits, labels, M >> 2, ignore, tid, nth, loss, cnt);
  } else {
    for (long i = tid; i < M; i += nth) ce_pixel(logits + i * C, C, labels[i], ignore, loss, cnt);
  }
  const double tl = block_sum(loss, sm);
  const double tc = block_sum(cnt, sm);
  if (threadIdx.x == 0 && tc > 0.0) {
    const int slot = blockIdx.x & (FI_CE_SLOTS - 1);
    atomicAdd(&acc[2 * slot], tl);
    atomicAdd(&acc[2 * slot + 1], tc);
  }
}

__device__ __forceinline__ double ce_fold(const double* acc, int which) {
  double t = 0.0;
#pragma unroll
  for (int s = 0; s < FI_CE_SLOTS; ++s) t += acc[2 * s + which];
  return t;
}

__global__ void ce_finalize_kernel(const double* acc, float* loss) { loss[0] = (float)(ce_fold(acc, 0) / ce_fold(acc, 1)); }

// one pixel's gradient: (softmax - onehot) / count * gscale, zeros where the label is ignored
__device__ __forceinline__ void ce_grad_pixel(const float* z, int C, int lb, int ignore, float inv, float* g) {
  if (lb == ignore) {
    for (int c = 0; c < C; ++c) g[c] = 0.f;
    return;
  }
  float mx = z[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
  float e[FI_MAX_CLASSES];
  float se = 0.f;
  for (int c = 0; c < C; ++c) {
    e[c] = expf(z[c] - mx);
    se += e[c];
  }
  const float rs = 1.f / se;
  for (int c = 0; c < C; ++c) g[c] = (e[c] * rs - (c == lb ? 1.f : 0.f)) * inv;
}

template <typename T, int C>
__device__ __forceinline__ void ce_bwd_vec(const float* __restrict__ logits, const uint8_t* __restrict__ labels, long M4, int ignore, float inv,
                                           T* __restrict__ dl) {
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < M4; q += (long)gridDim.x * blockDim.x) {
    float z[4 * C], g[4 * C];
    uint32_t lw;
    ce_load4<C>(logits, labels, q, z, lw);
#pragma unroll
    for (int k = 0; k < 4; ++k) ce_grad_pixel(z + C * k, C, (int)((lw >> (8 * k)) & 0xFFu), ignore, inv, g + C * k);
    if constexpr (sizeof(T) == 4) {
      float4* o = reinterpret_cast<float4*>(dl) + (size_t)C * q;
#pragma unroll
      for (int v = 0; v < C; ++v) o[v] = make_float4(g[4 * v], g[4 * v + 1], g[4 * v + 2], g[4 * v + 3]);
    } else {
      typename Quad<T>::q_t* o = reinterpret_cast<typename Quad<T>::q_t*>(dl) + (size_t)C * q;      // 4 C elements = C 8-byte words
#pragma unroll
      for (int v = 0; v < C; ++v) {
        float w[4] = {g[4 * v], g[4 * v + 1], g[4 * v + 2], g[4 * v + 3]};
        o[v] = Quad<T>::pack(w);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits,
                                                     const uint8_t* __restrict__ labels, long M, int C, int ignore,
                                                     const double* __restrict__ acc, const float* gscale,
                                                     T* __restrict__ dl) {
  const float gs = gscale ? gscale[0] : 1.f;
  const float inv = (float)((double)gs / ce_fold(acc, 1));
  if (C == 2 && (M & 3) == 0) return ce_bwd_vec<T, 2>(logits, labels, M >> 2, ignore, inv, dl);
  if (C == 3 && (M & 3) == 0) return ce_bwd_vec<T, 3>(logits, labels, M >> 2, ignore, inv, dl);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long)gridDim.x * blockDim.x) {
    float g[FI_MAX_CLASSES];
    ce_grad_pixel(logits + i * C, C, labels[i], ignore, inv, g);
    T* o = dl + i * C;
    for (int c = 0; c < C; ++c) o[c] = from_f32<T>(g[c]);
  }
}

extern "C" int fi_ce_fwd(const float* logits, const uint8_t* labels, long M, int C, int ignore_index, double* acc,
                         void* stream) {
  if (!logits || !labels || !acc) return FI_ERR_NULL;
  if (C < 1 || C > FI_MAX_CLASSES) return FI_ERR_SHAPE;
  const bool vec = (C == 2 || C == 3) && (M & 3) == 0;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(grid_for(M, vec ? 256 * 4 * 2 : 256 * 8)), dim3(256), 0, (hipStream_t)stream, logits, labels, M,
                     C, ignore_index, acc);
  FI_CHECK_LAUNCH();
  return 0;
}
extern "C" int fi_ce_finalize(const double* acc, float* loss, void* stream) {
  if (!acc || !loss) return FI_ERR_NULL;
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, loss);
  FI_CHECK_LAUNCH();
  return 0;
}
extern "C" int fi_ce_bwd(const float* logits, const uint8_t* labels, long M, int C, int ignore_index,
                         const double* acc, const float* gscale, void* dlogits, int dtype, void* stream) {
  if (!logits || !labels || !acc || !dlogits) return FI_ERR_NULL;
  if (C < 1 || C > FI_MAX_CLASSES) return FI_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FI_F32)
    hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(grid_for(M, 256 * 2)), dim3(256), 0, st, logits, labels, M, C,
                       ignore_index, acc, gscale, (float*)dlogits);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(grid_for(M, 256 * 2)), dim3(256), 0, st, logits, labels, M, C,
                       ignore_index, acc, gscale, (bf16_t*)dlogits);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(ce_bwd_kernel<f16_t>, dim3(grid_for(M, 256 * 2)), dim3(256), 0, st, logits, labels, M, C,
                       ignore_index, acc, gscale, (f16_t*)dlogits);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Dice bookkeeping
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dice_counts_kernel(const float* __restrict__ logits,
                                                          const uint8_t* __restrict__ gt, long M, int C,
                                                          unsigned long long* counts) {
  // per-thread counters for up to FI_MAX_CLASSES-1 foreground classes
  unsigned int inter[FI_MAX_CLASSES - 1], np[FI_MAX_CLASSES - 1], ng[FI_MAX_CLASSES - 1];
  for (int k = 0; k < FI_MAX_CLASSES - 1; ++k) inter[k] = np[k] = ng[k] = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long)gridDim.x * blockDim.x) {
    const float* z = logits + i * C;
    int best = 0;
    float mx = z[0];
    for (int c = 1; c < C; ++c)
      if (z[c] > mx) {
        mx = z[c];
        best = c;
      }
    const int g = gt[i];
    for (int k = 1; k < C; ++k) {
      const bool p = (k == 1) ? (best == 1) : (best >= 1);
      const bool q = (k == 1) ? (g == 1) : (g >= 1);
      inter[k - 1] += (p && q);
      np[k - 1] += p;
      ng[k - 1] += q;
    }
  }
  __shared__ double sm[4];
  for (int k = 0; k < C - 1; ++k) {
    const double a = block_sum((double)inter[k], sm);
    const double b = block_sum((double)np[k], sm);
    const double c = block_sum((double)ng[k], sm);
    if (threadIdx.x == 0) {
      atomicAdd(&counts[k * 3 + 0], (unsigned long long)a);
      atomicAdd(&counts[k * 3 + 1], (unsigned long long)b);
      atomicAdd(&counts[k * 3 + 2], (unsigned long long)c);
    }
  }
}

extern "C" int fi_dice_counts(const float* logits, const uint8_t* gt, long M, int C, long long* counts,
                              void* stream) {
  if (!logits || !gt || !counts) return FI_ERR_NULL;
  if (C < 2 || C > FI_MAX_CLASSES) return FI_ERR_SHAPE;
  hipLaunchKernelGGL(dice_counts_kernel, dim3(grid_for(M, 256 * 4)), dim3(256), 0, (hipStream_t)stream, logits, gt, M,
                     C, (unsigned long long*)counts);
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Surface distances for the 95th-percentile Hausdorff distance (medpy.metric.binary.hd95, val_2D.py:14):
//   border(m) = m AND NOT erode(m) with the 4-neighbourhood cross (connectivity 1; outside the image counts as 0),
//   d(a -> B) = Euclidean distance from border pixel a of one mask to the nearest border pixel of the other.
// fi_seg_borders lists the border pixels of the prediction (argmax of the logits) and of the ground truth for one
// foreground class (class 1: label == 1; classes >= 2: label >= 1, val_2D.py:66-74); fi_surface_distances fills in the
// distances by exhaustive search (exact: integer squared distances, one fp64 square root).  The percentile itself is
// taken on the host from the returned distances.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool seg_is(const float* z, int C, int k) {
  int best = 0;
  float mx = z[0];
  for (int c = 1; c < C; ++c)
    if (z[c] > mx) {
      mx = z[c];
      best = c;
    }
  return k == 1 ? best == 1 : best >= 1;
}
__global__ __launch_bounds__(256) void seg_borders_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ gt,
                                                          int H, int W, int C, int k, int* __restrict__ plist,
                                                          int* __restrict__ glist, int* counts) {
  const long M = (long)H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long)gridDim.x * blockDim.x) {
    const int yy = (int)(i / W), xx = (int)(i % W);
    const int dy[4] = {-1, 1, 0, 0}, dx[4] = {0, 0, -1, 1};
    if (seg_is(logits + i * C, C, k)) {
      bool inner = true;
      for (int q = 0; q < 4; ++q) {
        const int y2 = yy + dy[q], x2 = xx + dx[q];
        inner = inner && y2 >= 0 && y2 < H && x2 >= 0 && x2 < W && seg_is(logits + ((long)y2 * W + x2) * C, C, k);
      }
      if (!inner) plist[atomicAdd(&counts[0], 1)] = (int)i;
    }
    const int g = gt[i];
    if (k == 1 ? g == 1 : g >= 1) {
      bool inner = true;
      for (int q = 0; q < 4; ++q) {
        const int y2 = yy + dy[q], x2 = xx + dx[q];
        bool on = y2 >= 0 && y2 < H && x2 >= 0 && x2 < W;
        if (on) {
          const int g2 = gt[(long)y2 * W + x2];
          on = k == 1 ? g2 == 1 : g2 >= 1;
        }
        inner = inner && on;
      }
      if (!inner) glist[atomicAdd(&counts[1], 1)] = (int)i;
    }
  }
}
// one workgroup per CHUNK of 256 source pixels; the other list streams through LDS in tiles of 1024
__global__ __launch_bounds__(256) void surface_dist_kernel(const int* __restrict__ a, const int* __restrict__ b,
                                                           const int* counts, int ia, int ib, int W,
                                                           double* __restrict__ out) {
  __shared__ int bx[1024], by[1024];
  const int na = counts[ia], nb = counts[ib];
  for (int base = blockIdx.x * 256; base < na; base += gridDim.x * 256) {
    const int i = base + threadIdx.x;
    int ax = 0, ay = 0;
    if (i < na) {
      ax = a[i] % W;
      ay = a[i] / W;
    }
    long best = 0x7fffffffffffffffL;
    for (int t0 = 0; t0 < nb; t0 += 1024) {
      __syncthreads();
      for (int t = threadIdx.x; t < 1024; t += 256) {
        const int j = t0 + t;
        const int v = j < nb ? b[j] : b[0];
        bx[t] = v % W;
        by[t] = v / W;
      }
      __syncthreads();
      const int lim = min(1024, nb - t0);
      for (int t = 0; t < lim; ++t) {
        const long ddx = ax - bx[t], ddy = ay - by[t];
        const long d2 = ddx * ddx + ddy * ddy;
        best = d2 < best ? d2 : best;
      }
    }
    if (i < na) out[i] = nb > 0 ? sqrt((double)best) : 0.0;
  }
}

extern "C" int fi_seg_borders(const float* logits, const uint8_t* gt, int H, int W, int C, int k, int* pred_list,
                              int* gt_list, int* counts, void* stream) {
  if (!logits || !gt || !pred_list || !gt_list || !counts) return FI_ERR_NULL;
  if (C < 2 || C > FI_MAX_CLASSES || k < 1 || k >= C || H < 1 || W < 1) return FI_ERR_SHAPE;
  hipLaunchKernelGGL(seg_borders_kernel, dim3(grid_for((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream, logits, gt, H,
                     W, C, k, pred_list, gt_list, counts);
  FI_CHECK_LAUNCH();
  return 0;
}
extern "C" int fi_surface_distances(const int* from_list, const int* to_list, const int* counts, int from_index,
                                    int to_index, int W, int max_from, double* out, void* stream) {
  if (!from_list || !to_list || !counts || !out) return FI_ERR_NULL;
  if (W < 1 || max_from < 1 || (from_index | to_index) & ~1) return FI_ERR_SHAPE;
  hipLaunchKernelGGL(surface_dist_kernel, dim3(grid_for(max_from, 256)), dim3(256), 0, (hipStream_t)stream, from_list,
                     to_list, counts, from_index, to_index, W, out);
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// AdamW
// ------------------------------------------------------------------------------------------------
__global__ void adamw_hyper_kernel(int* step, float* hyper, const double* lr_state, float beta1, float beta2,
                                   float wd) {
  const int t = step[0] + 1;
  step[0] = t;
  const double lr = lr_state[0];
  const double bc1 = 1.0 - pow((double)beta1, (double)t);
  const double bc2 = 1.0 - pow((double)beta2, (double)t);
  hyper[0] = (float)lr;
  hyper[1] = (float)(1.0 - lr * (double)wd);
  hyper[2] = (float)(lr / bc1);
  hyper[3] = (float)sqrt(bc2);
}

__global__ void lr_poly_kernel(int* iter, double* lr_state, double base_lr, double max_iter) {
  const int it = iter[0] + 1;
  iter[0] = it;
  lr_state[0] = base_lr * pow(1.0 - (double)it / max_iter, 0.9);
}

extern "C" int fi_adamw_hyper(int* step, float* hyper, const double* lr_state, float beta1, float beta2, float wd,
                              void* stream) {
  if (!step || !hyper || !lr_state) return FI_ERR_NULL;
  hipLaunchKernelGGL(adamw_hyper_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, hyper, lr_state, beta1,
                     beta2, wd);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_lr_poly_advance(int* iter, double* lr_state, double base_lr, double max_iter, void* stream) {
  if (!iter || !lr_state) return FI_ERR_NULL;
  hipLaunchKernelGGL(lr_poly_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, iter, lr_state, base_lr, max_iter);
  FI_CHECK_LAUNCH();
  return 0;
}

__global__ __launch_bounds__(256) void adamw_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, long n,
                                                         const float* __restrict__ hyper, float beta1, float beta2,
                                                         float eps, bf16_t* __restrict__ shadow) {
  if (hyper[0] < 0.f) return;   // fi_amp_guard: the unscaled gradients held an inf/NaN -> this step is skipped
  const float decay = hyper[1], step = hyper[2], bc2s = hyper[3];
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i] * decay;                       // param.mul_(1 - lr*wd)
    const float mi = m[i] + omb1 * (gi - m[i]);    // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * beta2 + (omb2 * gi) * gi;  // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(vi) / bc2s + eps;
    pi = pi + (-step * mi) / denom;                // param.addcdiv_(exp_avg, denom, -step_size)
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (shadow) shadow[i] = (bf16_t)pi;
  }
}

extern "C" int fi_adamw_step(float* p, const float* g, float* m, float* v, long n, const float* hyper, float beta1,
                             float beta2, float eps, void* shadow_bf16, void* stream) {
  if (!p || !g || !m || !v || !hyper) return FI_ERR_NULL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(adamw_step_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                     hyper, beta1, beta2, eps, (bf16_t*)shadow_bf16);
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// SGD with momentum and (coupled) weight decay: the single-site trainer's optimizer
// (/root/reference/code/Unet_pCE.py:88-89: SGD(lr, momentum=0.9, weight_decay=1e-4), dampening 0, no nesterov)
//   g' = g + wd*p;  buf = momentum*buf + g'  (buf starts at 0, which reproduces torch's first-step buf = g');  p -= lr*buf
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ buf, long n, const double* lr_state,
                                                       float momentum, float wd, const float* hyper) {
  if (hyper && hyper[0] < 0.f) return;          // fi_amp_guard convention: skipped step
  const float lr = (float)lr_state[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] + wd * p[i];
    const float b = momentum * buf[i] + gi;
    buf[i] = b;
    p[i] = p[i] - lr * b;
  }
}
extern "C" int fi_sgd_step(float* p, const float* g, float* momentum_buf, long n, const double* lr_state, float momentum,
                           float weight_decay, const float* skip_hyper, void* stream) {
  if (!p || !g || !momentum_buf || !lr_state) return FI_ERR_NULL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sgd_step_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, p, g, momentum_buf, n,
                     lr_state, momentum, weight_decay, skip_hyper);
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// dynamic loss scaling (torch.cuda.amp.GradScaler semantics; flower_pCE_2D.py:47-48,143-146), all state on the device
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void amp_unscale_kernel(float* __restrict__ g, long n, const float* __restrict__ scale,
                                                          float* found_inf) {
  const float inv = 1.0f / scale[0];
  bool bad = false;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = g[i] * inv;
    bad |= !isfinite(v);
    g[i] = v;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[0] = 1.0f;     // every writer stores the same value
}
__global__ void amp_guard_kernel(int* step, float* hyper, const float* found_inf) {
  if (found_inf[0] != 0.f) {
    step[0] -= 1;            // GradScaler.step() does not call optimizer.step(): the step count does not advance
    hyper[0] = -1.f;         // makes fi_adamw_step a no-op
  }
}
__global__ void amp_update_kernel(float* scale, int* tracker, float* found_inf, float growth, float backoff,
                                  int interval) {
  if (found_inf[0] != 0.f) {
    scale[0] *= backoff;
    tracker[0] = 0;
  } else {
    const int t = tracker[0] + 1;
    if (t == interval) {
      scale[0] *= growth;
      tracker[0] = 0;
    } else {
      tracker[0] = t;
    }
  }
  found_inf[0] = 0.f;
}

extern "C" int fi_amp_unscale(float* grads, long n, const float* scale, float* found_inf, void* stream) {
  if (!grads || !scale || !found_inf) return FI_ERR_NULL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(amp_unscale_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, grads, n, scale,
                     found_inf);
  FI_CHECK_LAUNCH();
  return 0;
}
extern "C" int fi_amp_guard(int* step, float* hyper, const float* found_inf, void* stream) {
  if (!step || !hyper || !found_inf) return FI_ERR_NULL;
  hipLaunchKernelGGL(amp_guard_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, hyper, found_inf);
  FI_CHECK_LAUNCH();
  return 0;
}
extern "C" int fi_amp_update(float* scale, int* growth_tracker, float* found_inf, float growth_factor,
                             float backoff_factor, int growth_interval, void* stream) {
  if (!scale || !growth_tracker || !found_inf) return FI_ERR_NULL;
  hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scale, growth_tracker, found_inf,
                     growth_factor, backoff_factor, growth_interval);
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// aggregation helpers
// ------------------------------------------------------------------------------------------------
__global__ void scale_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float a, int divide) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = divide ? __fdiv_rn(x[i], a) : __fmul_rn(x[i], a);
}
extern "C" int fi_scale(const float* x, float* y, long n, float a, int divide, void* stream) {
  if (!x || !y) return FI_ERR_NULL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, x, y, n, a, divide);
  FI_CHECK_LAUNCH();
  return 0;
}

__global__ void axpy_kernel(float* __restrict__ acc, const float* __restrict__ x, long n, float a) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    acc[i] = acc[i] + x[i] * a;     // plain operators: the pragma does not reach into the __f*_rn header inlines
}
extern "C" int fi_axpy(float* acc, const float* x, long n, float a, void* stream) {
  if (!acc || !x) return FI_ERR_NULL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, acc, x, n, a);
  FI_CHECK_LAUNCH();
  return 0;
}

// flwr 1.0.0 FedOpt server optimizers (flwr/server/strategy/{fedadagrad,fedadam,fedyogi}.py, restated from the published
// source -- flwr is absent: parity unpinned) on the flat fp32 state, every product / sum rounded to fp32 like numpy does:
//   delta = agg - cur;  m = b1*m + (1-b1)*delta;  v: adagrad v + delta^2 | adam b2*v + (1-b2)*delta^2 |
//   yogi v - (1-b2)*delta^2*sign(v - delta^2);  cur = cur + eta*m / (sqrt(v) + tau)
__global__ void fedopt_step_kernel(int mode, float* __restrict__ cur, const float* __restrict__ agg, float* __restrict__ m,
                                   float* __restrict__ v, long n, float eta, float b1, float omb1, float b2, float omb2,
                                   float tau) {
#pragma clang fp contract(off)   // numpy rounds every product and sum: no fused multiply-add (plain operators below -- the
                                 // pragma does not reach into the __f*_rn header inlines, which DO get contracted)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = agg[i] - cur[i];
    const float mi = b1 * m[i] + omb1 * d;
    const float d2 = d * d;
    float vi = v[i];
    if (mode == 0) {
      vi = vi + d2;
    } else if (mode == 1) {
      vi = b2 * vi + omb2 * d2;
    } else {
      const float df = vi - d2;
      const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : df);          // np.sign: 0 -> 0, nan -> nan
      vi = vi - (omb2 * d2) * sg;
    }
    m[i] = mi;
    v[i] = vi;
    cur[i] = cur[i] + (eta * mi) / (sqrtf(vi) + tau);
  }
}
extern "C" int fi_fedopt_step(int mode, float* cur, const float* agg, float* m, float* v, long n, float eta, float beta1,
                              float one_minus_beta1, float beta2, float one_minus_beta2, float tau, void* stream) {
  if (!cur || !agg || !m || !v) return FI_ERR_NULL;
  if (mode < 0 || mode > 2) return FI_ERR_UNSUPPORTED;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(fedopt_step_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, mode, cur, agg, m, v,
                     n, eta, beta1, one_minus_beta1, beta2, one_minus_beta2, tau);
  FI_CHECK_LAUNCH();
  return 0;
}

__global__ void ala_update_kernel(float* __restrict__ w, float* __restrict__ temp, const float* __restrict__ grad,
                                  const float* __restrict__ local, const float* __restrict__ global, long n,
                                  float eta, const float* __restrict__ skip) {
  if (skip && skip[0] != 0.f) return;            // the batch's gradients overflowed (amp): nothing moves
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = local[i] - global[i];
    float wi = w[i] - eta * (grad[i] * d);
    wi = fminf(fmaxf(wi, 0.f), 1.f);
    w[i] = wi;
    temp[i] = global[i] + d * wi;
  }
}
extern "C" int fi_ala_update(float* w, float* temp, const float* grad, const float* local, const float* global,
                             long n, float eta, const float* skip, void* stream) {
  if (!w || !temp || !grad || !local || !global) return FI_ERR_NULL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(ala_update_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, w, temp, grad,
                     local, global, n, eta, skip);
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// PCS helpers: global avg / max pool, channel gate
// ------------------------------------------------------------------------------------------------
// grid (N, ceil(C/64)); block 256 = 4 pixel lanes x 64 channels
template <typename T>
__global__ __launch_bounds__(256) void global_avgmax_kernel(const T* __restrict__ x, float* __restrict__ avg,
                                                            float* __restrict__ mx, int* __restrict__ amax, int HW,
                                                            int C) {
  const int n = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  float s = 0.f, m = -INFINITY;
  int mi = 0;
  if (c < C)
    for (int p = pl; p < HW; p += 4) {
      const float v = to_f32(x[((size_t)n * HW + p) * C + c]);
      s += v;
      if (v > m) {
        m = v;
        mi = p;
      }
    }
  __shared__ float ss[4][64], sm[4][64];
  __shared__ int si[4][64];
  ss[pl][threadIdx.x & 63] = s;
  sm[pl][threadIdx.x & 63] = m;
  si[pl][threadIdx.x & 63] = mi;
  __syncthreads();
  if (pl == 0 && c < C) {
    const int l = threadIdx.x & 63;
    float ts = 0.f, tm = -INFINITY;
    int ti = 0;
    for (int q = 0; q < 4; ++q) {
      ts += ss[q][l];
      // first occurrence in scan order: strict > on value, lower index wins ties
      if (sm[q][l] > tm || (sm[q][l] == tm && si[q][l] < ti)) {
        tm = sm[q][l];
        ti = si[q][l];
      }
    }
    avg[(size_t)n * C + c] = ts / (float)HW;
    mx[(size_t)n * C + c] = tm;
    if (amax) amax[(size_t)n * C + c] = ti;
  }
}

extern "C" int fi_global_avgmax(int dtype, const void* x, float* avg, float* mx, int* amax, int N, int HW, int C,
                                void* stream) {
  if (!x || !avg || !mx) return FI_ERR_NULL;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(N, fi_cdiv(C, 64)), b(256);
  if (dtype == FI_F32)
    hipLaunchKernelGGL(global_avgmax_kernel<float>, g, b, 0, st, (const float*)x, avg, mx, amax, HW, C);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(global_avgmax_kernel<bf16_t>, g, b, 0, st, (const bf16_t*)x, avg, mx, amax, HW, C);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(global_avgmax_kernel<f16_t>, g, b, 0, st, (const f16_t*)x, avg, mx, amax, HW, C);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}

// Split form for the full-resolution maps (PCS pools 12 or 84 images of 512 x 512 x 16: the (N, C/64) grid above is 12-84
// workgroups with a quarter of their lanes busy, 1.3 TB/s).  Stage 1: grid (S, N), a workgroup reduces one pixel range of one
// image with 16-byte loads (thread = pixel lane x channel vector), lanes combined through LDS in lane order; its partial
// (sum, max, first arg-max) goes to workspace[n][s][3][C].  Stage 2 folds the S partials of an image in range order.
// Every order is fixed: deterministic; ties keep the lowest pixel index (first occurrence in scan order), as above.
template <typename T>
__global__ __launch_bounds__(256) void global_avgmax_part_kernel(const T* __restrict__ x, float* __restrict__ part, int HW,
                                                                 int C, int S) {
  constexpr int VG = DT<T>::VG;
  typedef typename DT<T>::vec_t vec_t;
  extern __shared__ float sh[];                 // [PL][C] sums, [PL][C] maxima, [PL][C] indices (as int)
  const int CV = C / VG, PL = 256 / CV;         // pixel lanes per workgroup (CV divides 256: host)
  const int s = blockIdx.x, n = blockIdx.y;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  const int chunk = (HW + S - 1) / S, p0 = s * chunk, p1 = min(HW, p0 + chunk);
  float sum[VG], mx[VG];
  int mi[VG];
#pragma unroll
  for (int j = 0; j < VG; ++j) sum[j] = 0.f, mx[j] = -INFINITY, mi[j] = 0;
  const T* const base = x + (size_t)n * HW * C + cv * VG;
  for (int p = p0 + pl; p < p1; p += PL) {
    float v[VG];
    VecWords<T>::unpack(*reinterpret_cast<const vec_t*>(base + (size_t)p * C), v);
#pragma unroll
    for (int j = 0; j < VG; ++j) {
      sum[j] += v[j];
      if (v[j] > mx[j]) mx[j] = v[j], mi[j] = p;
    }
  }
  float* ss = sh;
  float* sm = sh + PL * C;
  int* si = reinterpret_cast<int*>(sh + 2 * PL * C);
#pragma unroll
  for (int j = 0; j < VG; ++j) {
    ss[pl * C + cv * VG + j] = sum[j];
    sm[pl * C + cv * VG + j] = mx[j];
    si[pl * C + cv * VG + j] = mi[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float ts = 0.f, tm = -INFINITY;
    int ti = 0;
    for (int q = 0; q < PL; ++q) {
      ts += ss[q * C + c];
      const float m = sm[q * C + c];
      const int i = si[q * C + c];
      if (m > tm || (m == tm && i < ti)) tm = m, ti = i;
    }
    float* dst = part + ((size_t)n * S + s) * 3 * C;
    dst[c] = ts;
    dst[C + c] = tm;
    reinterpret_cast<int*>(dst)[2 * C + c] = ti;
  }
}

static __global__ __launch_bounds__(256) void global_avgmax_fold_kernel(const float* __restrict__ part, float* __restrict__ avg,
                                                                        float* __restrict__ mx, int* __restrict__ amax, int HW,
                                                                        int C, int S) {
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float ts = 0.f, tm = -INFINITY;
    int ti = 0;
    for (int s = 0; s < S; ++s) {
      const float* src = part + ((size_t)n * S + s) * 3 * C;
      ts += src[c];
      const float m = src[C + c];
      const int i = reinterpret_cast<const int*>(src)[2 * C + c];
      if (m > tm || (m == tm && i < ti)) tm = m, ti = i;
    }
    avg[(size_t)n * C + c] = ts / (float)HW;
    mx[(size_t)n * C + c] = tm;
    if (amax) amax[(size_t)n * C + c] = ti;
  }
}

// pixel ranges per image the split form wants for this shape (0: use fi_global_avgmax); workspace = N * S * 3 * C floats
extern "C" int fi_global_avgmax_ranges(int dtype, int N, int HW, int C) {
  const int vg = dtype == FI_F32 ? 4 : 8;
  if (N < 1 || HW < 1 || C < vg || C % vg || 256 % (C / vg) || C > 256) return 0;
  if ((long)HW * C < (1L << 17)) return 0;                   // small maps: one workgroup per (image, 64 channels) is enough
  long S = 2048 / N;
  if (S > HW / 128) S = HW / 128;
  return S < 2 ? 0 : (int)S;
}

extern "C" int fi_global_avgmax_split(int dtype, const void* x, float* avg, float* mx, int* amax, int N, int HW, int C,
                                      float* workspace, long workspace_bytes, void* stream) {
  if (!x || !avg || !mx || !workspace) return FI_ERR_NULL;
  const int S = fi_global_avgmax_ranges(dtype, N, HW, C);
  if (S < 2) return FI_ERR_UNSUPPORTED;
  if (workspace_bytes < (long)N * S * 3 * C * (long)sizeof(float)) return FI_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int vg = dtype == FI_F32 ? 4 : 8, PL = 256 / (C / vg);
  const size_t lds = (size_t)3 * PL * C * sizeof(float);
  const dim3 g(S, N), b(256);
  if (dtype == FI_F32)
    hipLaunchKernelGGL(global_avgmax_part_kernel<float>, g, b, lds, st, (const float*)x, workspace, HW, C, S);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(global_avgmax_part_kernel<bf16_t>, g, b, lds, st, (const bf16_t*)x, workspace, HW, C, S);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(global_avgmax_part_kernel<f16_t>, g, b, lds, st, (const f16_t*)x, workspace, HW, C, S);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  hipLaunchKernelGGL(global_avgmax_fold_kernel, dim3(N), dim3(256), 0, st, workspace, avg, mx, amax, HW, C, S);
  FI_CHECK_LAUNCH();
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void channel_gate_fwd_kernel(const T* __restrict__ x, const float* __restrict__ h,
                                                               T* __restrict__ y, long n_elem, int HW, int C) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_elem; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long n = i / ((long)C * HW);
    const float xv = to_f32(x[i]);
    y[i] = from_f32<T>(xv * h[n * C + c] + xv);
  }
}

extern "C" int fi_channel_gate_fwd(int dtype, const void* x, const float* h, void* y, int N, int HW, int C,
                                   void* stream) {
  if (!x || !h || !y) return FI_ERR_NULL;
  const long n = (long)N * HW * C;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FI_F32)
    hipLaunchKernelGGL(channel_gate_fwd_kernel<float>, dim3(grid_for(n, 256 * 4)), dim3(256), 0, st, (const float*)x,
                       h, (float*)y, n, HW, C);
  else if (dtype == FI_BF16)
    hipLaunchKernelGGL(channel_gate_fwd_kernel<bf16_t>, dim3(grid_for(n, 256 * 4)), dim3(256), 0, st,
                       (const bf16_t*)x, h, (bf16_t*)y, n, HW, C);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(channel_gate_fwd_kernel<f16_t>, dim3(grid_for(n, 256 * 4)), dim3(256), 0, st,
                       (const f16_t*)x, h, (f16_t*)y, n, HW, C);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}

// grid (N, ceil(C/64)); block = PLN pixel lanes x 64 channels (PLN = 16 for maps of 256+ pixels: the grid is only
// N * C/64 workgroups, the lanes are where the parallelism is).  dh by LDS reduction over the pixel lanes, in lane order.
template <typename T, int PLN>
__global__ __launch_bounds__(64 * PLN) void channel_gate_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                               const float* __restrict__ h,
                                                               const int* __restrict__ amax,
                                                               const float* __restrict__ davg,
                                                               const float* __restrict__ dmx, T* __restrict__ dx,
                                                               float* __restrict__ dh, int HW, int C) {
  const int n = blockIdx.x, l = threadIdx.x & 63, c = blockIdx.y * 64 + l, pl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C) {
    const float gate = 1.f + h[(size_t)n * C + c];
    const float ga = davg ? davg[(size_t)n * C + c] / (float)HW : 0.f;
    const float gm = dmx ? dmx[(size_t)n * C + c] : 0.f;
    const int am = amax ? amax[(size_t)n * C + c] : -1;
    for (int p = pl; p < HW; p += PLN) {
      const size_t o = ((size_t)n * HW + p) * C + c;
      const float g = to_f32(dy[o]);
      s += g * to_f32(x[o]);
      float v = g * gate + ga;
      if (p == am) v += gm;
      dx[o] = from_f32<T>(v);
    }
  }
  __shared__ float ss[PLN][64];
  ss[pl][l] = s;
  __syncthreads();
  if (pl == 0 && c < C && dh) {
    float t = ss[0][l];
#pragma unroll
    for (int q = 1; q < PLN; ++q) t += ss[q][l];
    dh[(size_t)n * C + c] = t;
  }
}

extern "C" int fi_channel_gate_bwd(int dtype, const void* x, const void* dy, const float* h, const int* amax,
                                   const float* davg, const float* dmx, void* dx, float* dh, int N, int HW, int C,
                                   void* stream) {
  if (!x || !dy || !h || !dx) return FI_ERR_NULL;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(N, fi_cdiv(C, 64));
  const bool wide = HW >= 256;
  const dim3 b(wide ? 1024 : 256);
#define FI_GATE_BWD(T_)                                                                                              \
  do {                                                                                                             \
    if (wide)                                                                                                      \
      hipLaunchKernelGGL((channel_gate_bwd_kernel<T_, 16>), g, b, 0, st, (const T_*)x, (const T_*)dy, h, amax, davg, \
                         dmx, (T_*)dx, dh, HW, C);                                                                 \
    else                                                                                                           \
      hipLaunchKernelGGL((channel_gate_bwd_kernel<T_, 4>), g, b, 0, st, (const T_*)x, (const T_*)dy, h, amax, davg,  \
                         dmx, (T_*)dx, dh, HW, C);                                                                 \
  } while (0)
  if (dtype == FI_F32)
    FI_GATE_BWD(float);
  else if (dtype == FI_BF16)
    FI_GATE_BWD(bf16_t);
  else if (dtype == FI_F16)
    FI_GATE_BWD(f16_t);
  else
    return FI_ERR_DTYPE;
#undef FI_GATE_BWD
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// hardware probe: ds_read_b64_tr_b16 (gfx950 LDS transpose read) with caller-chosen per-lane addresses
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short probe_s16x4;
__global__ void probe_tr16_kernel(const short* in, const int* offs, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  const int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = in[i];
  __syncthreads();
  const probe_s16x4 v =
      __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) probe_s16x4*)(lds + offs[l]));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
extern "C" int fi_probe_tr16(const short* in, const int* offs, short* out, void* stream) {
  if (!in || !offs || !out) return FI_ERR_NULL;
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, offs, out);
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// PersonalizedChannelSelection's gate as ONE launch per direction (/root/reference/code/networks/unet.py:103-144):
//   e  = W1b relu(W1a onehot(who))                  fc1: 1x1 convs K -> C -> C, no bias
//   a  = W2b relu(W2a [avg ; e]),  m = W2b relu(W2a [mx ; e])      fc2: 2C -> C/16 -> C, shared, no bias
//   h  = sigmoid(a + m)
// One workgroup per image; everything fp32 (a few hundred kFLOP).  The reference evaluates these as 1x1 convolutions of
// [B,C,1,1] maps: each output is the dot product taken in channel order, which is the order of the loops here.  The
// hidden activations (C/16 each) are kept for the backward pass; the PCS weights are frozen in the reference (they are not
// registered, unet.py:172-177), so backward only produces d/d avg and d/d max.
// ------------------------------------------------------------------------------------------------
#define FI_PCS_MAXC 512
__global__ __launch_bounds__(256) void pcs_gate_fwd_kernel(const float* __restrict__ avg, const float* __restrict__ mx,
                                                           const int* __restrict__ who, const float* __restrict__ w1a,
                                                           const float* __restrict__ w1b, const float* __restrict__ w2a,
                                                           const float* __restrict__ w2b, float* __restrict__ h,
                                                           float* __restrict__ hidden, int C, int K) {
  __shared__ float v[3][FI_PCS_MAXC];         // avg, max, e (first as e1)
  __shared__ float e1[FI_PCS_MAXC];
  __shared__ float t[2][FI_PCS_MAXC / 16];
  const int b = blockIdx.x, tid = threadIdx.x, R = C / 16;
  const int k = who[b];
  for (int c = tid; c < C; c += 256) {
    v[0][c] = avg[(size_t)b * C + c];
    v[1][c] = mx[(size_t)b * C + c];
    const float x = w1a[(size_t)c * K + k];                    // column `who` of fc1[0]: the one-hot embedding picks it
    e1[c] = x > 0.f ? x : 0.f;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    const float* row = w1b + (size_t)c * C;
    for (int j = 0; j < C; ++j) s += row[j] * e1[j];
    v[2][c] = s;
  }
  __syncthreads();
  // hidden layer: 2 x R dot products of length 2C, one wave-quarter (16 lanes) each would idle most lanes; a thread each
  for (int i = tid; i < 2 * R; i += 256) {
    const int which = i / R, r = i % R;
    const float* row = w2a + (size_t)r * 2 * C;
    float s = 0.f;
    for (int j = 0; j < C; ++j) s += row[j] * v[which][j];
    for (int j = 0; j < C; ++j) s += row[C + j] * v[2][j];
    t[which][r] = s > 0.f ? s : 0.f;
    hidden[((size_t)b * 2 + which) * R + r] = t[which][r];
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const float* row = w2b + (size_t)c * R;
    float a = 0.f, m = 0.f;
    for (int r = 0; r < R; ++r) {
      a += row[r] * t[0][r];
      m += row[r] * t[1][r];
    }
    h[(size_t)b * C + c] = 1.f / (1.f + expf(-(a + m)));
  }
}

__global__ __launch_bounds__(256) void pcs_gate_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                           const float* __restrict__ hidden, const float* __restrict__ w2a,
                                                           const float* __restrict__ w2b, float* __restrict__ davg,
                                                           float* __restrict__ dmx, int C) {
  __shared__ float ds[FI_PCS_MAXC];
  __shared__ float dt[2][FI_PCS_MAXC / 16];
  const int b = blockIdx.x, tid = threadIdx.x, R = C / 16;
  for (int c = tid; c < C; c += 256) {
    const float hv = h[(size_t)b * C + c];
    ds[c] = dh[(size_t)b * C + c] * hv * (1.f - hv);            // through the sigmoid; a and m share it
  }
  __syncthreads();
  for (int i = tid; i < 2 * R; i += 256) {
    const int which = i / R, r = i % R;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += w2b[(size_t)c * R + r] * ds[c];
    dt[which][r] = hidden[((size_t)b * 2 + which) * R + r] > 0.f ? s : 0.f;     // through the ReLU
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, m = 0.f;
    for (int r = 0; r < R; ++r) {
      const float wv = w2a[(size_t)r * 2 * C + c];
      a += wv * dt[0][r];
      m += wv * dt[1][r];
    }
    davg[(size_t)b * C + c] = a;
    dmx[(size_t)b * C + c] = m;
  }
}

extern "C" int fi_pcs_gate_fwd(const float* avg, const float* mx, const int* who, const float* w1a, const float* w1b,
                               const float* w2a, const float* w2b, float* h, float* hidden, int B, int C, int K,
                               void* stream) {
  if (!avg || !mx || !who || !w1a || !w1b || !w2a || !w2b || !h || !hidden) return FI_ERR_NULL;
  if (B < 1 || C < 16 || C % 16 || C > FI_PCS_MAXC || K < 1) return FI_ERR_SHAPE;
  hipLaunchKernelGGL(pcs_gate_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, avg, mx, who, w1a, w1b, w2a, w2b, h,
                     hidden, C, K);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_pcs_gate_bwd(const float* dh, const float* h, const float* hidden, const float* w2a, const float* w2b,
                               float* davg, float* dmx, int B, int C, void* stream) {
  if (!dh || !h || !hidden || !w2a || !w2b || !davg || !dmx) return FI_ERR_NULL;
  if (B < 1 || C < 16 || C % 16 || C > FI_PCS_MAXC) return FI_ERR_SHAPE;
  hipLaunchKernelGGL(pcs_gate_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dh, h, hidden, w2a, w2b, davg, dmx, C);
  FI_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// FedICRA's LC loss (/root/reference/code/flower_pCE_2D.py:128-139) and its place in the total, one launch per direction:
//   loss_lc = -(1 / G) sum_g mean((h - o_g)^2)        h: own heat-map [n], o: the G other clients' [G][n]
//   total   = loss_ce + alpha * loss_lc
// out = {total, loss_lc}; dcoef[i] = d loss_lc / d h[i] = -(2 / (G n)) sum_g (h[i] - o_g[i]) is kept for backward, which is
// dh = g * alpha * dcoef (g = the upstream gradient of `total`, a device scalar).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void lc_loss_fwd_kernel(const float* __restrict__ h, const float* __restrict__ others,
                                                           const float* __restrict__ loss_ce, float alpha, int G, int n,
                                                           float* __restrict__ out, float* __restrict__ dcoef) {
  __shared__ double part[16];
  double acc = 0.0;
  const float inv = 1.f / ((float)G * (float)n);
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float hv = h[i];
    float d1 = 0.f;
    for (int g = 0; g < G; ++g) {
      const float d = hv - others[(size_t)g * n + i];
      acc += (double)(d * d);
      d1 += d;
    }
    dcoef[i] = -2.f * inv * d1;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += part[w];
    const float lc = -(float)(s / ((double)G * (double)n));
    out[1] = lc;
    out[0] = loss_ce[0] + alpha * lc;
  }
}

__global__ __launch_bounds__(256) void scale_by_device_scalar_kernel(const float* __restrict__ src, const float* __restrict__ g,
                                                                     float k, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i] * (g[0] * k);
}

extern "C" int fi_lc_loss_fwd(const float* h, const float* others, const float* loss_ce, float alpha, int G, int n,
                              float* out, float* dcoef, void* stream) {
  if (!h || !others || !loss_ce || !out || !dcoef) return FI_ERR_NULL;
  if (G < 1 || n < 1) return FI_ERR_SHAPE;
  hipLaunchKernelGGL(lc_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, h, others, loss_ce, alpha, G, n, out, dcoef);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_lc_loss_bwd(const float* dcoef, const float* g, float alpha, float* dh, int n, void* stream) {
  if (!dcoef || !g || !dh) return FI_ERR_NULL;
  if (n < 1) return FI_ERR_SHAPE;
  hipLaunchKernelGGL(scale_by_device_scalar_kernel, dim3(fi_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dcoef, g, alpha, dh, n);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_abi_version(void) { return FI_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------
// (partial) Dice loss: pDLoss / DiceLoss of /root/reference/code/utils/losses.py:156-232.
//   probs fp32 NHWC [B][HW][C] (softmax already applied), labels uint8 [B][HW].
//   Reference quirk reproduced (pDLoss): the ignore mask keeps shape [B,1,H,W] while score/target planes are
//   [B,H,W], so score*target*mask broadcasts to [B,B,H,W]:  sum_{b,b'} s[b'] t[b'] m[b]  =
//   sum_hw M[hw] * sum_b' s[b'][hw] t[b'][hw]  with  M[hw] = sum_b m[b][hw].   ignore_index < 0: plain DiceLoss
//   (M = 1).  acc (fp64 [C][3]) += {intersect, z_sum = sum s^2, y_sum = sum t^2}; caller zeroes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pdice_fwd_kernel(const float* __restrict__ probs,
                                                        const uint8_t* __restrict__ labels, int B, long HW, int C,
                                                        int ignore, double* acc) {
  __shared__ double sm[4];
  double I[FI_MAX_CLASSES], Z[FI_MAX_CLASSES], Y[FI_MAX_CLASSES];
  for (int c = 0; c < FI_MAX_CLASSES; ++c) I[c] = Z[c] = Y[c] = 0.0;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
    float M = 1.f;
    if (ignore >= 0) {
      M = 0.f;
      for (int b = 0; b < B; ++b) M += (labels[(size_t)b * HW + p] != ignore) ? 1.f : 0.f;
    }
    for (int b = 0; b < B; ++b) {
      const int lb = labels[(size_t)b * HW + p];
      const float* s = probs + ((size_t)b * HW + p) * C;
      for (int c = 0; c < C; ++c) {
        const float sv = s[c], tv = (lb == c) ? 1.f : 0.f;
        I[c] += (double)(sv * tv * M);
        Z[c] += (double)(sv * sv * M);
        Y[c] += (double)(tv * tv * M);
      }
    }
  }
  for (int c = 0; c < C; ++c) {
    const double i_ = block_sum(I[c], sm), z_ = block_sum(Z[c], sm), y_ = block_sum(Y[c], sm);
    if (threadIdx.x == 0) {
      atomicAdd(&acc[c * 3 + 0], i_);
      atomicAdd(&acc[c * 3 + 1], z_);
      atomicAdd(&acc[c * 3 + 2], y_);
    }
  }
}

// loss = sum_c w_c * (1 - (2 I_c + 1e-5) / (Z_c + Y_c + 1e-5)) / C     (losses.py:170-192, 209-232)
__global__ void pdice_finalize_kernel(const double* acc, const float* weight, int C, float* loss) {
  double l = 0.0;
  for (int c = 0; c < C; ++c) {
    const double d = 1.0 - (2.0 * acc[c * 3] + 1e-5) / (acc[c * 3 + 1] + acc[c * 3 + 2] + 1e-5);
    l += d * (weight ? (double)weight[c] : 1.0);
  }
  loss[0] = (float)(l / C);
}

// d loss / d s[b][hw][c] = g * w_c / C * M[hw] * ( -2 t / D_c + 2 s (2 I_c + eps) / D_c^2 ),  D_c = Z_c + Y_c + eps
__global__ __launch_bounds__(256) void pdice_bwd_kernel(const float* __restrict__ probs,
                                                        const uint8_t* __restrict__ labels, int B, long HW, int C,
                                                        int ignore, const double* __restrict__ acc,
                                                        const float* weight, const float* gscale,
                                                        float* __restrict__ dprobs) {
  float k1[FI_MAX_CLASSES], k2[FI_MAX_CLASSES];
  const float g = gscale ? gscale[0] : 1.f;
  for (int c = 0; c < C; ++c) {
    const double D = acc[c * 3 + 1] + acc[c * 3 + 2] + 1e-5;
    const double w = (weight ? (double)weight[c] : 1.0) * (double)g / C;
    k1[c] = (float)(-2.0 * w / D);
    k2[c] = (float)(2.0 * w * (2.0 * acc[c * 3] + 1e-5) / (D * D));
  }
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
    float M = 1.f;
    if (ignore >= 0) {
      M = 0.f;
      for (int b = 0; b < B; ++b) M += (labels[(size_t)b * HW + p] != ignore) ? 1.f : 0.f;
    }
    for (int b = 0; b < B; ++b) {
      const int lb = labels[(size_t)b * HW + p];
      const float* s = probs + ((size_t)b * HW + p) * C;
      float* o = dprobs + ((size_t)b * HW + p) * C;
      for (int c = 0; c < C; ++c) o[c] = M * (k1[c] * ((lb == c) ? 1.f : 0.f) + k2[c] * s[c]);
    }
  }
}

extern "C" int fi_pdice_fwd(const float* probs, const uint8_t* labels, int B, long HW, int C, int ignore_index,
                            double* acc, void* stream) {
  if (!probs || !labels || !acc) return FI_ERR_NULL;
  if (C < 1 || C > FI_MAX_CLASSES || B < 1) return FI_ERR_SHAPE;
  hipLaunchKernelGGL(pdice_fwd_kernel, dim3(grid_for(HW, 256)), dim3(256), 0, (hipStream_t)stream, probs, labels, B,
                     HW, C, ignore_index, acc);
  FI_CHECK_LAUNCH();
  return 0;
}
extern "C" int fi_pdice_finalize(const double* acc, const float* weight, int C, float* loss, void* stream) {
  if (!acc || !loss) return FI_ERR_NULL;
  if (C < 1 || C > FI_MAX_CLASSES) return FI_ERR_SHAPE;
  hipLaunchKernelGGL(pdice_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, weight, C, loss);
  FI_CHECK_LAUNCH();
  return 0;
}
extern "C" int fi_pdice_bwd(const float* probs, const uint8_t* labels, int B, long HW, int C, int ignore_index,
                            const double* acc, const float* weight, const float* gscale, float* dprobs,
                            void* stream) {
  if (!probs || !labels || !acc || !dprobs) return FI_ERR_NULL;
  if (C < 1 || C > FI_MAX_CLASSES || B < 1) return FI_ERR_SHAPE;
  hipLaunchKernelGGL(pdice_bwd_kernel, dim3(grid_for(HW, 256)), dim3(256), 0, (hipStream_t)stream, probs, labels, B,
                     HW, C, ignore_index, acc, weight, gscale, dprobs);
  FI_CHECK_LAUNCH();
  return 0;
}
