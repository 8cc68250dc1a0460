// The LAST convolution of the 3D U-Net: Conv3d(16 -> n_classes, 1x1x1) to fp32 logits (/root/reference/code/networks/unet_3D.py:57
// `self.final = nn.Conv3d(filters[0], n_classes, 1)`; n_classes <= 4) on full-resolution volumes.  The general path ran it as a
// zero fill of the fp32 output plus one accumulating 2D implicit-GEMM launch per sample, a dtype cast of the loss gradient in front
// of the backward and two more padded GEMMs: 77 + 88 + 126 us per 2 x 128^3 batch at 0.15-0.27 of an HBM roofline that is all there
// is to this layer (16 multiply-adds per voxel and class against 32 bytes read).  Three streaming kernels instead, one pass each:
//   forward : a thread per voxel -- its 16 channels as two 16-byte loads, n_classes dot products, bias, fp32 stores;
//   dgrad   : a thread per voxel -- the fp32 logit gradient straight from the loss (no cast pass), 16 channels out as two stores;
//   wgrad   : grid-stride over voxels, 16 x n_classes + n_classes accumulators per thread, wave / workgroup fold, one partial row
//             per workgroup and a fixed-order sum by a second launch (deterministic).
// The filter (<= 4 x 16 floats) and bias are read from device memory into LDS by every workgroup.
#include "common.h"

namespace {

constexpr int PC = 16;            // input channels
constexpr int PMAX = 4;           // output channels at most
constexpr int POINT_WG = 1024;    // workgroups (= partial rows) of the filter-gradient launch
constexpr int PROW = PMAX * PC + PMAX;

template <typename T> __device__ __forceinline__ void load16(const T* x, long i, float (&f)[PC]) {
  const typename DT<T>::vec_t v0 = reinterpret_cast<const typename DT<T>::vec_t*>(x + i * PC)[0];
  const typename DT<T>::vec_t v1 = reinterpret_cast<const typename DT<T>::vec_t*>(x + i * PC)[1];
  float lo[8], hi[8];
  VecWords<T>::unpack(v0, lo);
  VecWords<T>::unpack(v1, hi);
#pragma unroll
  for (int c = 0; c < 8; ++c) f[c] = lo[c], f[8 + c] = hi[c];
}

template <typename T>
__global__ __launch_bounds__(256) void conv3d_point_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, int co, float* __restrict__ y, long vox) {
  __shared__ float sw[PMAX][PC], sb[PMAX];
  if (threadIdx.x < PMAX * PC) (&sw[0][0])[threadIdx.x] = (int)(threadIdx.x / PC) < co ? w[threadIdx.x] : 0.f;
  if (threadIdx.x < PMAX) sb[threadIdx.x] = (bias && (int)threadIdx.x < co) ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= vox) return;
  float f[PC];
  load16<T>(x, i, f);
  float* dst = y + i * co;
#pragma unroll
  for (int o = 0; o < PMAX; ++o) {
    float s = sb[o];
#pragma unroll
    for (int c = 0; c < PC; ++c) s = __builtin_fmaf(f[c], sw[o][c], s);
    if (o < co) dst[o] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void conv3d_point_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, int co,
                                                                 T* __restrict__ dx, long vox) {
  __shared__ float sw[PMAX][PC];
  if (threadIdx.x < PMAX * PC) (&sw[0][0])[threadIdx.x] = (int)(threadIdx.x / PC) < co ? w[threadIdx.x] : 0.f;
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= vox) return;
  float g[PMAX];
#pragma unroll
  for (int o = 0; o < PMAX; ++o) g[o] = o < co ? dy[i * co + o] : 0.f;
  float lo[8], hi[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int o = 0; o < PMAX; ++o) {
      a = __builtin_fmaf(g[o], sw[o][c], a);
      b = __builtin_fmaf(g[o], sw[o][8 + c], b);
    }
    lo[c] = a, hi[c] = b;
  }
  reinterpret_cast<typename DT<T>::vec_t*>(dx + i * PC)[0] = VecWords<T>::pack(lo);
  reinterpret_cast<typename DT<T>::vec_t*>(dx + i * PC)[1] = VecWords<T>::pack(hi);
}

template <typename T>
__global__ __launch_bounds__(256) void conv3d_point_wgrad_kernel(const T* __restrict__ x, const float* __restrict__ dy, int co,
                                                                 float* __restrict__ part, long vox) {
  float acc[PMAX][PC], accb[PMAX];
#pragma unroll
  for (int o = 0; o < PMAX; ++o) {
    accb[o] = 0.f;
#pragma unroll
    for (int c = 0; c < PC; ++c) acc[o][c] = 0.f;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < vox; i += (long)gridDim.x * blockDim.x) {
    float f[PC], g[PMAX];
    load16<T>(x, i, f);
#pragma unroll
    for (int o = 0; o < PMAX; ++o) g[o] = o < co ? dy[i * co + o] : 0.f;
#pragma unroll
    for (int o = 0; o < PMAX; ++o) {
      accb[o] += g[o];
#pragma unroll
      for (int c = 0; c < PC; ++c) acc[o][c] = __builtin_fmaf(g[o], f[c], acc[o][c]);
    }
  }
  __shared__ float red[4][PROW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 0; o < PMAX; ++o) {
#pragma unroll
    for (int c = 0; c < PC; ++c) {
      const float s = wave_sum(acc[o][c]);
      if (lane == 0) red[wave][o * PC + c] = s;
    }
    const float s = wave_sum(accb[o]);
    if (lane == 0) red[wave][PMAX * PC + o] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < PROW; e += 256) part[(size_t)blockIdx.x * PROW + e] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
}

// dw[o][c] += / dbias[o] += the partial rows, in row order (fp64 sum): deterministic
__global__ __launch_bounds__(64) void conv3d_point_reduce_kernel(const float* part, int rows, int co, float* dw, float* dbias) {
  const int e = blockIdx.x;
  double tot = 0.0;
  for (int s = threadIdx.x; s < rows; s += 64) tot += (double)part[(size_t)s * PROW + e];
  tot = wave_sum(tot);
  if (threadIdx.x != 0) return;
  if (e < PMAX * PC) {
    if (dw && e / PC < co) dw[e] += (float)tot;                // dw is [co][16]: row o, column c = e
  } else if (dbias && e - PMAX * PC < co) {
    dbias[e - PMAX * PC] += (float)tot;
  }
}

inline bool point_ok(int dtype, long vox, int co) { return (dtype == FI_BF16 || dtype == FI_F16) && vox >= 1 && co >= 1 && co <= PMAX; }

}  // namespace

extern "C" long fi_conv3d_point_wgrad_workspace(void) { return (long)POINT_WG * PROW * (long)sizeof(float); }

extern "C" int fi_conv3d_point_fwd(int dtype, long voxels, int cout, const void* x, const float* w, const float* bias, float* y,
                                   void* stream) {
  if (!x || !w || !y) return FI_ERR_NULL;
  if (!point_ok(dtype, voxels, cout)) return dtype == FI_F32 ? FI_ERR_UNSUPPORTED : FI_ERR_SHAPE;
  const dim3 g((unsigned)((voxels + 255) / 256)), b(256);
  if (dtype == FI_BF16)
    hipLaunchKernelGGL(conv3d_point_fwd_kernel<bf16_t>, g, b, 0, (hipStream_t)stream, (const bf16_t*)x, w, bias, cout, y, voxels);
  else
    hipLaunchKernelGGL(conv3d_point_fwd_kernel<f16_t>, g, b, 0, (hipStream_t)stream, (const f16_t*)x, w, bias, cout, y, voxels);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_conv3d_point_dgrad(int dtype, long voxels, int cout, const float* dy, const float* w, void* dx, void* stream) {
  if (!dy || !w || !dx) return FI_ERR_NULL;
  if (!point_ok(dtype, voxels, cout)) return dtype == FI_F32 ? FI_ERR_UNSUPPORTED : FI_ERR_SHAPE;
  const dim3 g((unsigned)((voxels + 255) / 256)), b(256);
  if (dtype == FI_BF16)
    hipLaunchKernelGGL(conv3d_point_dgrad_kernel<bf16_t>, g, b, 0, (hipStream_t)stream, dy, w, cout, (bf16_t*)dx, voxels);
  else
    hipLaunchKernelGGL(conv3d_point_dgrad_kernel<f16_t>, g, b, 0, (hipStream_t)stream, dy, w, cout, (f16_t*)dx, voxels);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_conv3d_point_wgrad(int dtype, long voxels, int cout, const void* x, const float* dy, float* dw, float* dbias,
                                     void* workspace, long workspace_bytes, void* stream) {
  if (!x || !dy || !workspace || (!dw && !dbias)) return FI_ERR_NULL;
  if (!point_ok(dtype, voxels, cout)) return dtype == FI_F32 ? FI_ERR_UNSUPPORTED : FI_ERR_SHAPE;
  if (workspace_bytes < fi_conv3d_point_wgrad_workspace()) return FI_ERR_SHAPE;
  const dim3 g(POINT_WG), b(256);
  if (dtype == FI_BF16)
    hipLaunchKernelGGL(conv3d_point_wgrad_kernel<bf16_t>, g, b, 0, (hipStream_t)stream, (const bf16_t*)x, dy, cout, (float*)workspace,
                       voxels);
  else
    hipLaunchKernelGGL(conv3d_point_wgrad_kernel<f16_t>, g, b, 0, (hipStream_t)stream, (const f16_t*)x, dy, cout, (float*)workspace,
                       voxels);
  hipLaunchKernelGGL(conv3d_point_reduce_kernel, dim3(PROW), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, POINT_WG, cout,
                     dw, dbias);
  FI_CHECK_LAUNCH();
  return 0;
}
