// 3D companions of the pooling / resampling kernels for the unet_3D surface:
// MaxPool3d(2) and trilinear x2 up-sampling (align_corners = False, nn.Upsample's default) over dense NDHWC
// (a volume is D consecutive NHWC slices, so the 2D convolution / normalisation kernels work on it slice-wise).
// HBM-bound gather kernels: one 16-byte channel vector per lane.
#include "common.h"

template <typename T>
__device__ __forceinline__ void ld(const T* p, float (&f)[DT<T>::VG]) {
  union {
    typename DT<T>::vec_t v;
    T e[DT<T>::VG];
  } u;
  u.v = *reinterpret_cast<const typename DT<T>::vec_t*>(p);
#pragma unroll
  for (int j = 0; j < DT<T>::VG; ++j) f[j] = to_f32(u.e[j]);
}
template <typename T>
__device__ __forceinline__ void st(T* p, const float (&f)[DT<T>::VG]) {
  union {
    typename DT<T>::vec_t v;
    T e[DT<T>::VG];
  } u;
#pragma unroll
  for (int j = 0; j < DT<T>::VG; ++j) u.e[j] = from_f32<T>(f[j]);
  *reinterpret_cast<typename DT<T>::vec_t*>(p) = u.v;
}

static inline int grid3(long work, int per_block) {
  long b = (work + per_block - 1) / per_block;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- MaxPool3d(kernel 2, stride 2).  x [N,D,H,W,C] -> y [N,D/2,H/2,W/2,C].  Backward routes dy to the FIRST maximum of
//      the window in (d, h, w) scan order (strict >), as ATen does.
//      ADD: dx = add + the routed gradient -- an encoder feature that also feeds a decoder skip connection receives two gradients,
//      and the sum is made here instead of by an elementwise launch of autograd's (one fp32 add per element, then the storage type).
template <typename T, bool BWD, bool ADD = false>
__global__ __launch_bounds__(256) void maxpool3d_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out,
                                                        int N, int D, int H, int W, int C, const T* __restrict__ add = nullptr) {
  constexpr int VG = DT<T>::VG;
  const int CV = C / VG, Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long nvec = (long)N * Do * Ho * Wo * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    p /= Ho;
    const int od = (int)(p % Do);
    const int n = (int)(p / Do);
    size_t offs[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int dz = q >> 2, dyy = (q >> 1) & 1, dx = q & 1;
      offs[q] = ((((size_t)n * D + 2 * od + dz) * H + 2 * oy + dyy) * W + 2 * ox + dx) * C + (size_t)cv * VG;
    }
    float v[8][VG];
#pragma unroll
    for (int q = 0; q < 8; ++q) ld<T>(x + offs[q], v[q]);
    if (!BWD) {
      float m[VG];
#pragma unroll
      for (int j = 0; j < VG; ++j) {
        float mm = v[0][j];
#pragma unroll
        for (int q = 1; q < 8; ++q) mm = v[q][j] > mm ? v[q][j] : mm;
        m[j] = mm;
      }
      st<T>(out + i * VG, m);
    } else {
      float g[VG], o[8][VG];
      ld<T>(dy + i * VG, g);
#pragma unroll
      for (int j = 0; j < VG; ++j) {
        int best = 0;
        float mm = v[0][j];
#pragma unroll
        for (int q = 1; q < 8; ++q)
          if (v[q][j] > mm) {
            mm = v[q][j];
            best = q;
          }
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q][j] = q == best ? g[j] : 0.f;
      }
      if (ADD) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float s[VG];
          ld<T>(add + offs[q], s);
#pragma unroll
          for (int j = 0; j < VG; ++j) o[q][j] += s[j];
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) st<T>(out + offs[q], o[q]);
    }
  }
}

// ---- trilinear x2, align_corners = False:  src = (o + 0.5) / 2 - 0.5 clamped at 0;  i0 = floor(src), i1 = min(i0 + 1,
//      in - 1), l1 = src - i0  (ATen's area_pixel_compute_source_index).
__device__ __forceinline__ void half_coord(int o, int in, int& i0, int& i1, float& l0, float& l1) {
  float src = ((float)o + 0.5f) * 0.5f - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + 1 < in ? i0 + 1 : in - 1;
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample3d_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int d, int h,
                                                             int w, int C) {
  constexpr int VG = DT<T>::VG;
  const int CV = C / VG, Do = 2 * d, Ho = 2 * h, Wo = 2 * w;
  const long nvec = (long)N * Do * Ho * Wo * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    p /= Ho;
    const int od = (int)(p % Do);
    const int n = (int)(p / Do);
    int z0, z1, y0, y1, x0, x1;
    float lz[2], ly[2], lx[2];
    half_coord(od, d, z0, z1, lz[0], lz[1]);
    half_coord(oy, h, y0, y1, ly[0], ly[1]);
    half_coord(ox, w, x0, x1, lx[0], lx[1]);
    const int zi[2] = {z0, z1}, yi[2] = {y0, y1}, xi[2] = {x0, x1};
    const T* b = x + (size_t)n * d * h * w * C + (size_t)cv * VG;
    float acc[VG];
#pragma unroll
    for (int j = 0; j < VG; ++j) acc[j] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float v[VG];
          ld<T>(b + (((size_t)zi[a] * h + yi[bb]) * w + xi[c]) * C, v);
          const float wgt = lz[a] * ly[bb] * lx[c];
#pragma unroll
          for (int j = 0; j < VG; ++j) acc[j] += wgt * v[j];
        }
    st<T>(y + i * VG, acc);
  }
}

// taps of input coordinate i along one axis (align_corners = False): outputs 2i-1 .. 2i+2 are the only candidates;
// wt[k] = weight output (2i - 1 + k) gives to i  (0 outside the tensor)
__device__ __forceinline__ void half_taps(int i, int in, float (&wt)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int o = 2 * i - 1 + k;
    float wv = 0.f;
    if (o >= 0 && o < 2 * in) {
      int i0, i1;
      float l0, l1;
      half_coord(o, in, i0, i1, l0, l1);
      if (i0 == i) wv += l0;
      if (i1 == i) wv += l1;
    }
    wt[k] = wv;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample3d_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int d, int h,
                                                             int w, int C) {
  constexpr int VG = DT<T>::VG;
  const int CV = C / VG, Do = 2 * d, Ho = 2 * h, Wo = 2 * w;
  const long nvec = (long)N * d * h * w * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int ix = (int)(p % w);
    p /= w;
    const int iy = (int)(p % h);
    p /= h;
    const int iz = (int)(p % d);
    const int n = (int)(p / d);
    float wz[4], wy[4], wx[4];
    half_taps(iz, d, wz);
    half_taps(iy, h, wy);
    half_taps(ix, w, wx);
    const T* b = dy + (size_t)n * Do * Ho * Wo * C + (size_t)cv * VG;
    float acc[VG];
#pragma unroll
    for (int j = 0; j < VG; ++j) acc[j] = 0.f;
    for (int a = 0; a < 4; ++a) {
      if (wz[a] == 0.f) continue;
      const int oz = min(max(2 * iz - 1 + a, 0), Do - 1);
      for (int bb = 0; bb < 4; ++bb) {
        if (wy[bb] == 0.f) continue;
        const int oy = min(max(2 * iy - 1 + bb, 0), Ho - 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int ox = min(max(2 * ix - 1 + c, 0), Wo - 1);
          float g[VG];
          ld<T>(b + (((size_t)oz * Ho + oy) * Wo + ox) * C, g);
          const float wgt = wz[a] * wy[bb] * wx[c];
#pragma unroll
          for (int j = 0; j < VG; ++j) acc[j] += wgt * g[j];
        }
      }
    }
    st<T>(dx + i * VG, acc);
  }
}


// ---- row forms of the two kernels above for the launches that matter (unet_3D's four UpBlocks at 2 x 128^3: 228 + 209 us of a
//      6.5 ms iteration in the flat forms -- three 64-bit divisions and 8 scattered loads per output vector forward, 64 candidate
//      loads under branches per input vector backward -- against 44 us of traffic each).  The interpolation is a tensor product:
//      forward: the 2 x 2 output rows (od in {2a-1, 2a}, oy in {2b-1, 2b}) of a workgroup blend the SAME four input rows
//               (z in {a-1, a}, y in {b-1, b}) -- loaded once, blended per output row into an fp32 row in LDS, and the x axis is two
//               LDS reads per output vector: 0.5 global loads per output vector instead of 8;
//      backward: a workgroup owns an input row; its 4 x 4 gradient rows fold into one fp32 row of 2w x C in LDS (16 coalesced
//               loads per vector, no branch: out-of-range rows are clamped addresses with zero weight), the x axis is four LDS reads.
//      Workgroup ids go round-robin to the 8 XCDs (an L2 each): XCD g takes a contiguous range of the (n, z, y) order, so rows that
//      neighbours share are fetched once per XCD.  Weights are half_coord's / half_taps', as in the flat forms.
__device__ __forceinline__ int xcd_contiguous(int nblocks_padded) {      // nblocks_padded = gridDim.x, a multiple of 8
  return (int)(blockIdx.x & 7u) * (nblocks_padded >> 3) + (int)(blockIdx.x >> 3);
}

template <typename T>
__global__ __launch_bounds__(256) void upsample3d_fwd_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int d, int h, int w,
                                                                  int C, int cv_shift) {
  constexpr int VG = DT<T>::VG, Q = VG / 4;
  extern __shared__ __attribute__((aligned(16))) float4 s_t[];     // [4 output rows][w][C] fp32
  const int Do = 2 * d, Ho = 2 * h, Wo = 2 * w, CV = 1 << cv_shift;
  const int logical = xcd_contiguous((int)gridDim.x);
  if (logical >= N * (d + 1) * (h + 1)) return;
  const int b = logical % (h + 1), a = (logical / (h + 1)) % (d + 1), n = logical / ((h + 1) * (d + 1));
  const int zA = max(a - 1, 0), zB = min(a, d - 1), yA = max(b - 1, 0), yB = min(b, h - 1);
  // weights of the (up to) two output coordinates of this block on the two input coordinates, per axis
  float wz[2][2], wy[2][2];
  bool vz[2], vy[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int od = 2 * a - 1 + k, oy = 2 * b - 1 + k;
    vz[k] = od >= 0 && od < Do;
    vy[k] = oy >= 0 && oy < Ho;
    int i0, i1;
    float l0, l1;
    half_coord(vz[k] ? od : 0, d, i0, i1, l0, l1);
    wz[k][0] = (i0 == zA ? l0 : 0.f) + (i1 == zA ? l1 : 0.f);
    wz[k][1] = zB == zA ? 0.f : (i0 == zB ? l0 : 0.f) + (i1 == zB ? l1 : 0.f);
    half_coord(vy[k] ? oy : 0, h, i0, i1, l0, l1);
    wy[k][0] = (i0 == yA ? l0 : 0.f) + (i1 == yA ? l1 : 0.f);
    wy[k][1] = yB == yA ? 0.f : (i0 == yB ? l0 : 0.f) + (i1 == yB ? l1 : 0.f);
  }
  const int inv = w << cv_shift, outv = Wo << cv_shift;             // vectors of an input row / an output row
  const size_t rowe = (size_t)w * C;
  const T* const rAA = x + (((size_t)n * d + zA) * h + yA) * rowe;
  const T* const rAB = x + (((size_t)n * d + zA) * h + yB) * rowe;
  const T* const rBA = x + (((size_t)n * d + zB) * h + yA) * rowe;
  const T* const rBB = x + (((size_t)n * d + zB) * h + yB) * rowe;
#pragma clang loop vectorize(disable) interleave(disable)
  for (int e = threadIdx.x; e < inv; e += 256) {
    float vAA[VG], vAB[VG], vBA[VG], vBB[VG];
    ld<T>(rAA + (size_t)e * VG, vAA);
    ld<T>(rAB + (size_t)e * VG, vAB);
    ld<T>(rBA + (size_t)e * VG, vBA);
    ld<T>(rBB + (size_t)e * VG, vBB);
#pragma unroll
    for (int kz = 0; kz < 2; ++kz)
#pragma unroll
      for (int ky = 0; ky < 2; ++ky) {
        float t[VG];
#pragma unroll
        for (int j = 0; j < VG; ++j)
          t[j] = wz[kz][0] * (wy[ky][0] * vAA[j] + wy[ky][1] * vAB[j]) + wz[kz][1] * (wy[ky][0] * vBA[j] + wy[ky][1] * vBB[j]);
#pragma unroll
        for (int q = 0; q < Q; ++q)
          s_t[((kz * 2 + ky) * inv + e) * Q + q] = make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
      }
  }
  __syncthreads();
#pragma clang loop vectorize(disable) interleave(disable)
  for (int o = threadIdx.x; o < outv; o += 256) {
    const int ox = o >> cv_shift, cv = o - (ox << cv_shift);
    int x0, x1;
    float l0, l1;
    half_coord(ox, w, x0, x1, l0, l1);
#pragma unroll
    for (int kz = 0; kz < 2; ++kz)
#pragma unroll
      for (int ky = 0; ky < 2; ++ky) {
        if (!(vz[kz] && vy[ky])) continue;                             // (workgroup-uniform: the tensor's faces)
        const float4* const t0 = s_t + ((kz * 2 + ky) * inv + (x0 << cv_shift) + cv) * Q;
        const float4* const t1 = s_t + ((kz * 2 + ky) * inv + (x1 << cv_shift) + cv) * Q;
        float r[VG];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const float4 u0 = t0[q], u1 = t1[q];
          r[4 * q] = l0 * u0.x + l1 * u1.x, r[4 * q + 1] = l0 * u0.y + l1 * u1.y;
          r[4 * q + 2] = l0 * u0.z + l1 * u1.z, r[4 * q + 3] = l0 * u0.w + l1 * u1.w;
        }
        const int od = 2 * a - 1 + kz, oy = 2 * b - 1 + ky;
        st<T>(y + (((size_t)n * Do + od) * Ho + oy) * Wo * C + (size_t)o * VG, r);
      }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample3d_bwd_rows_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int d, int h, int w,
                                                                  int C, int cv_shift) {
  constexpr int VG = DT<T>::VG, Q = VG / 4;
  extern __shared__ __attribute__((aligned(16))) float4 s_t[];     // [2w][C] fp32
  const int Do = 2 * d, Ho = 2 * h, Wo = 2 * w, CV = 1 << cv_shift;
  const int logical = xcd_contiguous((int)gridDim.x);
  if (logical >= N * d * h) return;
  const int iy = logical % h, iz = (logical / h) % d, n = logical / (h * d);
  float wz[4], wy[4];
  half_taps(iz, d, wz);
  half_taps(iy, h, wy);
  const size_t rowe = (size_t)Wo * C;
  const T* rows[4][4];
  float wgt[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int oz = min(max(2 * iz - 1 + a, 0), Do - 1), oy = min(max(2 * iy - 1 + b, 0), Ho - 1);
      rows[a][b] = dy + (((size_t)n * Do + oz) * Ho + oy) * rowe;
      wgt[a][b] = wz[a] * wy[b];                                       // 0 for a row outside the tensor (half_taps)
    }
  const int outv = Wo << cv_shift, inv = w << cv_shift;
#pragma clang loop vectorize(disable) interleave(disable)
  for (int i = threadIdx.x; i < outv; i += 256) {
    typedef unsigned raw16 __attribute__((ext_vector_type(4)));          // 16 bytes = VG elements of any T
    raw16 v[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) v[a][b] = *reinterpret_cast<const raw16*>(rows[a][b] + (size_t)i * VG);
    // all sixteen requests before the first use (hipcc otherwise issues five, waits, and trickles the rest in one at a time)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(v[a][b]));
    float acc[VG];
#pragma unroll
    for (int j = 0; j < VG; ++j) acc[j] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        union {
          raw16 v;
          T e[VG];
        } u;
        u.v = v[a][b];
#pragma unroll
        for (int j = 0; j < VG; ++j) acc[j] += wgt[a][b] * to_f32(u.e[j]);
      }
#pragma unroll
    for (int q = 0; q < Q; ++q) s_t[i * Q + q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
  }
  __syncthreads();
  T* const out = dx + (((size_t)n * d + iz) * h + iy) * (size_t)w * C;
#pragma clang loop vectorize(disable) interleave(disable)
  for (int e = threadIdx.x; e < inv; e += 256) {
    const int ix = e >> cv_shift, cv = e - (ix << cv_shift);
    float wx[4];
    half_taps(ix, w, wx);
    float acc[VG];
#pragma unroll
    for (int j = 0; j < VG; ++j) acc[j] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ox = min(max(2 * ix - 1 + c, 0), Wo - 1);
      const float4* const t = s_t + ((ox << cv_shift) + cv) * Q;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float4 u = t[q];
        acc[4 * q] += wx[c] * u.x, acc[4 * q + 1] += wx[c] * u.y, acc[4 * q + 2] += wx[c] * u.z, acc[4 * q + 3] += wx[c] * u.w;
      }
    }
    st<T>(out + (size_t)e * VG, acc);
  }
}

// the row forms apply to whole-vector channel counts with C / VG a power of two and rows that fit LDS (FI_UP3D_ROWS=0: flat forms, A/B)
static inline int up3d_rows_shift(int w, int C, int vg, int rows_in_lds) {
  static const long on = [] {
    const char* v = getenv("FI_UP3D_ROWS");
    return v ? atol(v) : 1L;
  }();
  if (!on) return -1;
  const int CV = C / vg;
  int shift = 0;
  while ((1 << shift) < CV) ++shift;
  if ((1 << shift) != CV) return -1;
  if ((long)rows_in_lds * w * C * 4 > 48 * 1024) return -1;
  if ((long)w * CV < 32) return -1;
  return shift;
}

template <typename T>
static int launch_up3d_fwd(const void* x, void* y, int N, int d, int h, int w, int C, hipStream_t st_) {
  constexpr int VG = DT<T>::VG;
  const long nvec = (long)N * 8 * d * h * w * (C / VG), blocks = (long)N * (d + 1) * (h + 1);
  const int shift = up3d_rows_shift(w, C, VG, 4);
  if (shift >= 0 && blocks >= 64 && blocks < (1L << 30)) {
    hipLaunchKernelGGL(upsample3d_fwd_rows_kernel<T>, dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(256), (size_t)4 * w * C * sizeof(float), st_,
                       (const T*)x, (T*)y, N, d, h, w, C, shift);
  } else {
    hipLaunchKernelGGL(upsample3d_fwd_kernel<T>, dim3(grid3(nvec, 256)), dim3(256), 0, st_, (const T*)x, (T*)y, N, d, h, w, C);
  }
  FI_CHECK_LAUNCH();
  return 0;
}

template <typename T>
static int launch_up3d_bwd(const void* dy, void* dx, int N, int d, int h, int w, int C, hipStream_t st_) {
  constexpr int VG = DT<T>::VG;
  const long nvec = (long)N * d * h * w * (C / VG), blocks = (long)N * d * h;
  const int shift = up3d_rows_shift(2 * w, C, VG, 1);
  if (shift >= 0 && blocks >= 64 && blocks < (1L << 30)) {
    hipLaunchKernelGGL(upsample3d_bwd_rows_kernel<T>, dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(256), (size_t)2 * w * C * sizeof(float), st_,
                       (const T*)dy, (T*)dx, N, d, h, w, C, shift);
  } else {
    hipLaunchKernelGGL(upsample3d_bwd_kernel<T>, dim3(grid3(nvec, 256)), dim3(256), 0, st_, (const T*)dy, (T*)dx, N, d, h, w, C);
  }
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_maxpool3d_fwd(int dtype, const void* x, void* y, int N, int D, int H, int W, int C, void* stream) {
  if (!x || !y) return FI_ERR_NULL;
  if ((D & 1) || (H & 1) || (W & 1)) return FI_ERR_SHAPE;
  hipStream_t st_ = (hipStream_t)stream;
  const int vg = dtype == FI_F32 ? 4 : 8;
  if (dtype != FI_F32 && dtype != FI_BF16 && dtype != FI_F16) return FI_ERR_DTYPE;
  if (C % vg) return FI_ERR_SHAPE;
  const long nvec = (long)N * (D / 2) * (H / 2) * (W / 2) * (C / vg);
  if (dtype == FI_F32)
    hipLaunchKernelGGL((maxpool3d_kernel<float, false>), dim3(grid3(nvec, 256)), dim3(256), 0, st_, (const float*)x,
                       (const float*)nullptr, (float*)y, N, D, H, W, C);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL((maxpool3d_kernel<f16_t, false>), dim3(grid3(nvec, 256)), dim3(256), 0, st_, (const f16_t*)x,
                       (const f16_t*)nullptr, (f16_t*)y, N, D, H, W, C);
  else
    hipLaunchKernelGGL((maxpool3d_kernel<bf16_t, false>), dim3(grid3(nvec, 256)), dim3(256), 0, st_, (const bf16_t*)x,
                       (const bf16_t*)nullptr, (bf16_t*)y, N, D, H, W, C);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_maxpool3d_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int D, int H, int W, int C,
                                void* stream) {
  if (!x || !dy || !dx) return FI_ERR_NULL;
  if ((D & 1) || (H & 1) || (W & 1)) return FI_ERR_SHAPE;
  hipStream_t st_ = (hipStream_t)stream;
  const int vg = dtype == FI_F32 ? 4 : 8;
  if (dtype != FI_F32 && dtype != FI_BF16 && dtype != FI_F16) return FI_ERR_DTYPE;
  if (C % vg) return FI_ERR_SHAPE;
  const long nvec = (long)N * (D / 2) * (H / 2) * (W / 2) * (C / vg);
  if (dtype == FI_F32)
    hipLaunchKernelGGL((maxpool3d_kernel<float, true>), dim3(grid3(nvec, 256)), dim3(256), 0, st_, (const float*)x,
                       (const float*)dy, (float*)dx, N, D, H, W, C);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL((maxpool3d_kernel<f16_t, true>), dim3(grid3(nvec, 256)), dim3(256), 0, st_, (const f16_t*)x,
                       (const f16_t*)dy, (f16_t*)dx, N, D, H, W, C);
  else
    hipLaunchKernelGGL((maxpool3d_kernel<bf16_t, true>), dim3(grid3(nvec, 256)), dim3(256), 0, st_, (const bf16_t*)x,
                       (const bf16_t*)dy, (bf16_t*)dx, N, D, H, W, C);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_maxpool3d_bwd_add(int dtype, const void* x, const void* dy, const void* add, void* dx, int N, int D, int H, int W,
                                    int C, void* stream) {
  if (!x || !dy || !add || !dx) return FI_ERR_NULL;
  if ((D & 1) || (H & 1) || (W & 1)) return FI_ERR_SHAPE;
  hipStream_t st_ = (hipStream_t)stream;
  const int vg = dtype == FI_F32 ? 4 : 8;
  if (dtype != FI_F32 && dtype != FI_BF16 && dtype != FI_F16) return FI_ERR_DTYPE;
  if (C % vg) return FI_ERR_SHAPE;
  const long nvec = (long)N * (D / 2) * (H / 2) * (W / 2) * (C / vg);
  if (dtype == FI_F32)
    hipLaunchKernelGGL((maxpool3d_kernel<float, true, true>), dim3(grid3(nvec, 256)), dim3(256), 0, st_, (const float*)x,
                       (const float*)dy, (float*)dx, N, D, H, W, C, (const float*)add);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL((maxpool3d_kernel<f16_t, true, true>), dim3(grid3(nvec, 256)), dim3(256), 0, st_, (const f16_t*)x,
                       (const f16_t*)dy, (f16_t*)dx, N, D, H, W, C, (const f16_t*)add);
  else
    hipLaunchKernelGGL((maxpool3d_kernel<bf16_t, true, true>), dim3(grid3(nvec, 256)), dim3(256), 0, st_, (const bf16_t*)x,
                       (const bf16_t*)dy, (bf16_t*)dx, N, D, H, W, C, (const bf16_t*)add);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_upsample3d2x_fwd(int dtype, const void* x, void* y, int N, int d, int h, int w, int C, void* stream) {
  if (!x || !y) return FI_ERR_NULL;
  hipStream_t st_ = (hipStream_t)stream;
  const int vg = dtype == FI_F32 ? 4 : 8;
  if (dtype != FI_F32 && dtype != FI_BF16 && dtype != FI_F16) return FI_ERR_DTYPE;
  if (C % vg) return FI_ERR_SHAPE;
  if (dtype == FI_F32) return launch_up3d_fwd<float>(x, y, N, d, h, w, C, st_);
  if (dtype == FI_F16) return launch_up3d_fwd<f16_t>(x, y, N, d, h, w, C, st_);
  return launch_up3d_fwd<bf16_t>(x, y, N, d, h, w, C, st_);
}

extern "C" int fi_upsample3d2x_bwd(int dtype, const void* dy, void* dx, int N, int d, int h, int w, int C, void* stream) {
  if (!dy || !dx) return FI_ERR_NULL;
  hipStream_t st_ = (hipStream_t)stream;
  const int vg = dtype == FI_F32 ? 4 : 8;
  if (dtype != FI_F32 && dtype != FI_BF16 && dtype != FI_F16) return FI_ERR_DTYPE;
  if (C % vg) return FI_ERR_SHAPE;
  if (dtype == FI_F32) return launch_up3d_bwd<float>(dy, dx, N, d, h, w, C, st_);
  if (dtype == FI_F16) return launch_up3d_bwd<f16_t>(dy, dx, N, d, h, w, C, st_);
  return launch_up3d_bwd<bf16_t>(dy, dx, N, d, h, w, C, st_);
}

// ------------------------------------------------------------------------------------------------------------------------
// Both one-launch operands of EVERY 3x3x3 convolution of a model in one launch (the 3D path's fi_pack_weights_multi): from the
// fp32 parameter [Cout][Cin][3][3][3] the forward operand [Cout][9][3][Cin] (fi_conv3d_fwd_fused: filter position major, the
// depth taps as channel groups) and the dgrad operand [Cin][9][3][Cout] with all three filter axes reversed (fi_conv3d_dgrad_fused),
// rounded to the storage type as fi_pack_weights rounds.  A unet_3D iteration rebuilt them with 51 cast / flip / permute launches
// of 4-6 us (0.27 ms of a 7.6 ms iteration).  table: rows of 6 int64 {src, dst forward, dst dgrad, Cout, Cin, first block}; a block
// takes 256 (co, ci) pairs of its tensor -- ci fastest for the forward operand's stores, co fastest for the dgrad operand's.
template <typename T>
__global__ __launch_bounds__(256) void pack_weights3d_multi_kernel(const long long* __restrict__ table, int ntensors) {
  int row = 0;
  for (int t = 1; t < ntensors; ++t)
    if ((long long)blockIdx.x >= table[(size_t)t * 6 + 5]) row = t;
  const long long* r = table + (size_t)row * 6;
  const float* __restrict__ src = reinterpret_cast<const float*>(r[0]);
  T* __restrict__ d0 = reinterpret_cast<T*>(r[1]);
  T* __restrict__ d1 = reinterpret_cast<T*>(r[2]);
  const int cout = (int)r[3], cin = (int)r[4];
  const long npair = (long)cout * cin;
  const long p = ((long)blockIdx.x - r[5]) * 256 + threadIdx.x;
  if (p >= npair) return;
  {                                                        // forward operand: pair = co * Cin + ci
    const int co = (int)(p / cin), ci = (int)(p % cin);
    const float* w = src + p * 27;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {                   // tap = t * 9 + rs
      const int t = tap / 9, rs = tap % 9;
      d0[((size_t)(co * 9 + rs) * 3 + t) * cin + ci] = from_f32<T>(w[tap]);
    }
  }
  {                                                        // dgrad operand: pair = ci * Cout + co
    const int ci = (int)(p / cout), co = (int)(p % cout);
    const float* w = src + ((size_t)co * cin + ci) * 27;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int t = tap / 9, rs = tap % 9;
      d1[((size_t)(ci * 9 + (8 - rs)) * 3 + (2 - t)) * cout + co] = from_f32<T>(w[tap]);
    }
  }
}

extern "C" int fi_pack_weights3d_multi(const long long* table, int ntensors, int nblocks, int dtype, void* stream) {
  if (!table) return FI_ERR_NULL;
  if (ntensors <= 0 || nblocks <= 0) return 0;
  hipStream_t st_ = (hipStream_t)stream;
  if (dtype == FI_BF16)
    hipLaunchKernelGGL(pack_weights3d_multi_kernel<bf16_t>, dim3(nblocks), dim3(256), 0, st_, table, ntensors);
  else if (dtype == FI_F16)
    hipLaunchKernelGGL(pack_weights3d_multi_kernel<f16_t>, dim3(nblocks), dim3(256), 0, st_, table, ntensors);
  else
    return FI_ERR_DTYPE;
  FI_CHECK_LAUNCH();
  return 0;
}
