// The FIRST convolution of the 3D U-Net: Conv3d(1 -> 16, 3x3x3, pad 1) on full-resolution volumes
// (/root/reference/code/networks/unet_3D.py:38 `UnetConv3(in_channels, filters[0])`, networks/utils.py:99-123; BASELINE configs[3]/[4]:
// 2 x 1 x 128^3).  With ONE input channel the contraction is 27 long: the implicit-GEMM forms pad it to 32 per depth tap and run
// three read-modify-write passes over the 134 MB output (259 us forward, 229 us filter gradient per 2 x 128^3 batch).  Round 5's
// answer was a vector-ALU stencil that streams the output once (86 / 105 us: 54 packed FMAs per thread and row -- ALU-bound); round 6
// puts the same sums on the matrix pipe with the taps as the contraction (forward) / the voxels as the contraction (filter
// gradient), an input halo tile in LDS and fragments assembled from two-byte LDS reads: 42 / 42 us against the 27 us of the 134 MB.
// Both kernels keep round 5's decomposition (a workgroup = 64 rows x 64 columns of one slice), its statistics and its partial-slice
// layout, so the second launch of the filter gradient and the host side are unchanged.
#include "common.h"

namespace {

typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr int FIRST_RY = 64;          // rows of a slice one workgroup walks (a run); its four waves take 16 (forward) / 32 (gradient) columns each
constexpr unsigned OOB = 0xFFFFFFF0u;

struct First3dArgs {
  const void* x;        // [N][D][H][W] (one channel)
  const float* w;       // [16][27] fp32, taps in (kd, kh, kw) order
  const float* bias;    // [16] or NULL
  void* y;              // forward: [N][D][H][W][16]; gradient: dy, the same layout
  double* stats;        // forward: [N][SLOTS][16][2] (stats_stride doubles per sample) or NULL
  long stats_stride;
  float* part;          // gradient: [workgroups][16 * 27 + 16]
  int N, D, H, W;
};

// The halo tile of the MFMA forms: rows rr = kd * (R + 2) + r of 66 columns, thread -> (rr, c) advancing by 256 elements (3 rows + 58
// columns).  Loads in batches of 13 before their LDS stores: one load-then-store per iteration serialises 52 L2 round trips (the
// first build: 49 us of which ~ 25 were this loop).
template <typename T, int TR, int TP>
__device__ __forceinline__ void first3d_fill_tile(T* xt, const __amdgpu_buffer_rsrc_t& rx, int tid, int R, int n, int z, int y0, int x0, int D, int H,
                                                  int W) {
  constexpr int NIT = (3 * TR * 66 + 255) / 256, BATCH = 13;
  const int rows = 3 * (R + 2);
  int rr = tid / 66, c = tid - rr * 66;
#pragma unroll 1
  for (int it0 = 0; it0 < NIT; it0 += BATCH) {
    if (rr >= rows) break;
    unsigned short v[BATCH];
    int idx[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int kd = rr >= 2 * (R + 2) ? 2 : (rr >= R + 2 ? 1 : 0), r = rr - kd * (R + 2);
      const int gz = z + kd - 1, gy = y0 - 1 + r, gx = x0 - 1 + c;
      const bool in = rr < rows;
      const bool ok = in && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned off = ok ? (unsigned)((((long)n * D + gz) * H + gy) * W + gx) * 2u : OOB;
      v[u] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rx, off, 0, 0);
      idx[u] = in ? (kd * TR + r) * TP + c : -1;
      c += 58, rr += 3;
      if (c >= 66) c -= 66, ++rr;
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u)
      if (idx[u] >= 0) xt[idx[u]] = __builtin_bit_cast(T, v[u]);
  }
}

// ---- the forward on the matrix pipe (round 6).  The stencil above is vector-ALU-bound: 27 x 16 multiply-adds per voxel are 54 packed
// FMAs per thread and row, 86 us per 2 x 128^3 against the 27 us the 134 MB output needs.  The same sum as ONE 16 x 16 x 32 MFMA per 16
// voxels: M = the 16 output channels (A = the filter), N = 16 consecutive voxels of a row, K = the 27 taps (32 slots).  The input halo
// tile (3 slices x 66 rows x 66 columns, 27 KB) is staged once per workgroup; a lane's B fragment is 8 two-byte LDS reads (K slot
// (g, j): g = lane >> 4 < 3 is the depth tap and j the first eight of its nine (kh, kw); g = 3 holds the three (kh, kw) = (2, 2) taps).
// The filter stays fp32-EXACT: each weight is split into three 16-bit parts (hi + mid + lo = its 24 bits), three MFMAs against the same B
// fragment -- the result differs from the fp32 stencil by accumulation order only, so the tests' "one rounding of the storage type"
// bar holds (two parts do not: 2^-16 of a term is above that bar where the 27 terms cancel).
template <typename T> struct FragOf;
template <> struct FragOf<bf16_t> { typedef bf16x8 type; };
template <> struct FragOf<f16_t> { typedef f16x8 type; };

template <typename T>
__global__ __launch_bounds__(256) void conv3d_first_fwd_mfma_kernel(First3dArgs a) {
  typedef typename FragOf<T>::type frag_t;
  constexpr int TR = FIRST_RY + 2, TP = 68;                    // tile rows (with halo), row pitch in elements (66 used)
  __shared__ __attribute__((aligned(16))) T xt[3 * TR * TP];
  __shared__ float red[4][4][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int D = a.D, H = a.H, W = a.W;
  const int xblocks = (W + 63) / 64, yruns = (H + FIRST_RY - 1) / FIRST_RY;
  int b = blockIdx.x;
  const int xb = b % xblocks;
  b /= xblocks;
  const int yr = b % yruns;
  b /= yruns;
  const int z = b % D, n = b / D;
  const int y0 = yr * FIRST_RY, y1 = min(H, y0 + FIRST_RY), R = y1 - y0;
  const int x = xb * 64 + wave * 16 + li;
  const unsigned vol = (unsigned)a.N * (unsigned)D * (unsigned)H * (unsigned)W;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, vol * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, vol * 32u, 0x00020000);

  first3d_fill_tile<T, TR, TP>(xt, rx, tid, R, n, z, y0, xb * 64, D, H, W);

  // ---- the filter as two A fragments (hi / lo parts of the fp32 weights) and the lane's eight tap offsets into the tile
  frag_t whi, wmid, wlo;
  unsigned off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int kd, kh, kw, t;
    if (g < 3) {
      kd = g, kh = j / 3, kw = j % 3, t = g * 9 + j;
    } else {
      kd = j < 3 ? j : 0, kh = 2, kw = 2, t = j < 3 ? j * 9 + 8 : -1;
    }
    const float wv = t >= 0 ? a.w[li * 27 + t] : 0.f;
    const T hi = (T)wv, mid = (T)(wv - (float)hi);
    whi[j] = hi;
    wmid[j] = mid;
    wlo[j] = (T)(wv - (float)hi - (float)mid);
    off[j] = (unsigned)(((kd * TR + kh) * TP + wave * 16 + li + kw) * 2);
  }
  f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.bias) bias4 = f32x4{a.bias[g * 4], a.bias[g * 4 + 1], a.bias[g * 4 + 2], a.bias[g * 4 + 3]};
  const unsigned xpen = x < W ? 0u : 0x80000000u;
  const float m = x < W ? 1.f : 0.f;
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const char* const tile = reinterpret_cast<const char*>(xt);
  unsigned yoff = ((unsigned)((((long)n * D + z) * H + y0) * W + x) * 16u + (unsigned)g * 4u) * 2u;
  for (int r = 0; r < R; ++r) {
    frag_t bv;
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = *reinterpret_cast<const T*>(tile + off[j] + (unsigned)(r * TP * 2));
    f32x4 acc = mfma16(wlo, bv, bias4);
    acc = mfma16(wmid, bv, acc);
    acc = mfma16(whi, bv, acc);
    float v4[4] = {acc[0], acc[1], acc[2], acc[3]};
    const typename Quad<T>::q_t st = Quad<T>::pack(v4);         // v4 := the values as stored
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, st), ry, yoff | xpen, 0, 0);
    yoff += (unsigned)W * 32u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = v4[i] * m;
      s4[i] += v;
      q4[i] += v * v;
    }
  }
  if (a.stats) {
    // a row of 16 lanes holds the same four channels (4 g + i) of 16 voxels: four row rotations, then the waves through LDS
    float vals[8] = {s4[0], s4[1], s4[2], s4[3], q4[0], q4[1], q4[2], q4[3]};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = vals[i];
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, true));   // row_ror:1
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, true));   // row_ror:2
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));   // row_ror:4
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8
      vals[i] = v;
    }
    if (li == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave][g][i] = vals[i];
    }
    __syncthreads();
    if (tid < 32) {                                            // channel c = tid >> 1, which = tid & 1 (sum | sum of squares)
      const int c = tid >> 1, which = tid & 1;
      double tot = 0.0;
      for (int wv = 0; wv < 4; ++wv) tot += (double)red[wv][c >> 2][which * 4 + (c & 3)];
      const int slot = blockIdx.x & (FI_STATS_SLOTS - 1);
      atomicAdd(&a.stats[(size_t)n * a.stats_stride + ((size_t)slot * 16 + c) * 2 + which], tot);
    }
  }
}

// ---- the filter gradient on the matrix pipe (round 6): dW[co][tap] = sum over voxels of dy[voxel][co] * x[voxel + tap], a GEMM with
// M = 16 channels, N = 27 taps (two blocks of 16; slot 27 multiplies ones: the bias gradient), K = voxels, 32 per MFMA.  A wave owns
// half of the workgroup's 64 columns and every other row of its run: the 1 KB gradient segment of a K block goes global -> registers ->
// the wave's own LDS plane [voxel][16 ch] and comes back transposed (`ds_read_b64_tr_b16`, the row-streaming kernels' operand read);
// the B fragment -- 8 voxels of one tap's shifted input row -- is 8 two-byte reads of the halo tile the forward uses.  No workgroup
// barrier inside the run.  The four waves' 16 x 32 sums are folded through LDS into the workgroup's partial slice; the second launch
// (conv3d_first_reduce_kernel) is unchanged.
template <typename T>
__global__ __launch_bounds__(256) void conv3d_first_wgrad_mfma_kernel(First3dArgs a) {
  typedef typename FragOf<T>::type frag_t;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  constexpr int TR = FIRST_RY + 2, TP = 68;
  __shared__ __attribute__((aligned(16))) T xt[3 * TR * TP];
  __shared__ __attribute__((aligned(16))) char dplane[4][2][1024];      // [wave][buffer][32 voxels x 32 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int D = a.D, H = a.H, W = a.W;
  const int xblocks = (W + 63) / 64, yruns = (H + FIRST_RY - 1) / FIRST_RY;
  int b = blockIdx.x;
  const int xb = b % xblocks;
  b /= xblocks;
  const int yr = b % yruns;
  b /= yruns;
  const int z = b % D, n = b / D;
  const int y0 = yr * FIRST_RY, y1 = min(H, y0 + FIRST_RY), R = y1 - y0;
  const unsigned vol = (unsigned)a.N * (unsigned)D * (unsigned)H * (unsigned)W;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, vol * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, vol * 32u, 0x00020000);

  first3d_fill_tile<T, TR, TP>(xt, rx, tid, R, n, z, y0, xb * 64, D, H, W);

  const int half = wave & 1, rpar = wave >> 1;                  // this wave: columns half * 32 .. + 31, rows rpar, rpar + 2, ...
  // B fragments: N block 0 = taps li, block 1 = taps 16 + li (27: ones -> the bias gradient; 28 ..: ignored, clamped for the address)
  unsigned boff[2];
  int bkw[2];
  bool ones1;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    int t = blk * 16 + li;
    if (blk == 1) ones1 = t == 27;
    t = t > 26 ? 26 : t;
    const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    boff[blk] = (unsigned)(((kd * TR + kh) * TP + half * 32 + 4 * g) * 2);     // 8-byte aligned: the column shift kw is applied in registers
    bkw[blk] = kw;
  }
  frag_t onesv;
#pragma unroll
  for (int j = 0; j < 8; ++j) onesv[j] = (T)1.0f;
  const int vx = lane >> 1, hh = lane & 1;                       // the gradient segment: lane -> voxel vx, channel half hh (16 B)
  const int gx = xb * 64 + half * 32 + vx;
  const unsigned dpen = gx < W ? 0u : 0x80000000u;
  const unsigned doff = ((unsigned)((((long)n * D + z) * H + y0 + rpar) * W + gx) * 16u + (unsigned)hh * 8u) * 2u;
  const unsigned dstep = (unsigned)W * 32u * 2u;                // two rows on
  const int trd = (4 * g + (li >> 2)) * 32 + (li & 3) * 8;      // the transposing read's lane address inside a plane (wgrad_rows.h)
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  // four gradient segments in flight per wave (16 waves x 4 KB per CU: one segment ahead left the loads latency-bound at 2 TB/s)
  v4u q0 = v4u{0u, 0u, 0u, 0u}, q1 = q0, q2 = q0, q3 = q0;
  auto fetch = [&](int r) __attribute__((always_inline)) {
    const unsigned o = r < R ? (doff + (unsigned)((r - rpar) >> 1) * dstep) | dpen : OOB;
    return __builtin_amdgcn_raw_buffer_load_b128(ry, o, 0, 0);
  };
  q0 = fetch(rpar), q1 = fetch(rpar + 2), q2 = fetch(rpar + 4), q3 = fetch(rpar + 6);
  __syncthreads();                                             // the tile is complete
  char* const myp = &dplane[wave][0][0];
  const char* const tile = reinterpret_cast<const char*>(xt);
  auto kblock = [&](int r, v4u& q, int buf) __attribute__((always_inline)) {
    *reinterpret_cast<v4u*>(myp + buf * 1024 + lane * 16) = q;
    q = fetch(r + 8);
    // A: channel li of voxels 4 g .. 4 g + 3 and 16 + 4 g .. + 3 of the segment
    typedef __attribute__((address_space(3))) s16x4 lds_v;
    union {
      s16x4 h[2];
      frag_t v;
    } au;
    au.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(myp + buf * 1024 + trd));
    au.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(myp + buf * 1024 + trd + 16 * 32));
    // B: voxels 4 g + kw .. + 3 and 16 + 4 g + kw .. + 3 of the tap's row.  Read as ALIGNED 16-byte windows and shifted by kw elements in
    // registers: 8-byte LDS reads at 2 or 4 (mod 8) work but cost ~10 x (measured: 80 us against 41 with every kw forced to 0)
    auto shifted = [&](const char* p, int kw) __attribute__((always_inline)) {
      const uint2 w0 = *reinterpret_cast<const uint2*>(p), w1 = *reinterpret_cast<const uint2*>(p + 8);
      const uint2 w2 = *reinterpret_cast<const uint2*>(p + 32), w3 = *reinterpret_cast<const uint2*>(p + 40);
      const bool two = kw == 2;
      const unsigned sh = kw == 1 ? 2u : 0u;
      const unsigned a0 = two ? w0.y : w0.x, a1 = two ? w1.x : w0.y, a2 = two ? w1.y : w1.x;
      const unsigned c0 = two ? w2.y : w2.x, c1 = two ? w3.x : w2.y, c2 = two ? w3.y : w3.x;
      union {
        unsigned u[4];
        frag_t v;
      } o;
      o.u[0] = __builtin_amdgcn_alignbyte(a1, a0, sh);
      o.u[1] = __builtin_amdgcn_alignbyte(a2, a1, sh);
      o.u[2] = __builtin_amdgcn_alignbyte(c1, c0, sh);
      o.u[3] = __builtin_amdgcn_alignbyte(c2, c1, sh);
      return o.v;
    };
    frag_t b0 = shifted(tile + boff[0] + (unsigned)(r * TP * 2), bkw[0]);
    frag_t b1 = shifted(tile + boff[1] + (unsigned)(r * TP * 2), bkw[1]);
    if (ones1) b1 = onesv;
    acc0 = mfma16(au.v, b0, acc0);
    acc1 = mfma16(au.v, b1, acc1);
  };
  int r = rpar;
  for (; r + 6 < R; r += 8) {
    kblock(r, q0, 0);
    kblock(r + 2, q1, 1);
    kblock(r + 4, q2, 0);
    kblock(r + 6, q3, 1);
  }
  if (r < R) kblock(r, q0, 0);
  if (r + 2 < R) kblock(r + 2, q1, 1);
  if (r + 4 < R) kblock(r + 4, q2, 0);
  // ---- fold the four waves: red[wave][co][32 taps]; D[row = co = 4 g + i][col = tap = li]
  __syncthreads();                                             // every wave is done with the tile: its space is the buffer
  float* const red = reinterpret_cast<float*>(xt);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    red[(wave * 16 + 4 * g + i) * 32 + li] = acc0[i];
    red[(wave * 16 + 4 * g + i) * 32 + 16 + li] = acc1[i];
  }
  __syncthreads();
  float* const slice = a.part + (size_t)blockIdx.x * (16 * 27 + 16);
  for (int e = tid; e < 16 * 28; e += 256) {
    const int c = e / 28, t = e - c * 28;
    const float tot = red[(0 * 16 + c) * 32 + t] + red[(1 * 16 + c) * 32 + t] + red[(2 * 16 + c) * 32 + t] + red[(3 * 16 + c) * 32 + t];
    slice[t < 27 ? c * 27 + t : 16 * 27 + c] = tot;
  }
}

// dw[e] += sum over the slices, in slice order (fp32 partials of <= 64 rows x 64 columns each; fp64 sum): deterministic
__global__ __launch_bounds__(64) void conv3d_first_reduce_kernel(const float* part, int slices, float* dw, float* dbias) {
  const int e = blockIdx.x, n = 16 * 27 + 16;
  double tot = 0.0;
  for (int s = threadIdx.x; s < slices; s += 64) tot += (double)part[(size_t)s * n + e];
  tot = wave_sum(tot);
  if (threadIdx.x == 0) {
    if (e < 16 * 27) {
      if (dw) dw[e] += (float)tot;
    } else if (dbias) {
      dbias[e - 16 * 27] += (float)tot;
    }
  }
}

inline long first3d_blocks(int N, int D, int H, int W) {
  return (long)N * D * ((H + FIRST_RY - 1) / FIRST_RY) * ((W + 63) / 64);
}

}  // namespace

extern "C" long fi_conv3d_first_wgrad_workspace(int N, int D, int H, int W) {
  return first3d_blocks(N, D, H, W) * (16 * 27 + 16) * (long)sizeof(float);
}

extern "C" int fi_conv3d_first_fwd(int dtype, int N, int D, int H, int W, const void* x, const float* w, const float* bias,
                                   void* y, double* stats, long stats_stride, void* stream) {
  if (!x || !w || !y) return FI_ERR_NULL;
  if (dtype != FI_BF16 && dtype != FI_F16) return FI_ERR_UNSUPPORTED;
  if (N < 1 || D < 1 || H < 1 || W < 1 || (long)N * D * H * W * 32 >= (1L << 31) - 64) return FI_ERR_SHAPE;
  First3dArgs a{x, w, bias, y, stats, stats_stride, nullptr, N, D, H, W};
  const dim3 g((unsigned)first3d_blocks(N, D, H, W)), b(256);
  if (dtype == FI_BF16)
    hipLaunchKernelGGL((conv3d_first_fwd_mfma_kernel<bf16_t>), g, b, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((conv3d_first_fwd_mfma_kernel<f16_t>), g, b, 0, (hipStream_t)stream, a);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_conv3d_first_wgrad(int dtype, int N, int D, int H, int W, const void* x, const void* dy, float* dw, float* dbias,
                                     void* workspace, long workspace_bytes, void* stream) {
  if (!x || !dy || !workspace || (!dw && !dbias)) return FI_ERR_NULL;
  if (dtype != FI_BF16 && dtype != FI_F16) return FI_ERR_UNSUPPORTED;
  if (N < 1 || D < 1 || H < 1 || W < 1 || (long)N * D * H * W * 32 >= (1L << 31) - 64) return FI_ERR_SHAPE;
  if (workspace_bytes < fi_conv3d_first_wgrad_workspace(N, D, H, W)) return FI_ERR_SHAPE;
  First3dArgs a{x, nullptr, nullptr, const_cast<void*>(dy), nullptr, 0, (float*)workspace, N, D, H, W};
  const long blocks = first3d_blocks(N, D, H, W);
  if (dtype == FI_BF16)
    hipLaunchKernelGGL((conv3d_first_wgrad_mfma_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((conv3d_first_wgrad_mfma_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(conv3d_first_reduce_kernel, dim3(16 * 27 + 16), dim3(64), 0, (hipStream_t)stream, (const float*)workspace,
                     (int)blocks, dw, dbias);
  FI_CHECK_LAUNCH();
  return 0;
}
