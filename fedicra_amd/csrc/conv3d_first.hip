// The FIRST convolution of the 3D U-Net: Conv3d(1 -> 16, 3x3x3, pad 1) on full-resolution volumes
// (/root/reference/code/networks/unet_3D.py:38 `UnetConv3(in_channels, filters[0])`, networks/utils.py:99-123; BASELINE configs[3]/[4]:
// 2 x 1 x 128^3).  With ONE input channel the contraction is 27 long: the implicit-GEMM forms pad it to 32 per depth tap and run
// three read-modify-write passes over the 134 MB output (259 us forward, 229 us filter gradient per 2 x 128^3 batch: 0.07 of
// their HBM roofline, 6.7 % of a unet_3D iteration).  It is not GEMM-shaped work: 27 multiply-adds per output element against
// 2 bytes stored -- a vector-ALU stencil that has to stream the output once.  Both directions here:
//
//   lane = 4 * vx + q: sixteen consecutive voxels of an x row, four lanes (channel quads q) per voxel -- a wave's store of a row
//     segment is 512 contiguous bytes (16 voxels x 16 channels x 2 B);
//   a thread walks DOWN the rows of one slice with the 3 x 3 x 3 neighbourhood of its voxel in registers: a step issues the nine
//     loads of the row AFTER next (three slices x three columns, range-checked raw buffer loads: out-of-image = 0, no branches,
//     L1 / L2 hits -- the 8 MB input volume is read 27 x 4 times out of cache) and rotates the four-row window by loop unrolling;
//   forward:  4 accumulators, 27 x 4 weights in registers, v_pk_fma_f32 pairs; bias, rounding to the storage type, ONE 8-byte
//     store per voxel and quad; InstanceNorm's per-sample statistics of the values AS STORED in registers over the whole run,
//     one fp64 atomic per channel and workgroup;
//   filter gradient: the mirror image, one DEPTH TAP per workgroup -- per step the quad's 4 gradient values x the 9 window values of
//     its input slice into 36 accumulators; per run ONE partial slice [16][27] (+ the bias gradient) filled by its three workgroups,
//     summed over the runs by a second, tiny launch in a fixed order (deterministic, like fi_wgrad_reduce_multi's slices).
#include "common.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ float ld16(const __amdgpu_buffer_rsrc_t& r, unsigned off);
template <> __device__ __forceinline__ float ld16<bf16_t>(const __amdgpu_buffer_rsrc_t& r, unsigned off) {
  return __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0) << 16);
}
template <> __device__ __forceinline__ float ld16<f16_t>(const __amdgpu_buffer_rsrc_t& r, unsigned off) {
  const unsigned short h = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0);
  return (float)__builtin_bit_cast(f16_t, h);
}

constexpr int FIRST_RY = 64;          // rows of a slice one workgroup walks (a run); 256 threads = 64 columns x 4 channel quads
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

// the window: win[kd][row][kw], row = image row relative to the current one (0: y - 1, 1: y, 2: y + 1), rotated by R
template <typename T, bool WGRAD>
__global__ __launch_bounds__(256, 2) void conv3d_first_kernel(First3dArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane & 3, vx = lane >> 2;
  const int D = a.D, H = a.H, W = a.W;
  const int xblocks = (W + 63) / 64, yruns = (H + FIRST_RY - 1) / FIRST_RY;
  int b = blockIdx.x;
  const int xb = b % xblocks;
  b /= xblocks;
  const int yr = b % yruns;
  b /= yruns;
  const int z = b % D, n = b / D;
  const int x = xb * 64 + wave * 16 + vx;
  const int y0 = yr * FIRST_RY, y1 = min(H, y0 + FIRST_RY);
  const bool xin = x < W;

  const unsigned vol = (unsigned)a.N * (unsigned)D * (unsigned)H * (unsigned)W;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, vol * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, vol * 32u, 0x00020000);

  // per-thread constants: byte offsets of the three slices' rows are formed per step; column validity is per lane
  // FOUR row slots: a step computes on rows y - 1 .. y + 1 while the nine loads of row y + 2 are in flight (a row's loads have a
  // whole step -- ~150 vector instructions -- to land).  No condition sits at a load: hipcc turns a wave-uniform one into a BRANCH
  // around the load with vmcnt(0) inside (the first build of this kernel: 120 us instead of ~50) and a per-lane one into exec-masked
  // blocks with duplicated loads.  Every out-of-volume case is a PENALTY bit instead: offsets are below 2^31 (host check), so
  // or-ing 0x80000000 into one puts it beyond the resource's range and the hardware returns zero -- per-thread constants for the
  // column neighbours, per-slice constants for z, and for the row a sign-bit trick on (yy, H - 1 - yy).
  const unsigned PEN = 0x80000000u;
  unsigned xpen[3], zpen[3];
  xpen[0] = (xin && x - 1 >= 0) ? 0u : PEN;
  xpen[1] = xin ? 0u : PEN;
  xpen[2] = (xin && x + 1 < W) ? 0u : PEN;
  long zrow[3];
#pragma unroll
  for (int kd = 0; kd < 3; ++kd) {
    const int zz = z + kd - 1;
    zpen[kd] = (unsigned)zz < (unsigned)D ? 0u : PEN;
    zrow[kd] = ((long)n * D + zz) * H;
  }
  float win[3][4][3];
  auto load_row = [&](int yy, int slot) __attribute__((always_inline)) {
    const unsigned ypen = ((unsigned)((yy | (H - 1 - yy)) >> 31)) << 31;      // yy < 0 or yy > H - 1: the sign bit, shifted back up
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const unsigned base = (unsigned)((zrow[kd] + yy) * W + x) * 2u;
      const unsigned pen = ypen | zpen[kd];
      win[kd][slot][0] = ld16<T>(rx, (base - 2u) | pen | xpen[0]);
      win[kd][slot][1] = ld16<T>(rx, base | pen | xpen[1]);
      win[kd][slot][2] = ld16<T>(rx, (base + 2u) | pen | xpen[2]);
    }
  };

  if constexpr (!WGRAD) {
    // ------------------------------------------------------------------------------------------------ forward
    f2 w01[27], w23[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      w01[t] = f2{a.w[(q * 4 + 0) * 27 + t], a.w[(q * 4 + 1) * 27 + t]};
      w23[t] = f2{a.w[(q * 4 + 2) * 27 + t], a.w[(q * 4 + 3) * 27 + t]};
    }
    f2 b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
    if (a.bias) b01 = f2{a.bias[q * 4], a.bias[q * 4 + 1]}, b23 = f2{a.bias[q * 4 + 2], a.bias[q * 4 + 3]};
    f2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
    load_row(y0 - 1, 0);
    load_row(y0, 1);
    load_row(y0 + 1, 2);
    auto step = [&](int y, auto rot) __attribute__((always_inline)) {
      constexpr int R = decltype(rot)::value;                  // window slot of row y - 1
      load_row(y + 2, (R + 3) % 4);                            // one step ahead
      f2 a01 = b01, a23 = b23;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float v = win[kd][(R + kh) % 4][kw];
            const int t = (kd * 3 + kh) * 3 + kw;
            a01 += w01[t] * f2{v, v};
            a23 += w23[t] * f2{v, v};
          }
      float v4[4] = {a01.x, a01.y, a23.x, a23.y};
      const typename Quad<T>::q_t st = Quad<T>::pack(v4);       // v4 := the values as stored
      const unsigned off = ((unsigned)((((long)n * D + z) * H + y) * W + x) * 16u + (unsigned)q * 4u) * 2u;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, st), ry, off | xpen[1], 0, 0);
      const float m = xin ? 1.f : 0.f;
      const f2 r01 = f2{v4[0], v4[1]} * f2{m, m}, r23 = f2{v4[2], v4[3]} * f2{m, m};
      s01 += r01, s23 += r23;
      q01 += r01 * r01, q23 += r23 * r23;
    };
    int y = y0;
    for (; y + 3 < y1; y += 4) {
      step(y, std::integral_constant<int, 0>());
      step(y + 1, std::integral_constant<int, 1>());
      step(y + 2, std::integral_constant<int, 2>());
      step(y + 3, std::integral_constant<int, 3>());
    }
    if (y < y1) step(y, std::integral_constant<int, 0>());
    if (y + 1 < y1) step(y + 1, std::integral_constant<int, 1>());
    if (y + 2 < y1) step(y + 2, std::integral_constant<int, 2>());
    if (a.stats) {
      // lanes of one quad class are 4 apart: two row rotations sum a 16-lane row's four voxels-per-class ... (x 4 voxels) -- then LDS
      __shared__ float red[4][4][4][8];                        // [wave][row of 16 lanes][quad][8 values]
      float vals[8] = {s01.x, s01.y, s23.x, s23.y, q01.x, q01.y, q23.x, q23.y};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float v = vals[i];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));   // row_ror:4
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8
        vals[i] = v;
      }
      if ((lane & 15) < 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[wave][lane >> 4][lane & 3][i] = vals[i];
      }
      __syncthreads();
      if (tid < 32) {                                          // channel c = tid >> 1, which = tid & 1 (sum | sum of squares)
        const int c = tid >> 1, which = tid & 1;
        double tot = 0.0;
        for (int wv = 0; wv < 4; ++wv)
          for (int r = 0; r < 4; ++r) tot += (double)red[wv][r][c >> 2][which * 4 + (c & 3)];
        const int slot = blockIdx.x & (FI_STATS_SLOTS - 1);
        atomicAdd(&a.stats[(size_t)n * a.stats_stride + ((size_t)slot * 16 + c) * 2 + which], tot);
      }
    }
  } else {
    // ------------------------------------------------------------------------------------------------ filter gradient
    // ONE depth tap per workgroup (blockIdx.y = kd): 9 taps x 4 channels = 36 accumulators per thread instead of 108 -- the whole
    // 27-tap form needed 379 registers (one wave per SIMD) and ran 289 us against the per-tap GEMMs' 229.  The three workgroups of a
    // run read the same gradient rows (L2) and one input slice each, and write disjoint thirds of the run's partial slice.
    const int kd = blockIdx.y;
    f2 g01[9], g23[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) g01[t] = g23[t] = f2{0.f, 0.f};
    f2 gb01 = {0.f, 0.f}, gb23 = {0.f, 0.f};
    const unsigned zp = zpen[0] * (kd == 0) | zpen[1] * (kd == 1) | zpen[2] * (kd == 2);     // (penalties are 0 or 2^31: a select without a branch)
    const long zr = kd == 0 ? zrow[0] : (kd == 1 ? zrow[1] : zrow[2]);
    float w3[4][3];
    auto load_row1 = [&](int yy, int slot) __attribute__((always_inline)) {
      const unsigned ypen = ((unsigned)((yy | (H - 1 - yy)) >> 31)) << 31;
      const unsigned base = (unsigned)((zr + yy) * W + x) * 2u;
      const unsigned pen = ypen | zp;
      w3[slot][0] = ld16<T>(rx, (base - 2u) | pen | xpen[0]);
      w3[slot][1] = ld16<T>(rx, base | pen | xpen[1]);
      w3[slot][2] = ld16<T>(rx, (base + 2u) | pen | xpen[2]);
    };
    auto load_dy = [&](int yy) __attribute__((always_inline)) {
      const unsigned off = ((unsigned)((((long)n * D + z) * H + yy) * W + x) * 16u + (unsigned)q * 4u) * 2u;
      const unsigned ypen = ((unsigned)((y1 - 1 - yy) >> 31)) << 31;           // the row behind the run: zeros (never accumulated twice)
      return __builtin_amdgcn_raw_buffer_load_b64(ry, off | ypen | xpen[1], 0, 0);
    };
    load_row1(y0 - 1, 0);
    load_row1(y0, 1);
    load_row1(y0 + 1, 2);
    v2u dnext = load_dy(y0);
    auto step = [&](int y, auto rot) __attribute__((always_inline)) {
      constexpr int R = decltype(rot)::value;
      load_row1(y + 2, (R + 3) % 4);                           // one step ahead, like the gradient row below
      const v2u raw = dnext;
      dnext = load_dy(y + 1);
      float d4[4];
      Quad<T>::unpack(__builtin_bit_cast(typename Quad<T>::q_t, raw), d4);
      const f2 d01 = {d4[0], d4[1]}, d23 = {d4[2], d4[3]};
      gb01 += d01, gb23 += d23;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float v = w3[(R + kh) % 4][kw];
          g01[kh * 3 + kw] += d01 * f2{v, v};
          g23[kh * 3 + kw] += d23 * f2{v, v};
        }
    };
    int y = y0;
    for (; y + 3 < y1; y += 4) {
      step(y, std::integral_constant<int, 0>());
      step(y + 1, std::integral_constant<int, 1>());
      step(y + 2, std::integral_constant<int, 2>());
      step(y + 3, std::integral_constant<int, 3>());
    }
    if (y < y1) step(y, std::integral_constant<int, 0>());
    if (y + 1 < y1) step(y + 1, std::integral_constant<int, 1>());
    if (y + 2 < y1) step(y + 2, std::integral_constant<int, 2>());
    // sum over the 16 voxel columns of a wave (lanes of one quad class: 4 apart within a row of 16 -- two row rotations -- then
    // the four rows and the four waves through LDS); thread e writes element e of this depth tap's third of the slice
    __shared__ float red[4][4][4][40];                         // [wave][row][quad][9 taps x 4 channels + 4 bias]
    auto fold = [&](float v) __attribute__((always_inline)) {
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
      return v;
    };
    const bool writer = (lane & 15) < 4;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float v0 = fold(g01[t].x), v1 = fold(g01[t].y), v2 = fold(g23[t].x), v3 = fold(g23[t].y);
      if (writer) {
        float* dst = &red[wave][lane >> 4][lane & 3][t * 4];
        dst[0] = v0, dst[1] = v1, dst[2] = v2, dst[3] = v3;
      }
    }
    {
      const float v0 = fold(gb01.x), v1 = fold(gb01.y), v2 = fold(gb23.x), v3 = fold(gb23.y);
      if (writer) {
        float* dst = &red[wave][lane >> 4][lane & 3][36];
        dst[0] = v0, dst[1] = v1, dst[2] = v2, dst[3] = v3;
      }
    }
    __syncthreads();
    float* const slice = a.part + (size_t)blockIdx.x * (16 * 27 + 16);
    for (int e = tid; e < 16 * 9 + 16; e += 256) {
      int c, idx, pos;                                         // channel, index inside the quad's 40-value record, slice position
      if (e < 16 * 9) {
        c = e / 9;
        const int t9 = e - c * 9;
        idx = t9 * 4 + (c & 3);
        pos = c * 27 + kd * 9 + t9;
      } else {
        if (kd != 1) break;                                    // the bias gradient once per run: by the centre tap's workgroup
        c = e - 16 * 9;
        idx = 36 + (c & 3);
        pos = 16 * 27 + c;
      }
      float tot = 0.f;
      for (int wv = 0; wv < 4; ++wv)
        for (int r = 0; r < 4; ++r) tot += red[wv][r][c >> 2][idx];
      slice[pos] = tot;
    }
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
    hipLaunchKernelGGL((conv3d_first_kernel<bf16_t, false>), g, b, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((conv3d_first_kernel<f16_t, false>), g, b, 0, (hipStream_t)stream, a);
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
  const dim3 g((unsigned)blocks, 3), b(256);                  // y: the depth tap
  if (dtype == FI_BF16)
    hipLaunchKernelGGL((conv3d_first_kernel<bf16_t, true>), g, b, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((conv3d_first_kernel<f16_t, true>), g, b, 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(conv3d_first_reduce_kernel, dim3(16 * 27 + 16), dim3(64), 0, (hipStream_t)stream, (const float*)workspace,
                     (int)blocks, dw, dbias);
  FI_CHECK_LAUNCH();
  return 0;
}
