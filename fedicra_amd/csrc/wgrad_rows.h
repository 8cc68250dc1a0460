// Row-streaming filter gradient for the THIN layers (16 / 32 channels a side at 512^2 and 256^2; gfx950 only).
//
// dW[co][kr][kc][ci] = sum_{n,r,c} dy[n][r][c][co] * x[n][r + kr - 1][c + kc - 1][ci]   (/root/reference/code/networks/unet.py:14-30:
// the weight gradient of ConvBlock's 3x3 convolutions; dbias = sum dy).  At these widths the layer is HBM-bound -- 64 ... 96
// bytes per pixel against 9 x 16 x 16 ... 32 multiply-adds -- and the tile kernels (conv_wgrad_kernel) sit at a quarter of
// the roofline: a 16 x 16-pixel tile is sixteen 512-byte row segments 16 KB apart, one tile of loads in flight per workgroup,
// two barriers and an LDS commit per 256 pixels.  Here a workgroup STREAMS image rows:
//   * it owns a run of x rows of one image strip (<= 256 columns); x row rho meets the three dy rows rho - 1, rho, rho + 1 (filter
//     rows kr = 2, 1, 0), so per 32-pixel K step three x operands (the column shifts kc) and three dy operands feed nine MFMAs
//     (v_mfma_f32_16x16x32) per 16 x 16 channel block -- a third of the LDS reads of the dy-major order;
//   * rows arrive as whole contiguous segments (8 ... 16 KB per row and tensor), one row of x and one of dy per step, staged
//     through TWO register stages (a row's loads are issued two steps before its store) into a 2-row (x) / 4-row (dy) LDS
//     ring: ONE barrier per row; 2-3 workgroups per CU keep several rows of loads in flight;
//   * LDS holds channel-BLOCK planes [block][pixel][16 ch] (32 bytes per pixel): the 32 lanes a transposing read serves at
//     once cover 8 consecutive pixels = 256 contiguous bytes (the K order inside an operand is permuted for that: the same
//     permutation on both operands), no padding, no conflicts;
//   * the four waves split the K steps of a row; their accumulators (9 taps x blocks x 4 registers) are folded through LDS once
//     per workgroup and written as ONE partial slice in the layout of conv_wgrad_kernel, which fi_wgrad_reduce_multi folds.
#pragma once
#include "conv_impl.h"
#include <type_traits>

struct WgRowsArgs {
  const void* x0;
  const void* x1;
  const void* dy;
  float* part;             // [items][part_stride]
  size_t part_stride;
  int N, H, W;
  int c0, c1;              // input channels from x0 / x1 (multiples of 16)
  int ws, strips, rpw, chunks;     // strip width, strips per row, x rows per item, items per (image, strip)
  int want_bias;
  int depth;               // conv_wgrad_rows3d_kernel: slices per volume (an "image" is a slice)
  int cd;                  // NARROW == 2: channels of the gradient tensor (<= 4); NARROW == 1: c0 <= 4 is the input's, c1 = 0
  int cout, nct, nit;      // conv_wgrad_rows64_kernel: gradient channels, gradient / input channel tiles per item
#ifdef FI_TRACE
  long long* trace;        // conv_wgrad_rows3d_kernel: [workgroup][wave][16 rows][8] s_memtime stamps (tools/rows3d_trace.py); debug builds only
#endif
};
#ifdef FI_TRACE
#define FI_TROW(slot)                                                                                                            \
  do {                                                                                                                           \
    if (a.trace && lane == 0 && rho - r0 >= 16 && rho - r0 < 32)                                                                 \
      a.trace[(((size_t)blockIdx.x * 4 + wave) * 16 + (rho - r0 - 16)) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define FI_TROW(slot) do { } while (0)
#endif

template <typename T>
__device__ __forceinline__ typename DT<T>::frag_t wgr_frag(const char* addr) {
  typedef __attribute__((address_space(3))) s16x4_t lds_v;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)addr);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(addr + 16 * 32));     // pixels + 16
  union {
    s16x4_t h[2];
    typename DT<T>::frag_t v;
  } u;
  u.h[0] = lo;
  u.h[1] = hi;
  return u.v;
}

// NCI / NCO: 16-channel blocks of the input / gradient side.  WSP = ws + 2 (x row with its two halo pixels).
// NARROW (NCI = NCO = 1): one side is a dense tensor of <= 4 channels -- 1: the input (the network's first convolution, 1 / 3 -> 16),
// 2: the gradient (the logits convolution, 16 -> n_class).  Its rows are staged element by element into the first 8 bytes of each
// pixel's 32-byte block of the plane; the other 24 stay zero from the start of the workgroup, the MFMA schedule is the same, and
// only the real rows / columns of the filter gradient are written to the slice (layout [co][9][ci] over the REAL channel counts).
template <typename T, int NCI, int NCO, int NARROW = 0>
__global__ __launch_bounds__(256) void conv_wgrad_rows_kernel(WgRowsArgs a) {
  static_assert(NARROW == 0 || (NCI == 1 && NCO == 1), "narrow sides: one block each");
  typedef typename DT<T>::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W, ws = a.ws;
  const int xplane = (ws + 2) * 32, dplane = ws * 32;
  const int xrow = NCI * xplane, drow = NCO * dplane;
  char* const xs = smem;                        // [2][NCI][ws + 2][16]
  char* const ds = smem + 2 * xrow;             // [4][NCO][ws][16]
  // item -> (image, strip, row chunk)
  int item = blockIdx.x;
  const int chunk = item % a.chunks;
  item /= a.chunks;
  const int strip = item % a.strips, n = item / a.strips;
  const int r0 = chunk * a.rpw, r1 = min(r0 + a.rpw, H);
  const int cs = strip * ws;
  const T* const x0 = reinterpret_cast<const T*>(a.x0) + (size_t)n * H * W * a.c0;
  const T* const x1 = reinterpret_cast<const T*>(a.x1) + (size_t)n * H * W * a.c1;
  const T* const dyg = reinterpret_cast<const T*>(a.dy) + (size_t)n * H * W * (NARROW == 2 ? a.cd : NCO * 16);
  // narrow sides: 16-bit raw buffer loads over this image's plane (an element the lane does not have: out-of-range offset -> zero;
  // predicated plain loads would become branches that serialise the round trips)
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const __amdgpu_buffer_rsrc_t rnx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(x0), 0, NARROW == 1 ? (unsigned)H * (unsigned)W * (unsigned)a.c0 * 2u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rnd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(dyg), 0, NARROW == 2 ? (unsigned)H * (unsigned)W * (unsigned)a.cd * 2u : 0u, 0x00020000);
  auto narrow_px = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned pix, int nch, bool ok) __attribute__((always_inline)) {
    unsigned e[4];                                 // <= 4 elements of pixel `pix` of the plane -> 8 bytes
#pragma unroll
    for (int c = 0; c < 4; ++c) e[c] = __builtin_amdgcn_raw_buffer_load_b16(rs, (ok && c < nch) ? (pix * (unsigned)nch + c) * 2u : OOB, 0, 0);
    return make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), 0u, 0u);
  };

  // PAIR (32 -> 32): a wave owns one (gradient block, input block) pair for every K step of the row -- 9 accumulators instead of 36, no
  // cross-wave sum, strips of <= 128 columns: three workgroups per CU where the K-split form had one (328 registers, 98 KB)
  constexpr bool PAIR = NCI == 2 && NCO == 2;
  // ---- staging: vectors of 8 channels.  x row: (ws + 2) pixels x 2 NCI vectors; dy row: ws pixels x 2 NCO vectors
  constexpr int MAXW = PAIR ? 128 : 256;
  constexpr int NXV = ((MAXW + 2) * 2 * NCI + 255) / 256, NDV = (MAXW * 2 * NCO + 255) / 256;
  const int nxv = (ws + 2) * 2 * NCI, ndv = ws * 2 * NCO;
  uint4 xrA[NXV], drA[NDV], xrB[NXV], drB[NDV];      // two register stages: a row's loads are issued two steps before its store
  // Per thread and vector, fixed for the whole run and computed once: the source at row 0 (column and channel folded in, columns outside
  // the image clamped to a valid address), the row pitch, the place in the LDS plane, a bit for "inside the strip" / "inside the
  // image".  A row step is then one multiply-add per load and a scalar row test (the per-row address arithmetic was a quarter of a
  // step of the 3D kernel: tools/rows3d_trace.py).  Lanes outside the image are zeroed when the row is STORED to LDS, two steps after
  // its loads were issued: a select at the load is a use the compiler places -- with its s_waitcnt vmcnt -- at the end of the same step.
  const char* xsrc[NXV];
  unsigned xpitch[NXV], xdst[NXV], dsrc[NDV], ddst[NDV], xin = 0, xcol = 0, din = 0;
  if constexpr (NARROW != 1) {
#pragma unroll
    for (int it = 0; it < NXV; ++it) {
      const int i = tid + it * 256;
      const int v = i % (2 * NCI), px = i / (2 * NCI);
      const int gx = cs + px - 1, ch = v * 8;
      const bool in = i < nxv, colok = in && gx >= 0 && gx < W;
      const bool first = ch < a.c0;
      xsrc[it] = reinterpret_cast<const char*>(first ? x0 + (size_t)(colok ? gx : 0) * a.c0 + ch : x1 + (size_t)(colok ? gx : 0) * a.c1 + (ch - a.c0));
      xpitch[it] = (unsigned)((first ? a.c0 : a.c1) * W * (int)sizeof(T));
      xdst[it] = (unsigned)((v >> 1) * xplane + px * 32 + (v & 1) * 16);
      xin |= (unsigned)in << it;
      xcol |= (unsigned)colok << it;
    }
  }
  if constexpr (NARROW != 2) {
#pragma unroll
    for (int it = 0; it < NDV; ++it) {
      const int i = tid + it * 256;
      const int v = i % (2 * NCO), px = i / (2 * NCO);
      const bool in = i < ndv;
      dsrc[it] = (unsigned)(((cs + (in ? px : 0)) * (NCO * 16) + v * 8) * (int)sizeof(T));
      ddst[it] = (unsigned)((v >> 1) * dplane + px * 32 + (v & 1) * 16);
      din |= (unsigned)in << it;
    }
  }
  auto load_x = [&](uint4 (&xr)[NXV], int rho) {                  // x row rho, columns cs - 1 .. cs + ws
    if constexpr (NARROW == 1) {
      const bool rowok = rho >= 0 && rho < H;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int px = tid + it * 256, gx = cs + px - 1;
        const bool ok = rowok && px < ws + 2 && gx >= 0 && gx < W;
        xr[it] = narrow_px(rnx, (unsigned)(rho * W + gx), a.c0, ok);
      }
      return;
    }
    const unsigned rc = (unsigned)min(max(rho, 0), H - 1);
#pragma unroll
    for (int it = 0; it < NXV; ++it) xr[it] = *reinterpret_cast<const uint4*>(xsrc[it] + (size_t)rc * xpitch[it]);
  };
  auto load_d = [&](uint4 (&dr)[NDV], int r) {                    // dy row r, columns cs .. cs + ws - 1
    if constexpr (NARROW == 2) {
      const bool ok = r >= 0 && r < H && tid < ws;
      dr[0] = narrow_px(rnd, (unsigned)(r * W + cs + tid), a.cd, ok);
      return;
    }
    const char* const base = reinterpret_cast<const char*>(dyg + (size_t)min(max(r, 0), H - 1) * W * (NCO * 16));     // uniform
#pragma unroll
    for (int it = 0; it < NDV; ++it) dr[it] = *reinterpret_cast<const uint4*>(base + dsrc[it]);
  };
  auto store_x = [&](const uint4 (&xr)[NXV], int rho) {           // the registers hold x row rho
    char* const dst = xs + (rho & 1) * xrow;
    if constexpr (NARROW == 1) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int px = tid + it * 256;
        if (px < ws + 2) *reinterpret_cast<uint2*>(dst + px * 32) = make_uint2(xr[it].x, xr[it].y);
      }
      return;
    }
    const bool rowok = rho >= 0 && rho < H;
#pragma unroll
    for (int it = 0; it < NXV; ++it)
      if ((xin >> it) & 1u) *reinterpret_cast<uint4*>(dst + xdst[it]) = fi_vec_select(rowok && ((xcol >> it) & 1u), xr[it]);
  };
  auto store_d = [&](const uint4 (&dr)[NDV], int r) {             // the registers hold dy row r
    char* const dst = ds + (r & 3) * drow;
    if constexpr (NARROW == 2) {
      if (tid < ws) *reinterpret_cast<uint2*>(dst + tid * 32) = make_uint2(dr[0].x, dr[0].y);
      return;
    }
    const bool rowok = r >= 0 && r < H;
#pragma unroll
    for (int it = 0; it < NDV; ++it)
      if ((din >> it) & 1u) *reinterpret_cast<uint4*>(dst + ddst[it]) = fi_vec_select(rowok, dr[it]);
  };

  constexpr int AO = PAIR ? 1 : NCO, AI = PAIR ? 1 : NCI;      // blocks a wave accumulates
  f32x4 acc[3][3][AO][AI];
  f32x4 accb[AO];
#pragma unroll
  for (int kr = 0; kr < 3; ++kr)
#pragma unroll
    for (int kc = 0; kc < 3; ++kc)
#pragma unroll
      for (int o = 0; o < AO; ++o)
#pragma unroll
        for (int i = 0; i < AI; ++i) acc[kr][kc][o][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < AO; ++o) accb[o] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int po = PAIR ? wave >> 1 : 0, pi = PAIR ? wave & 1 : 0;      // PAIR: this wave's blocks
  const frag_t onesv = WgFrag<T>::ones();

  // operand address of a lane: pixel 4 g + (li >> 2) of the K step (+ 16 for the second read), channels 4 (li & 3) .. + 3
  const int g = lane >> 4, li = lane & 15;
  const int laneoff = (4 * g + (li >> 2)) * 32 + (li & 3) * 8;

  if constexpr (NARROW != 0) {                  // the channels a narrow side does not have: zero for the life of the workgroup
    for (int i = tid * 16; i < 2 * xrow + 4 * drow; i += 256 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
  }
  // prologue: x row r0 and dy rows r0 - 1, r0, r0 + 1 into the ring; stage A <- (x r0 + 1, dy r0 + 2), stage B <- (x r0 + 2, dy r0 + 3)
  load_x(xrA, r0);
  load_d(drA, r0 - 1);
  load_d(drB, r0);
  store_x(xrA, r0);
  store_d(drA, r0 - 1);
  load_d(drA, r0 + 1);
  store_d(drB, r0);
  store_d(drA, r0 + 1);
  load_x(xrA, r0 + 1);
  load_d(drA, r0 + 2);
  load_x(xrB, r0 + 2);
  load_d(drB, r0 + 3);
  __syncthreads();

  const int nks = ws / 32;
  auto step = [&](int rho, uint4 (&xr)[NXV], uint4 (&dr)[NDV]) __attribute__((always_inline)) {
    store_x(xr, rho + 1);                       // x row rho + 1, dy row rho + 2: slots nobody reads during this step
    store_d(dr, rho + 2);
    load_x(xr, rho + 3);                        // back in two steps
    load_d(dr, rho + 4);
    const char* const xb = xs + (rho & 1) * xrow + pi * xplane + laneoff;
    const char* db[3];
#pragma unroll
    for (int kr = 0; kr < 3; ++kr) db[kr] = ds + ((rho - kr + 1) & 3) * drow + po * dplane + laneoff;
    for (int ks = PAIR ? 0 : wave; ks < nks; ks += PAIR ? 1 : 4) {
      frag_t av[3][AO], bv[3][AI];
#pragma unroll
      for (int kr = 0; kr < 3; ++kr)
#pragma unroll
        for (int o = 0; o < AO; ++o) av[kr][o] = wgr_frag<T>(db[kr] + o * dplane + ks * 32 * 32);
#pragma unroll
      for (int kc = 0; kc < 3; ++kc)
#pragma unroll
        for (int i = 0; i < AI; ++i) bv[kc][i] = wgr_frag<T>(xb + i * xplane + (ks * 32 + kc) * 32);
      if (a.want_bias && pi == 0) {
#pragma unroll
        for (int o = 0; o < AO; ++o) accb[o] = mfma16(av[1][o], onesv, accb[o]);
      }
#pragma unroll
      for (int kr = 0; kr < 3; ++kr)
#pragma unroll
        for (int kc = 0; kc < 3; ++kc)
#pragma unroll
          for (int o = 0; o < AO; ++o)
#pragma unroll
            for (int i = 0; i < AI; ++i) acc[kr][kc][o][i] = mfma16(av[kr][o], bv[kc][i], acc[kr][kc][o][i]);
    }
    fi_lds_barrier();
  };
  // (no conditional step inside the loop: the merge of the two register stages behind it would be copies of registers whose loads are
  // still in flight -- an early s_waitcnt vmcnt)
  int rho = r0;
  for (; rho + 1 < r1; rho += 2) {
    step(rho, xrA, drA);
    step(rho + 1, xrB, drB);
  }
  if (rho < r1) step(rho, xrA, drA);

  if constexpr (PAIR) {
    // this wave's pair straight to the item's slice: slice[(co * 9 + t) * cin + ci], D[row = co = g * 4 + r][col = ci = li]
    float* const slice = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int kr = 0; kr < 3; ++kr)
#pragma unroll
      for (int kc = 0; kc < 3; ++kc)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          slice[((size_t)(po * 16 + g * 4 + r) * 9 + kr * 3 + kc) * (NCI * 16) + pi * 16 + li] = acc[kr][kc][0][0][r];
    if (a.want_bias && pi == 0 && li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) slice[(size_t)NCO * 16 * 9 * NCI * 16 + po * 16 + g * 4 + r] = accb[0][r];
    }
    return;
  }
  // ---- fold the four waves' sums through LDS (the ring is dead), fixed order: deterministic
  constexpr int NACC = 9 * NCO * NCI * 256;
  float* const red = reinterpret_cast<float*>(smem);       // [tap][o][i][co 16][ci 16], then [NCO * 16] for the bias
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int kr = 0; kr < 3; ++kr)
#pragma unroll
        for (int kc = 0; kc < 3; ++kc)
#pragma unroll
          for (int o = 0; o < AO; ++o)
#pragma unroll
            for (int i = 0; i < AI; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                // D[row = co = g * 4 + r][col = ci = li]
                float* dst = &red[((((kr * 3 + kc) * NCO + o) * NCI + i) * 16 + g * 4 + r) * 16 + li];
                *dst = (w == 0) ? acc[kr][kc][o][i][r] : *dst + acc[kr][kc][o][i][r];
              }
      if (li == 0) {
#pragma unroll
        for (int o = 0; o < AO; ++o)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* dst = &red[NACC + o * 16 + g * 4 + r];
            *dst = (w == 0) ? accb[o][r] : *dst + accb[o][r];
          }
      }
    }
  }
  __syncthreads();
  // this item's slice: slice[(co * 9 + t) * cin + ci], bias behind it (conv_wgrad_kernel's layout)
  const int CIN = NARROW == 1 ? a.c0 : NCI * 16, COUT = NARROW == 2 ? a.cd : NCO * 16;     // (constants unless a side is narrow)
  float* const slice = a.part + (size_t)blockIdx.x * a.part_stride;
  for (int e = tid; e < 9 * COUT * CIN; e += 256) {
    const int ci = e % CIN, t = (e / CIN) % 9, co = e / (CIN * 9);
    slice[e] = red[((((t * NCO) + (co >> 4)) * NCI + (ci >> 4)) * 16 + (co & 15)) * 16 + (ci & 15)];
  }
  if (a.want_bias && tid < COUT) slice[(size_t)COUT * 9 * CIN + tid] = red[NACC + tid];
}

template <typename T, int NCI, int NCO, int NARROW = 0>
static int launch_conv_wgrad_rows(const WgRowsArgs& a, int items, hipStream_t st) {
  size_t lds = (size_t)2 * NCI * (a.ws + 2) * 32 + (size_t)4 * NCO * a.ws * 32;
  const size_t red = (NCI == 2 && NCO == 2) ? 0 : (size_t)(9 * NCO * NCI * 256 + NCO * 16) * sizeof(float);
  if (lds < red) lds = red;
  if (NCI == 2 && NCO == 2 && a.ws > 128) return FI_ERR_UNSUPPORTED;
  // more than 64 KB (NCI = 2 at 256-wide strips: 65 792 B) must be allowed per kernel, like every other big-LDS launcher
  static const bool allowed = fi_allow_big_lds(reinterpret_cast<const void*>(&conv_wgrad_rows_kernel<T, NCI, NCO, NARROW>));
  if (lds > 160 * 1024 || (lds > 64 * 1024 && !allowed)) return FI_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((conv_wgrad_rows_kernel<T, NCI, NCO, NARROW>), dim3((unsigned)items), dim3(256), lds, st, a);
  FI_CHECK_LAUNCH();
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------------
// The same for the 3x3x3 filter gradient of the 3D layers whose slices are >= 32 wide (unet_3D at 128^3: 16 -> 16, 48 -> 16; at
// 64^3: 16 -> 32, 32 -> 32, 96 -> 32; at 32^3: 32 / 64 / 192 -> 64; /root/reference/code/networks/utils.py:99-123), layout of the
// one-launch form: dw [cout][9][3][c0 + c1], the depth taps as channel groups.  An item is a run of x rows of ONE slice sigma; x
// row (sigma, rho) meets the NINE gradient rows (sigma - kd + 1, rho - kr + 1): per K step 3 x operands and 9 dy operands feed 27
// MFMAs per pair of 16-channel blocks (one gradient block x one input block: 27 accumulators of 4 registers, all a wave can hold).
// The dy ring holds 4 rows of each of the three slices (every dy row is loaded by the items of three slices: they sit on one XCD).
// A workgroup owns a tile of NCO gradient x NCI input blocks of its item (a.nct x a.nit tiles per item, next to each other in the
// launch order); its P = NCO * NCI pairs go to the waves: P = 1: the four waves split the K steps, P = 2: two waves a pair, P = 3, 4:
// a wave a pair (no cross-wave sum: the accumulators go straight to the item's partial slice).
template <typename T, int NCI, int NCO, int MAXW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_rows3d_kernel(WgRowsArgs a) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int P = NCI * NCO;
  static_assert(P >= 1 && P <= 4, "pairs of 16-channel blocks per workgroup");
  constexpr int KSPLIT = P == 1 ? 4 : (P == 2 ? 2 : 1);          // waves sharing a pair (they split the K steps of a row)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W, ws = a.ws, D = a.depth;
  const int xplane = (ws + 2) * 32, dplane = ws * 32;
  const int xrow = NCI * xplane, drow = NCO * dplane;
  char* const xs = smem;                        // [2][NCI][ws + 2][16]
  char* const ds = smem + 2 * xrow;             // [3 kd][4][NCO][ws][16]
  // workgroup -> (item, tile): an XCD (blockIdx % 8) takes a contiguous run of the logical order, so the dy rows the items of the slices
  // sigma - 1, sigma, sigma + 1 (and the tiles of an item) all load come out of that XCD's L2 once
  const int ntiles = a.nct * a.nit;
  const int total = a.N * a.strips * a.chunks * ntiles, per_xcd = (total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= total) return;
  const int tile = logical % ntiles;
  int item = logical / ntiles;
  const int slice_idx = item;
  const int cit = tile % a.nit, cot = tile / a.nit;
  const int cib = cit * NCI * 16, cob = cot * NCO * 16;
  const int cin = a.c0 + a.c1, cout = a.cout;
  const int chunk = item % a.chunks;
  item /= a.chunks;
  const int strip = item % a.strips, n = item / a.strips;        // n: slice index over all volumes
  const int sigma = n % D;
  const int r0 = chunk * a.rpw, r1 = min(r0 + a.rpw, H);
  const int cs = strip * ws;
  const size_t plane = (size_t)H * W;
  const T* const x0 = reinterpret_cast<const T*>(a.x0) + (size_t)n * plane * a.c0;
  const T* const x1 = reinterpret_cast<const T*>(a.x1) + (size_t)n * plane * a.c1;
  const T* const dyv = reinterpret_cast<const T*>(a.dy) + cob;

  constexpr int NXV = ((MAXW + 2) * 2 * NCI + 255) / 256, NDV = (MAXW * 2 * NCO + 255) / 256;
  const int nxv = (ws + 2) * 2 * NCI, ndv = ws * 2 * NCO;
  uint4 xrA[NXV], xrB[NXV], drA[3][NDV], drB[3][NDV];
  // Everything a row's loads and LDS stores need per thread is fixed for the whole run and computed once: the source pointer of each
  // vector at row 0 (column and channel folded in, out-of-range columns clamped to a valid address), its row pitch, its place in the
  // LDS plane, and a bit per vector for "inside the strip".  A row step then costs one 64-bit multiply-add per load and a scalar row
  // test -- recomputing the addresses per row was 0.9 + 0.8 of a step's 6.6 k cycles at 48 -> 16 and half of a step at 16 -> 16
  // (tools/rows3d_trace.py), issue slots taken from the MFMAs of the workgroup sharing the SIMDs.
  const char* xsrc[NXV];
  unsigned xpitch[NXV], xdst[NXV], xin = 0, xcol = 0;
#pragma unroll
  for (int it = 0; it < NXV; ++it) {
    const int i = tid + it * 256;
    const int v = i % (2 * NCI), px = i / (2 * NCI);
    const int gx = cs + px - 1, ch = cib + v * 8;
    const bool in = i < nxv, colok = in && gx >= 0 && gx < W;
    const bool first = ch < a.c0;
    xsrc[it] = reinterpret_cast<const char*>(first ? x0 + (size_t)(colok ? gx : 0) * a.c0 + ch : x1 + (size_t)(colok ? gx : 0) * a.c1 + (ch - a.c0));
    xpitch[it] = (unsigned)((first ? a.c0 : a.c1) * W * (int)sizeof(T));
    xdst[it] = (unsigned)((v >> 1) * xplane + px * 32 + (v & 1) * 16);
    xin |= (unsigned)in << it;
    xcol |= (unsigned)colok << it;
  }
  unsigned dsrc[NDV], ddst[NDV], din = 0;
#pragma unroll
  for (int it = 0; it < NDV; ++it) {
    const int i = tid + it * 256;
    const int v = i % (2 * NCO), px = i / (2 * NCO);
    const bool in = i < ndv;
    dsrc[it] = (unsigned)(((cs + (in ? px : 0)) * cout + v * 8) * (int)sizeof(T));
    ddst[it] = (unsigned)((v >> 1) * dplane + px * 32 + (v & 1) * 16);
    din |= (unsigned)in << it;
  }
  auto load_x = [&](uint4 (&xr)[NXV], int rho) {
    const unsigned rc = (unsigned)min(max(rho, 0), H - 1);   // out-of-range rows read a valid one; store_x zeroes them
#pragma unroll
    for (int it = 0; it < NXV; ++it) xr[it] = *reinterpret_cast<const uint4*>(xsrc[it] + (size_t)rc * xpitch[it]);
  };
  // The zeroing of out-of-range lanes happens when a row is STORED to LDS, two steps after its loads were issued: a select right
  // at the load is a use of the loaded registers that the compiler places (with its s_waitcnt vmcnt) at the end of the same step.
  auto load_d = [&](uint4 (&dr)[3][NDV], int r) {           // dy rows r of the slices sigma + 1, sigma, sigma - 1 (kd = 0, 1, 2)
    const int rc = min(max(r, 0), H - 1);
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int sd = sigma - kd + 1;
      const bool sok = sd >= 0 && sd < D;
      const char* const base = reinterpret_cast<const char*>(dyv + ((size_t)(sok ? n - kd + 1 : n) * plane + (size_t)rc * W) * cout);   // uniform
#pragma unroll
      for (int it = 0; it < NDV; ++it) dr[kd][it] = *reinterpret_cast<const uint4*>(base + dsrc[it]);
    }
  };
  auto store_x = [&](const uint4 (&xr)[NXV], int rho) {      // the registers hold x row rho
    char* const dst = xs + (rho & 1) * xrow;
    const bool rowok = rho >= 0 && rho < H;
#pragma unroll
    for (int it = 0; it < NXV; ++it)
      if ((xin >> it) & 1u) *reinterpret_cast<uint4*>(dst + xdst[it]) = fi_vec_select(rowok && ((xcol >> it) & 1u), xr[it]);
  };
  auto store_d = [&](const uint4 (&dr)[3][NDV], int r) {      // the registers hold dy rows r of the three slices
    const bool rowok = r >= 0 && r < H;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const int sd = sigma - kd + 1;
      const bool ok = rowok && sd >= 0 && sd < D;
      char* const dst = ds + (kd * 4 + (r & 3)) * drow;
#pragma unroll
      for (int it = 0; it < NDV; ++it)
        if ((din >> it) & 1u) *reinterpret_cast<uint4*>(dst + ddst[it]) = fi_vec_select(ok, dr[kd][it]);
    }
  };

  f32x4 acc[3][3][3];                            // [kd][kr][kc]
  f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kr = 0; kr < 3; ++kr)
#pragma unroll
      for (int kc = 0; kc < 3; ++kc) acc[kd][kr][kc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const frag_t onesv = WgFrag<T>::ones();
  const int g = lane >> 4, li = lane & 15;
  const int laneoff = (4 * g + (li >> 2)) * 32 + (li & 3) * 8;
  const int pair = KSPLIT == 4 ? 0 : (KSPLIT == 2 ? (wave & 1) : wave);      // this wave's (gradient block, input block)
  const bool active = pair < P;
  const int ob = active ? pair / NCI : 0, ib = active ? pair % NCI : 0;
  const int ks0 = KSPLIT == 4 ? wave : (KSPLIT == 2 ? (wave >> 1) : 0), ksstep = KSPLIT;
  // the bias sum rides along as a 28th MFMA against ones (zeros where it is not wanted: no branch inside the MFMA block)
  union {
    frag_t v;
    uint4 u;
  } bias_u;
  bias_u.v = onesv;
  const bool bias_wave = a.want_bias && cit == 0 && ib == 0 && active;
  if (!bias_wave) bias_u.u = make_uint4(0u, 0u, 0u, 0u);
  const frag_t bias_b = bias_u.v;

  load_x(xrA, r0);
  load_d(drA, r0 - 1);
  load_d(drB, r0);
  store_x(xrA, r0);
  store_d(drA, r0 - 1);
  load_d(drA, r0 + 1);
  store_d(drB, r0);
  store_d(drA, r0 + 1);
  load_x(xrA, r0 + 1);
  load_d(drA, r0 + 2);
  load_x(xrB, r0 + 2);
  load_d(drB, r0 + 3);
  __syncthreads();

  const int nks = ws / 32;
  auto step = [&](int rho, uint4 (&xr)[NXV], uint4 (&dr)[3][NDV]) __attribute__((always_inline)) {
    FI_TROW(0);
    store_x(xr, rho + 1);
    store_d(dr, rho + 2);
    FI_TROW(1);
    load_x(xr, rho + 3);
    load_d(dr, rho + 4);
    FI_TROW(2);
    if (active) {
      // A K step is three groups of 9 MFMAs (one depth tap each: 3 gradient fragments x the 3 column-shifted input fragments).  The
      // reads of group j + 1 are issued as one block BEFORE the MFMAs of group j, the scheduler fenced between the blocks: left to
      // itself the compiler reads each gradient fragment right before its three MFMAs -- nine exposed LDS latencies per step.
      // Two sets of 3 + 3 fragments: the same 48 registers (a whole step ahead would be 96 and cost the second workgroup of a CU).
      const char* const xb = xs + (rho & 1) * xrow + ib * xplane + laneoff;
      const char* dbk[3];
#pragma unroll
      for (int kr = 0; kr < 3; ++kr) dbk[kr] = ds + ((rho - kr + 1) & 3) * drow + ob * dplane + laneoff;
      auto rd_b = [&](frag_t (&bv)[3], int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int kc = 0; kc < 3; ++kc) bv[kc] = wgr_frag<T>(xb + (ks * 32 + kc) * 32);
      };
      auto rd_a = [&](frag_t (&av)[3], int ks, int kd) __attribute__((always_inline)) {
#pragma unroll
        for (int kr = 0; kr < 3; ++kr) av[kr] = wgr_frag<T>(dbk[kr] + kd * 4 * drow + ks * 32 * 32);
      };
      auto mm = [&](const frag_t (&av)[3], const frag_t (&bv)[3], auto kdc) __attribute__((always_inline)) {
        constexpr int kd = decltype(kdc)::value;
        if constexpr (kd == 1) accb = mfma16(av[1], bias_b, accb);
#pragma unroll
        for (int kr = 0; kr < 3; ++kr)
#pragma unroll
          for (int kc = 0; kc < 3; ++kc) acc[kd][kr][kc] = mfma16(av[kr], bv[kc], acc[kd][kr][kc]);
      };
      // one K step: its first group's fragments are in a0 / b0; leaves the next step's first group in a1 / b1
      auto kstep = [&](frag_t (&a0)[3], frag_t (&a1)[3], frag_t (&b0)[3], frag_t (&b1)[3], int ks) __attribute__((always_inline)) {
        rd_a(a1, ks, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(a0, b0, std::integral_constant<int, 0>());
        __builtin_amdgcn_sched_barrier(0);
        rd_a(a0, ks, 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(a1, b0, std::integral_constant<int, 1>());
        __builtin_amdgcn_sched_barrier(0);
        if (ks + ksstep < nks) {
          rd_b(b1, ks + ksstep);
          rd_a(a1, ks + ksstep, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mm(a0, b0, std::integral_constant<int, 2>());
        __builtin_amdgcn_sched_barrier(0);
      };
      frag_t avA[3], avB[3], bvA[3], bvB[3];
      int ks = ks0;
      if (ks < nks) {
        rd_b(bvA, ks);
        rd_a(avA, ks, 0);
      }
      for (; ks < nks; ks += 2 * ksstep) {
        kstep(avA, avB, bvA, bvB, ks);
        if (ks + ksstep < nks) kstep(avB, avA, bvB, bvA, ks + ksstep);
      }
    }
    FI_TROW(3);
    fi_lds_barrier();
    FI_TROW(4);
  };
  // (no conditional step inside the loop: a merge of the two register stages after it would be copies of registers whose loads are
  // still in flight -- another early s_waitcnt vmcnt)
  int rho = r0;
  for (; rho + 1 < r1; rho += 2) {
    step(rho, xrA, drA);
    step(rho + 1, xrB, drB);
  }
  if (rho < r1) step(rho, xrA, drA);

  // slice[((co * 9 + t) * 3 + kd) * cin + ci], bias behind it (the one-launch form's layout); D[row = co = g * 4 + r][col = ci = li]
  float* const slice = a.part + (size_t)slice_idx * a.part_stride;
  if constexpr (KSPLIT > 1) {
    // ---- red[pair][kd * 9 + t][co 16][ci 16]: the waves of a pair take turns adding (fixed order: deterministic); the ring is dead
    constexpr int NACC = 27 * 256;
    float* const red = reinterpret_cast<float*>(smem) + (size_t)pair * (NACC + 16);
    for (int w = 0; w < KSPLIT; ++w) {
      __syncthreads();
      if (ks0 == w) {
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
          for (int kr = 0; kr < 3; ++kr)
#pragma unroll
            for (int kc = 0; kc < 3; ++kc)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float* dst = &red[((kd * 9 + kr * 3 + kc) * 16 + g * 4 + r) * 16 + li];
                *dst = w == 0 ? acc[kd][kr][kc][r] : *dst + acc[kd][kr][kc][r];
              }
        if (li == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* dst = &red[NACC + g * 4 + r];
            *dst = w == 0 ? accb[r] : *dst + accb[r];
          }
        }
      }
    }
    __syncthreads();
    const float* const redall = reinterpret_cast<const float*>(smem);
    for (int e = tid; e < P * 16 * 27 * 16; e += 256) {
      const int c = e & 15, kd = (e >> 4) % 3, t = (e / 48) % 9, co = (e / 432) & 15, pr = e / 6912;
      const int o = pr / NCI, i = pr % NCI;
      slice[(((size_t)(cob + o * 16 + co) * 9 + t) * 3 + kd) * cin + cib + i * 16 + c] = redall[(size_t)pr * (NACC + 16) + ((kd * 9 + t) * 16 + co) * 16 + c];
    }
    if (a.want_bias && cit == 0 && tid < NCO * 16)
      slice[(size_t)cout * 27 * cin + cob + tid] = redall[(size_t)((tid >> 4) * NCI) * (NACC + 16) + NACC + (tid & 15)];
  } else {
    // pair-owning waves: no cross-wave sum, so the accumulators go straight to the slice -- a reduction buffer of 27 x P x 1 KB
    // (83 KB at 48 -> 16) would be the workgroup's LDS footprint and leave ONE workgroup per CU where the row ring allows two
    if (active) {
#pragma unroll
      for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int kr = 0; kr < 3; ++kr)
#pragma unroll
          for (int kc = 0; kc < 3; ++kc)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              slice[(((size_t)(cob + ob * 16 + g * 4 + r) * 9 + kr * 3 + kc) * 3 + kd) * cin + cib + ib * 16 + li] = acc[kd][kr][kc][r];
      if (bias_wave && li == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) slice[(size_t)cout * 27 * cin + cob + ob * 16 + g * 4 + r] = accb[r];
      }
    }
  }
}

template <typename T, int NCI, int NCO, int MAXW>
static int launch_conv_wgrad_rows3d(const WgRowsArgs& a, int items, hipStream_t st) {
  constexpr int P = NCI * NCO;
  if (a.ws > MAXW) return FI_ERR_UNSUPPORTED;
  size_t lds = (size_t)2 * NCI * (a.ws + 2) * 32 + (size_t)12 * NCO * a.ws * 32;
  const size_t red = P > 2 ? 0 : (size_t)P * (27 * 256 + 16) * sizeof(float);
  if (lds < red) lds = red;
  static const bool allowed = fi_allow_big_lds(reinterpret_cast<const void*>(&conv_wgrad_rows3d_kernel<T, NCI, NCO, MAXW>));
  if (lds > 160 * 1024 || (lds > 64 * 1024 && !allowed)) return FI_ERR_UNSUPPORTED;
  const long total = (long)items * a.nct * a.nit;
  hipLaunchKernelGGL((conv_wgrad_rows3d_kernel<T, NCI, NCO, MAXW>), dim3((unsigned)(((total + 7) / 8) * 8)), dim3(256), lds, st, a);
  FI_CHECK_LAUNCH();
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------------
// The same scheme for the CHANNEL-RICH 3x3 layers (32 ... 256 channels a side on 64^2 ... 256^2 maps: the decoder and the deep
// encoder levels).  The 16 x 16-quadrant tile kernel (conv_wgrad_quad_kernel) spends 3.5-7 us per 256-pixel tile of which its 72
// MFMAs are 0.5: a fresh 1 KB fragment per MFMA and wave is 256 B/clk of LDS reads against the 128 the LDS delivers, every tile
// costs two barriers and a global round trip, and a 32 x 32 channel tile re-reads each tensor Cout / 32 (Cin / 32) times.  Here a
// workgroup owns a 64 x 64 (or 32 x 64 / 64 x 32) channel tile of an item (a run of x rows of one image strip, as above): per
// 32-pixel K step a wave reads 3 x OB gradient and 3 x IB input operands for 9 x OB x IB MFMAs -- (2, 2) blocks per wave: a third of
// a fragment per MFMA, 85 B/clk for the four waves at the matrix pipe's full rate -- rows arrive as contiguous segments through
// the same two register stages and LDS ring with ONE barrier per row, and every wave owns its channel blocks for the whole run: no
// cross-wave sum, accumulators (9 x OB x IB x 4 registers = 144 for the 64 x 64 tile) go straight to the item's partial slice in
// conv_wgrad_kernel's layout.  An item's channel tiles sit next to each other in the launch order of ONE XCD, so the rows they
// share come out of that XCD's L2.
template <typename T, int TCO, int TCI, int MAXW>
__global__ __launch_bounds__(256) void conv_wgrad_rows64_kernel(WgRowsArgs a) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int WO = (TCO == 4 && TCI == 4) ? 2 : (TCO == 4 ? 4 : 1), WI = 4 / WO;     // wave grid over the channel tile
  constexpr int OB = TCO / WO, IB = TCI / WI;                                          // 16-channel blocks per wave
  static_assert(OB >= 1 && IB >= 1 && OB * WO == TCO && IB * WI == TCI, "tile / wave grid");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wo = wave / WI, wi = wave % WI;
  const int H = a.H, W = a.W, ws = a.ws;
  const int xplane = (ws + 2) * 32, dplane = ws * 32;
  const int xrow = TCI * xplane, drow = TCO * dplane;
  char* const xs = smem;                        // [2][TCI][ws + 2][16]
  char* const ds = smem + 2 * xrow;             // [4][TCO][ws][16]
  // workgroup -> (item, gradient tile, input tile): an XCD (blockIdx % 8) takes a contiguous run of the logical order
  const int ntiles = a.nct * a.nit;
  const int total = a.N * a.strips * a.chunks * ntiles, per_xcd = (total + 7) >> 3;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= total) return;
  const int tile = logical % ntiles;
  int item = logical / ntiles;
  const int slice_idx = item;
  const int cit = tile % a.nit, cot = tile / a.nit;
  const int chunk = item % a.chunks;
  item /= a.chunks;
  const int strip = item % a.strips, n = item / a.strips;
  const int r0 = chunk * a.rpw, r1 = min(r0 + a.rpw, H);
  const int cs = strip * ws;
  const int cin = a.c0 + a.c1, cout = a.cout;
  const int cib = cit * TCI * 16, cob = cot * TCO * 16;
  const T* const x0 = reinterpret_cast<const T*>(a.x0) + (size_t)n * H * W * a.c0;
  const T* const x1 = reinterpret_cast<const T*>(a.x1) + (size_t)n * H * W * a.c1;
  const T* const dyg = reinterpret_cast<const T*>(a.dy) + (size_t)n * H * W * cout + cob;

  constexpr int NXV = ((MAXW + 2) * 2 * TCI + 255) / 256, NDV = (MAXW * 2 * TCO + 255) / 256;
  const int nxv = (ws + 2) * 2 * TCI, ndv = ws * 2 * TCO;
  uint4 xrA[NXV], drA[NDV], xrB[NXV], drB[NDV];
  // per thread and vector, once for the run: source at row 0, row pitch, LDS offset, in-strip / in-image bits (conv_wgrad_rows_kernel)
  const char* xsrc[NXV];
  unsigned xpitch[NXV], xdst[NXV], dsrc[NDV], ddst[NDV], xin = 0, xcol = 0, din = 0;
#pragma unroll
  for (int it = 0; it < NXV; ++it) {
    const int i = tid + it * 256;
    const int v = i % (2 * TCI), px = i / (2 * TCI);
    const int gx = cs + px - 1, ch = cib + v * 8;
    const bool in = i < nxv, colok = in && gx >= 0 && gx < W;
    const bool first = ch < a.c0;
    xsrc[it] = reinterpret_cast<const char*>(first ? x0 + (size_t)(colok ? gx : 0) * a.c0 + ch : x1 + (size_t)(colok ? gx : 0) * a.c1 + (ch - a.c0));
    xpitch[it] = (unsigned)((first ? a.c0 : a.c1) * W * (int)sizeof(T));
    xdst[it] = (unsigned)((v >> 1) * xplane + px * 32 + (v & 1) * 16);
    xin |= (unsigned)in << it;
    xcol |= (unsigned)colok << it;
  }
#pragma unroll
  for (int it = 0; it < NDV; ++it) {
    const int i = tid + it * 256;
    const int v = i % (2 * TCO), px = i / (2 * TCO);
    const bool in = i < ndv;
    dsrc[it] = (unsigned)(((cs + (in ? px : 0)) * cout + v * 8) * (int)sizeof(T));
    ddst[it] = (unsigned)((v >> 1) * dplane + px * 32 + (v & 1) * 16);
    din |= (unsigned)in << it;
  }
  auto load_x = [&](uint4 (&xr)[NXV], int rho) {
    const unsigned rc = (unsigned)min(max(rho, 0), H - 1);
#pragma unroll
    for (int it = 0; it < NXV; ++it) xr[it] = *reinterpret_cast<const uint4*>(xsrc[it] + (size_t)rc * xpitch[it]);
  };
  auto load_d = [&](uint4 (&dr)[NDV], int r) {
    const char* const base = reinterpret_cast<const char*>(dyg + (size_t)min(max(r, 0), H - 1) * W * cout);     // uniform
#pragma unroll
    for (int it = 0; it < NDV; ++it) dr[it] = *reinterpret_cast<const uint4*>(base + dsrc[it]);
  };
  auto store_x = [&](const uint4 (&xr)[NXV], int rho) {           // the registers hold x row rho; lanes outside the image are zeroed here
    char* const dst = xs + (rho & 1) * xrow;
    const bool rowok = rho >= 0 && rho < H;
#pragma unroll
    for (int it = 0; it < NXV; ++it)
      if ((xin >> it) & 1u) *reinterpret_cast<uint4*>(dst + xdst[it]) = fi_vec_select(rowok && ((xcol >> it) & 1u), xr[it]);
  };
  auto store_d = [&](const uint4 (&dr)[NDV], int r) {
    char* const dst = ds + (r & 3) * drow;
    const bool rowok = r >= 0 && r < H;
#pragma unroll
    for (int it = 0; it < NDV; ++it)
      if ((din >> it) & 1u) *reinterpret_cast<uint4*>(dst + ddst[it]) = fi_vec_select(rowok, dr[it]);
  };

  f32x4 acc[3][3][OB][IB];
  f32x4 accb[OB];
#pragma unroll
  for (int kr = 0; kr < 3; ++kr)
#pragma unroll
    for (int kc = 0; kc < 3; ++kc)
#pragma unroll
      for (int o = 0; o < OB; ++o)
#pragma unroll
        for (int i = 0; i < IB; ++i) acc[kr][kc][o][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < OB; ++o) accb[o] = f32x4{0.f, 0.f, 0.f, 0.f};
  const frag_t onesv = WgFrag<T>::ones();
  const bool bias_wave = a.want_bias && cit == 0 && wi == 0;

  const int g = lane >> 4, li = lane & 15;
  const int laneoff = (4 * g + (li >> 2)) * 32 + (li & 3) * 8;

  load_x(xrA, r0);
  load_d(drA, r0 - 1);
  load_d(drB, r0);
  store_x(xrA, r0);
  store_d(drA, r0 - 1);
  load_d(drA, r0 + 1);
  store_d(drB, r0);
  store_d(drA, r0 + 1);
  load_x(xrA, r0 + 1);
  load_d(drA, r0 + 2);
  load_x(xrB, r0 + 2);
  load_d(drB, r0 + 3);
  __syncthreads();

  const int nks = ws / 32;
  auto step = [&](int rho, uint4 (&xr)[NXV], uint4 (&dr)[NDV]) __attribute__((always_inline)) {
    store_x(xr, rho + 1);
    store_d(dr, rho + 2);
    load_x(xr, rho + 3);
    load_d(dr, rho + 4);
    const char* const xb = xs + (rho & 1) * xrow + wi * IB * xplane + laneoff;
    const char* db[3];
#pragma unroll
    for (int kr = 0; kr < 3; ++kr) db[kr] = ds + ((rho - kr + 1) & 3) * drow + wo * OB * dplane + laneoff;
    for (int ks = 0; ks < nks; ++ks) {
      frag_t av[3][OB], bv[3][IB];
#pragma unroll
      for (int kr = 0; kr < 3; ++kr)
#pragma unroll
        for (int o = 0; o < OB; ++o) av[kr][o] = wgr_frag<T>(db[kr] + o * dplane + ks * 32 * 32);
#pragma unroll
      for (int kc = 0; kc < 3; ++kc)
#pragma unroll
        for (int i = 0; i < IB; ++i) bv[kc][i] = wgr_frag<T>(xb + i * xplane + (ks * 32 + kc) * 32);
      if (bias_wave) {
#pragma unroll
        for (int o = 0; o < OB; ++o) accb[o] = mfma16(av[1][o], onesv, accb[o]);
      }
#pragma unroll
      for (int kr = 0; kr < 3; ++kr)
#pragma unroll
        for (int kc = 0; kc < 3; ++kc)
#pragma unroll
          for (int o = 0; o < OB; ++o)
#pragma unroll
            for (int i = 0; i < IB; ++i) acc[kr][kc][o][i] = mfma16(av[kr][o], bv[kc][i], acc[kr][kc][o][i]);
    }
    fi_lds_barrier();
  };
  int rho = r0;                                 // (no conditional step inside the loop: conv_wgrad_rows_kernel)
  for (; rho + 1 < r1; rho += 2) {
    step(rho, xrA, drA);
    step(rho + 1, xrB, drB);
  }
  if (rho < r1) step(rho, xrA, drA);

  // ---- this wave's blocks of the item's slice: slice[(co * 9 + t) * cin + ci], D[row = co = g * 4 + r][col = ci = li]
  float* const slice = a.part + (size_t)slice_idx * a.part_stride;
#pragma unroll
  for (int kr = 0; kr < 3; ++kr)
#pragma unroll
    for (int kc = 0; kc < 3; ++kc)
#pragma unroll
      for (int o = 0; o < OB; ++o)
#pragma unroll
        for (int i = 0; i < IB; ++i) {
          const int gci = cib + (wi * IB + i) * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int gco = cob + (wo * OB + o) * 16 + g * 4 + r;
            slice[((size_t)gco * 9 + kr * 3 + kc) * cin + gci] = acc[kr][kc][o][i][r];
          }
        }
  if (bias_wave && li == 0) {
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) slice[(size_t)cout * 9 * cin + cob + (wo * OB + o) * 16 + g * 4 + r] = accb[o][r];
  }
}

template <typename T, int TCO, int TCI>
static int launch_conv_wgrad_rows64(const WgRowsArgs& a, int items, hipStream_t st) {
  const size_t lds = (size_t)2 * TCI * (a.ws + 2) * 32 + (size_t)4 * TCO * a.ws * 32;
  const void* kern = reinterpret_cast<const void*>(&conv_wgrad_rows64_kernel<T, TCO, TCI, 128>);
  static bool allowed = fi_allow_big_lds(kern);
  if (!allowed || lds > 160 * 1024 || a.ws > 128) return FI_ERR_UNSUPPORTED;
  const long total = (long)items * a.nct * a.nit;
  hipLaunchKernelGGL((conv_wgrad_rows64_kernel<T, TCO, TCI, 128>), dim3((unsigned)(((total + 7) / 8) * 8)), dim3(256), lds, st, a);
  FI_CHECK_LAUNCH();
  return 0;
}
