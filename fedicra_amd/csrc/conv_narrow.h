// Forward convolution of the layers with a NARROW input side (gfx950 only): the network's first convolution (in_chns = 1 or 3
// -> 16, /root/reference/code/networks/unet.py:14-30 as instantiated at :82 / :163) and the input gradient of the logits
// convolution (n_class <= 4 gradient channels -> 16, unet.py:228), 3x3, 16-bit storage, dense NHWC with C <= 4.
//
// The tile kernels pad such an input to 8-channel vectors and stage it element by element with clamped 64-bit addresses; 12 x
// 512^2 x 3 -> 16 took 60 us against an HBM floor of 15 (19 MB in, 100 MB out).  Here
//   * a workgroup owns a 16-row x 64-column tile; its 18 x 66 halo tile is staged once, a pixel = 4 channel slots of 16 bits
//     (8 bytes, slot 3 / the channels the input does not have = 0); rows are 80 pixels apart in LDS, which puts the four k-groups
//     of a fragment read 32 banks apart;
//   * the contraction is laid out for the pixel-major tile: k-group g < 3 of a lane is filter ROW g, its 8 elements in the first
//     MFMA the taps s = 0, 1 (2 pixels x 4 slots: one 16-byte run of the tile), in the second MFMA tap s = 2 (4 slots); k-group 3
//     is zero.  Two v_mfma_f32_16x16x32 per 16 pixels -- the layer is memory-bound, the matrix pipe idles either way;
//   * epilogue, BatchNorm statistics and statistics groups exactly as conv_thin_kernel (values as stored, fp64 slots).
// Each output element is the same fp32 sum of products in a different order than the tile kernels' (another K layout): equal to
// the rounding of the 16-bit store in all but rare ties.
#pragma once
#include "conv_impl.h"

template <typename T>
__global__ __launch_bounds__(256) void conv_narrow_in_kernel(ConvArgs a) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  typedef typename DT<T>::frag_t frag_t;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  constexpr int TH = 16, TW = 64, XH = TH + 2, XW = TW + 2, XWP = 80, MF = 4;
  __shared__ uint2 xs[XH * XWP];
  __shared__ float red[4 * 16 * 2];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kg = lane >> 4;
  const int cin = a.c0, cout = a.co0, H = a.H, W = a.W;
  int tile = blockIdx.x;
  const int tx = tile % a.tilesX;
  tile /= a.tilesX;
  const int ty = tile % a.tilesY, n = tile / a.tilesY;

  // ---- the filter as two "A" fragments: row = output channel li, k-group kg = filter row
  // (16-bit raw buffer loads: an element the lane does not have takes an out-of-range offset and the hardware returns zero.
  //  Plain loads under a predicate become branches -- hipcc sinks a load whose value is selected against a constant into the
  //  taken side -- and branches serialise the loads' round trips)
  constexpr unsigned OOB = 0xFFFFFFF0u;
  unsigned wa[4] = {0u, 0u, 0u, 0u}, wb[2] = {0u, 0u};
  {
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (unsigned)(cout * 9 * cin) * 2u, 0x00020000);
    const bool wok = kg < 3 && li < cout;
    const unsigned wo = (unsigned)((li * 9 + kg * 3) * cin) * 2u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int s = j >> 2, c = j & 3;
      const unsigned e = __builtin_amdgcn_raw_buffer_load_b16(rw, (wok && c < cin) ? wo + (unsigned)(s * cin + c) * 2u : OOB, 0, 0);
      wa[j >> 1] |= e << (16 * (j & 1));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned e = __builtin_amdgcn_raw_buffer_load_b16(rw, (wok && c < cin) ? wo + (unsigned)(2 * cin + c) * 2u : OOB, 0, 0);
      wb[c >> 1] |= e << (16 * (c & 1));
    }
  }
  const frag_t fa = __builtin_bit_cast(frag_t, (v4u){wa[0], wa[1], wa[2], wa[3]});
  const frag_t fb = __builtin_bit_cast(frag_t, (v4u){wb[0], wb[1], 0u, 0u});
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = kg * 4 + r;
    bv[r] = (a.bias && co < cout) ? a.bias[co] : 0.f;
  }

  // ---- halo tile: one pixel per thread and pass, its <= 4 elements packed into 8 bytes (zero outside the image); all loads of
  //      the tile are in flight at once
  {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x0), 0, (unsigned)a.N * (unsigned)H * (unsigned)W * (unsigned)cin * 2u, 0x00020000);
    constexpr int NP = (XH * XW + 255) / 256;
    unsigned e[NP][4];
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int p = tid + it * 256;
      const int row = p / XW, col = p - row * XW;
      const int gy = ty * TH + row - 1, gx = tx * TW + col - 1;
      const bool ok = p < XH * XW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const unsigned o = (unsigned)((n * H + gy) * W + gx) * (unsigned)cin * 2u;
#pragma unroll
      for (int c = 0; c < 4; ++c) e[it][c] = __builtin_amdgcn_raw_buffer_load_b16(rx, (ok && c < cin) ? o + 2u * c : OOB, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int p = tid + it * 256;
      const int row = p / XW, col = p - row * XW;
      if (p < XH * XW) xs[row * XWP + col] = make_uint2(e[it][0] | (e[it][1] << 16), e[it][2] | (e[it][3] << 16));
    }
  }
  __syncthreads();

  const unsigned esz = sizeof(T);
  const __amdgpu_buffer_rsrc_t ry =
      __builtin_amdgcn_make_buffer_rsrc(a.y0, 0, (unsigned)a.N * (unsigned)H * (unsigned)W * (unsigned)cout * esz, 0x00020000);
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  const int krow = kg < 3 ? kg : 0;

#pragma unroll
  for (int seg = 0; seg < TW / 16; ++seg) {
    f32x4 acc[MF];
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const uint2* px = &xs[(wave * MF + m + krow) * XWP + seg * 16 + li];
      const uint2 v0 = px[0], v1 = px[1], v2 = px[2];
      v4u b1 = {v0.x, v0.y, v1.x, v1.y}, b2 = {v2.x, v2.y, 0u, 0u};
      if (kg == 3) {                                               // (its filter fragment is zero; a NaN of the tile must not meet it)
        b1 = (v4u){0u, 0u, 0u, 0u};
        b2 = b1;
      }
      acc[m] = mfma16(fa, __builtin_bit_cast(frag_t, b1), f32x4{0.f, 0.f, 0.f, 0.f});
      acc[m] = mfma16(fb, __builtin_bit_cast(frag_t, b2), acc[m]);
    }
    // epilogue of conv_thin_kernel: two tile rows swap halves across the 16-lane rows, a lane stores 8 channels of one pixel
    const int gx = tx * TW + seg * 16 + li;
    const bool colok = gx < W;
#pragma unroll
    for (int mp = 0; mp < MF; mp += 2) {
      v2u q[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = mp + h;
        const float mk = (colok && ty * TH + wave * MF + m < H) ? 1.f : 0.f;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[m][r] + bv[r];
        q[h] = __builtin_bit_cast(v2u, Quad<T>::pack(v));          // v := the values as stored
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vm = v[r] * mk;                               // tile overhang does not count
          ssum[r] += vm;
          ssq[r] += vm * v[r];
        }
      }
      const v2u lo = __builtin_amdgcn_permlane16_swap(q[0].x, q[1].x, false, false);
      const v2u hi = __builtin_amdgcn_permlane16_swap(q[0].y, q[1].y, false, false);
      const v4u out = {lo.x, hi.x, lo.y, hi.y};                    // channels cg .. cg+7 of pixel row mp + (kg & 1)
      const int gy = ty * TH + wave * MF + mp + (kg & 1);
      const int cg = (kg >> 1) * 8;
      const unsigned o = ((unsigned)((n * H + gy) * W + gx) * (unsigned)cout + (unsigned)cg) * esz;
      __builtin_amdgcn_raw_buffer_store_b128(out, ry, (colok && gy < H && cg < cout) ? o : OOB, 0, 0);
    }
  }

  if (a.stats) {                                                    // uniform
    const int grp = a.gimages > 0 ? n / a.gimages : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = fi_row16_sum(ssum[r]), q = fi_row16_sum(ssq[r]);
      if (li == 0) {
        red[(wave * 16 + kg * 4 + r) * 2 + 0] = s;
        red[(wave * 16 + kg * 4 + r) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int c = tid >> 1, which = tid & 1;
      if (c < cout) {
        double tot = 0.0;
#pragma unroll
        for (int wv_ = 0; wv_ < 4; ++wv_) tot += (double)red[(wv_ * 16 + c) * 2 + which];
        const int slot = blockIdx.x & (FI_STATS_SLOTS - 1);
        atomicAdd(&a.stats[(size_t)grp * a.stats_gstride + ((size_t)slot * cout + c) * 2 + which], tot);
      }
    }
  }
}

// a.tilesX / a.tilesY: 64-column x 16-row tiles (set by the caller).  cout = 8 or 16, cin <= 4, one source, plain epilogue.
template <typename T>
static int launch_conv_narrow_in(const ConvArgs& a, hipStream_t st) {
  const long ntile = (long)a.N * a.tilesX * a.tilesY;
  hipLaunchKernelGGL((conv_narrow_in_kernel<T>), dim3((unsigned)ntile), dim3(256), 0, st, a);
  FI_CHECK_LAUNCH();
  return 0;
}
