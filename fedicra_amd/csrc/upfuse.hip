// UpBlock's first half as ONE kernel (gfx950): u = Upsample2x(conv1x1(z)), z = act(BN(y)) of the producing ConvBlock or a plain
// activation.  /root/reference/code/networks/unet.py:57-70 (self.conv1x1 -> self.up, bilinear, align_corners=True).
//
// The two-launch form writes the low-resolution convolution output and reads it back four vectors per output vector (the
// up-sampling kernel is bound by the vector L1: 64 B of loads per 16-byte store); here a workgroup convolves R + 2 input rows of
// one image on the matrix pipe into LDS (rounded to the storage type exactly as the separate launch stores them) and
// interpolates 2R output rows out of LDS: the low-resolution tensor never exists, an input row is interpolated along x ONCE
// per output column and reused by the ~4 output rows it reaches.
//
// Phase 1 (flat GEMM over the tile's nrows x w pixels; rows of one image are contiguous in NHWC): per 16 pixels
//   D[cout][pixel] = W[cout][cin] x Z^T[cin][pixel] with v_mfma_f32_16x16x32: A = filter rows (LDS, rows padded by 16 B so the
//   16 rows of a fragment fall on distinct banks; kept in registers when the whole filter is <= 16 fragments), B = a pixel's 8
//   consecutive channels straight from global memory (16 B per lane, 8 vectors per lane in flight, double buffered), the
//   producer's BatchNorm + LeakyReLU applied in registers (fi_bn_act_fwd's arithmetic and rounding).  A lane ends up with 4
//   consecutive output channels of one pixel: + bias, round, one 8-byte LDS write.
// Phase 2: a thread owns output columns (ox, 8-channel vector); x geometry once per column, the row geometry from a small LDS
//   table (wave-uniform); h(row) = lx0 * a + lx1 * b per input row, out = ly0 * h(y0) + ly1 * h(y1): the separate kernel's
//   expression, operand for operand (fi_lerp2 in both), so the two forms give the same bits whenever the convolution does.
#include "common.h"

namespace {

struct UpFuseArgs {
  const void* x;
  const void* wmat;            // [COUT][CIN] storage type (fi_pack_weights mode 0 of a 1x1 filter)
  const float* bias;           // [COUT] or nullptr
  void* y;                     // [N][2h][2w][COUT]
  const float* scale;          // fp32 [groups][CIN] or nullptr
  const float* shift;
  float slope;
  int N, h, w, gimages, R, tiles, total, chunk;
  float sh, sw;
};

__device__ __forceinline__ void lin_coord(int o, int in, float sc, int& i0, int& i1, float& l0, float& l1) {   // == ops.hip
  const float src = sc * (float)o;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + 1 < in ? i0 + 1 : in - 1;
  l1 = src - (float)i0;
  if (l1 < 0.f) l1 = 0.f;
  if (l1 > 1.f) l1 = 1.f;
  l0 = 1.f - l1;
}

template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(256) void conv1x1_up2x_kernel(const UpFuseArgs a) {
  typedef typename DT<T>::frag_t frag_t;
  constexpr int KS = CIN / 32, NB = COUT / 16, CV = COUT / 8;
  constexpr int WS = CIN * 2 + 16;                       // filter row stride in LDS (bytes)
  constexpr bool REGW = NB * KS <= 16;                   // whole filter as register fragments
  constexpr bool REGC = KS <= 2;                         // BatchNorm coefficients in registers
  constexpr int PF = KS >= 8 ? 1 : 8 / KS;               // pixel blocks per load group: 8 vectors per lane in flight
  static_assert(CIN % 32 == 0 && COUT % 16 == 0 && (CV & (CV - 1)) == 0, "channel counts");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- which tile: consecutive tiles of an image share two input rows, so an XCD (blockIdx % 8) takes a contiguous run
  const int logical = (int)(blockIdx.x & 7) * a.chunk + (int)(blockIdx.x >> 3);
  if (logical >= a.total) return;
  const int n = logical / a.tiles, tile = logical - n * a.tiles;
  const int h = a.h, w = a.w, Ho = 2 * h, Wo = 2 * w;
  const int oyb = tile * 2 * a.R;
  const int nout = min(2 * a.R, Ho - oyb);
  int ylo, yhi, t0_, t1_;
  float f0_, f1_;
  lin_coord(oyb, h, a.sh, ylo, t1_, f0_, f1_);
  lin_coord(oyb + nout - 1, h, a.sh, t0_, yhi, f0_, f1_);
  const int nrows = yhi - ylo + 1;                       // <= R + 2
  const int P = nrows * w;                               // pixels convolved by this workgroup

  char* const s_rows = smem;                                                     // [(R+2) * w][COUT] T
  char* const s_w = s_rows + (size_t)(a.R + 2) * w * COUT * 2;                   // [COUT] rows of WS bytes
  float* const s_sc = reinterpret_cast<float*>(s_w + COUT * WS);                 // [CIN]
  float* const s_sh = s_sc + CIN;                                                // [CIN]
  float* const s_b = s_sh + CIN;                                                 // [COUT]
  int4* const s_tab = reinterpret_cast<int4*>(s_b + COUT);                       // [2R] {y0 - ylo, y1 - ylo, ly0, ly1}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kg = lane >> 4;
  const bool xform = a.scale != nullptr;
  const T* const xin = reinterpret_cast<const T*>(a.x) + ((size_t)n * h + ylo) * w * CIN + kg * 8;
  const int nblk = (P + 15) >> 4;
  const int ngroups = (nblk + 4 * PF - 1) / (4 * PF);    // load groups per wave: group q of wave v = blocks (q*4 + v)*PF ..

  auto issue = [&](int q, uint4 (&v)[PF * KS]) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      int pix = ((q * 4 + wave) * PF + p) * 16 + li;
      pix = pix < P ? pix : 0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        v[p * KS + ks] = *reinterpret_cast<const uint4*>(xin + (size_t)pix * CIN + ks * 32);
    }
  };
  uint4 cur[PF * KS], nxt[PF * KS];
  issue(0, cur);                                         // in flight while the filter is staged

  // ---- filter, coefficients, bias, row table -> LDS
  {
    const T* const wsrc = reinterpret_cast<const T*>(a.wmat);
    constexpr int RV = CIN / 8;
    for (int v = tid; v < COUT * RV; v += 256) {
      const int r = v / RV, c = v - r * RV;
      *reinterpret_cast<uint4*>(s_w + r * WS + c * 16) = *reinterpret_cast<const uint4*>(wsrc + (size_t)r * CIN + c * 8);
    }
    if (xform) {
      const int g = a.gimages > 0 ? n / a.gimages : 0;
      for (int c = tid; c < CIN; c += 256) {
        s_sc[c] = a.scale[(size_t)g * CIN + c];
        s_sh[c] = a.shift[(size_t)g * CIN + c];
      }
    }
    for (int c = tid; c < COUT; c += 256) s_b[c] = a.bias ? a.bias[c] : 0.f;
    if (tid < nout) {
      int y0, y1;
      float l0, l1;
      lin_coord(oyb + tid, h, a.sh, y0, y1, l0, l1);
      s_tab[tid] = make_int4(y0 - ylo, y1 - ylo, __float_as_int(l0), __float_as_int(l1));
    }
  }
  __syncthreads();

  // ---- phase 1
  {
    frag_t wreg[REGW ? NB * KS : 1];
    if constexpr (REGW) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          wreg[nb * KS + ks] = *reinterpret_cast<const frag_t*>(s_w + (nb * 16 + li) * WS + (ks * 32 + kg * 8) * 2);
    }
    float csc[REGC ? KS * 8 : 1], csh[REGC ? KS * 8 : 1];
    if constexpr (REGC) {
      if (xform) {
#pragma unroll
        for (int i = 0; i < KS * 8; ++i) {
          csc[i] = s_sc[(i >> 3) * 32 + kg * 8 + (i & 7)];
          csh[i] = s_sh[(i >> 3) * 32 + kg * 8 + (i & 7)];
        }
      }
    }
    float bv[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[nb][j] = s_b[nb * 16 + kg * 4 + j];

    for (int q = 0; q < ngroups; ++q) {
      if (q + 1 < ngroups) issue(q + 1, nxt);
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int blk = (q * 4 + wave) * PF + p;
        if (blk < nblk) {                                // wave-uniform
          f32x4 acc[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            uint4 raw = cur[p * KS + ks];
            if (xform) {
              float f[8];
              VecWords<T>::unpack(raw, f);
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                float sc, sh;
                if constexpr (REGC) {
                  sc = csc[ks * 8 + j], sh = csh[ks * 8 + j];
                } else {
                  sc = s_sc[ks * 32 + kg * 8 + j], sh = s_sh[ks * 32 + kg * 8 + j];
                }
                const float t = f[j] * sc + sh;
                f[j] = fmaxf(t, t * a.slope);
              }
              raw = VecWords<T>::pack(f);
            }
            const frag_t bfrag = __builtin_bit_cast(frag_t, raw);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              frag_t af;
              if constexpr (REGW) {
                af = wreg[nb * KS + ks];
              } else {
                af = *reinterpret_cast<const frag_t*>(s_w + (nb * 16 + li) * WS + (ks * 32 + kg * 8) * 2);
              }
              acc[nb] = mfma16(af, bfrag, acc[nb]);
            }
          }
          const int pix = blk * 16 + li;
          if (pix < P) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              float v[4] = {acc[nb][0] + bv[nb][0], acc[nb][1] + bv[nb][1], acc[nb][2] + bv[nb][2], acc[nb][3] + bv[nb][3]};
              *reinterpret_cast<typename Quad<T>::q_t*>(s_rows + ((size_t)pix * COUT + nb * 16 + kg * 4) * 2) = Quad<T>::pack(v);
            }
          }
        }
      }
      if (q + 1 < ngroups) {
#pragma unroll
        for (int i = 0; i < PF * KS; ++i) cur[i] = nxt[i];
      }
    }
  }
  __syncthreads();

  // ---- phase 2
  {
    constexpr int CVS = CV == 1 ? 0 : CV == 2 ? 1 : CV == 4 ? 2 : CV == 8 ? 3 : CV == 16 ? 4 : 5;
    const int ncol = Wo * CV;
    const int rowb = w * COUT * 2;                       // bytes per convolved row in LDS
    T* const ybase = reinterpret_cast<T*>(a.y) + ((size_t)n * Ho + oyb) * Wo * COUT;
    for (int c = tid; c < ncol; c += 256) {
      const int ox = c >> CVS, cv = c & (CV - 1);
      int x0, x1;
      float lx0, lx1;
      lin_coord(ox, w, a.sw, x0, x1, lx0, lx1);
      const char* const p0 = s_rows + (x0 * COUT + cv * 8) * 2;
      const char* const p1 = s_rows + (x1 * COUT + cv * 8) * 2;
      auto hrow = [&](int yr, float (&hv)[8]) {
        float fa[8], fb[8];
        VecWords<T>::unpack(*reinterpret_cast<const uint4*>(p0 + yr * rowb), fa);
        VecWords<T>::unpack(*reinterpret_cast<const uint4*>(p1 + yr * rowb), fb);
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = fi_lerp2(lx0, fa[j], lx1, fb[j]);
      };
      float h0[8], h1[8];
      int cy0 = -1, cy1 = -1;
      T* yo = ybase + (size_t)c * 8;
      for (int r = 0; r < nout; ++r) {
        const int4 tb = s_tab[r];
        const int y0 = __builtin_amdgcn_readfirstlane(tb.x), y1 = __builtin_amdgcn_readfirstlane(tb.y);
        const float ly0 = __int_as_float(tb.z), ly1 = __int_as_float(tb.w);
        if (y0 != cy0) {
          if (y0 == cy1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) h0[j] = h1[j];
          } else {
            hrow(y0, h0);
          }
          cy0 = y0;
        }
        if (y1 != cy1) {
          hrow(y1, h1);
          cy1 = y1;
        }
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fi_lerp2(ly0, h0[j], ly1, h1[j]);
        *reinterpret_cast<uint4*>(yo) = VecWords<T>::pack(o);
        yo += (size_t)Wo * COUT;
      }
    }
  }
}

inline float up_scale(int in) { return in > 1 ? (float)(in - 1) / (float)(2 * in - 1) : 0.f; }   // == ops.hip

int g_rows = 0;                // fi_upfuse_tuning: 0 = default

template <typename T, int CIN, int COUT>
int launch(const UpFuseArgs& a0, hipStream_t st) {
  UpFuseArgs a = a0;
  auto lds_bytes = [&](int R) {
    return (long)(R + 2) * a.w * COUT * 2 + (long)COUT * (CIN * 2 + 16) + 2L * CIN * 4 + COUT * 4 + 2L * R * 16;
  };
  // rows per workgroup (tools/upfbench.py, profiles/r04_upfbench.txt): 3 on the two wide levels -- 40 KB of LDS, three workgroups
  // per CU, one of them always storing (84 x 256^2 32 -> 16: 206 us against 216 at 6; 12 images: 35 against 40) -- and 6 from 128
  // input channels on, where phase 1 is the heavier half and the two halo rows cost more than the occupancy buys (64^2 128 -> 64:
  // 71 us against 76-82; 32^2 256 -> 128, one workgroup per CU beside its 68 KB filter either way: 52 against 65), unless that
  // leaves CUs without a workgroup
  const bool deep = CIN >= 128;
  int R = g_rows > 0 ? g_rows : (deep && (long)a.N * ((a.h + 5) / 6) >= 256 ? 6 : 3);
  if (R > a.h) R = a.h;
  const long budget = deep ? 156 * 1024 : 78 * 1024;
  while (R > 1 && lds_bytes(R) > budget) --R;
  if (lds_bytes(R) > budget) return FI_ERR_UNSUPPORTED;
  if (2 * R > 256) return FI_ERR_UNSUPPORTED;            // the row table is filled by one thread per output row
  a.R = R;
  a.tiles = (a.h + R - 1) / R;
  a.total = a.N * a.tiles;
  a.chunk = (a.total + 7) / 8;
  const void* kern = reinterpret_cast<const void*>(&conv1x1_up2x_kernel<T, CIN, COUT>);
  static bool allowed = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  if (!allowed) return FI_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((conv1x1_up2x_kernel<T, CIN, COUT>), dim3((unsigned)(a.chunk * 8)), dim3(256), (size_t)lds_bytes(R), st, a);
  FI_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int dispatch(const UpFuseArgs& a, int cin, int cout, hipStream_t st) {
  if (cin == 32 && cout == 16) return launch<T, 32, 16>(a, st);
  if (cin == 64 && cout == 32) return launch<T, 64, 32>(a, st);
  if (cin == 128 && cout == 64) return launch<T, 128, 64>(a, st);
  if (cin == 256 && cout == 128) return launch<T, 256, 128>(a, st);
  if (cin == 32 && cout == 32) return launch<T, 32, 32>(a, st);
  if (cin == 64 && cout == 64) return launch<T, 64, 64>(a, st);
  return FI_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int fi_upfuse_tuning(int rows) {
  g_rows = rows;
  return 0;
}

extern "C" int fi_conv1x1_up2x_fwd(int dtype, int N, int h, int w, int cin, int cout, const FiInXform* t0, int group_images,
                                   const void* x, const void* wmat, const float* bias, void* y, void* stream) {
  if (!x || !wmat || !y) return FI_ERR_NULL;
  if (N < 1 || h < 1 || w < 1 || group_images < 0 || (group_images > 0 && N % group_images)) return FI_ERR_SHAPE;
  if (dtype != FI_BF16 && dtype != FI_F16) return FI_ERR_UNSUPPORTED;       // fp32 parity mode keeps the two launches
  if ((long)N * h >= (1L << 28) || (long)w * cout >= (1L << 20)) return FI_ERR_UNSUPPORTED;
  UpFuseArgs a{};
  a.x = x, a.wmat = wmat, a.bias = bias, a.y = y;
  if (t0 && t0->scale) {
    if (t0->pool || t0->drop_mode != FI_DROP_NONE || !t0->shift) return FI_ERR_UNSUPPORTED;
    if (t0->slope < 0.f || t0->slope > 1.f) return FI_ERR_UNSUPPORTED;
    a.scale = t0->scale, a.shift = t0->shift, a.slope = t0->slope;
  }
  a.N = N, a.h = h, a.w = w, a.gimages = (t0 && t0->scale) ? group_images : 0;
  a.sh = up_scale(h), a.sw = up_scale(w);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FI_BF16) return dispatch<bf16_t>(a, cin, cout, st);
  return dispatch<f16_t>(a, cin, cout, st);
}
