// Implicit-GEMM 2D convolution for gfx950 (forward / dgrad share one kernel; wgrad is its own).
//
// Mapping onto the hardware (DESIGN.md section "conv"):
//   GEMM view  M = output pixels, N = Cout, K = k*k*Cin.
//   One 256-thread workgroup (4 waves) owns a TH x 16 pixel tile of one image and a BN = 16*NF
//   slab of output channels.  Per Cin chunk (CK channels) the (TH+2) x 18 input halo tile and the
//   [BN][k*k*CK] weight slab are staged ONCE through LDS with 16-byte coalesced NHWC loads; the
//   k*k taps are then shifted LDS reads, so each activation byte leaves HBM/L2 once per
//   workgroup.  A wave owns MF = TH/4 rows of the tile; one MFMA "M" fragment is the 16
//   consecutive pixels of a row.  MFMA: v_mfma_f32_16x16x32_bf16 (bf16) or the exact-fp32
//   v_mfma_f32_16x16x4_f32; accumulators are fp32 in both.
//   Epilogue: + bias, optional fp32 store, optional read-modify-write (gradient accumulation),
//   optional per-channel (sum, sum^2) for training-mode BatchNorm: per-lane partials ->
//   wave shuffle over the 4 row groups -> LDS over the 4 waves -> one fp64 atomic per channel.
//   The two-source loader folds torch.cat([skip, up], 1) into the gather; the two-destination
//   epilogue is its adjoint for dgrad.
#pragma once
#include <type_traits>

#include "common.h"

// Input transform applied by the loader of the fused ("probe") forward: the source holds the RAW output y of the
// producing convolution and the consumer evaluates z = dropout(act(scale[c] * y + shift[c])) -- BatchNorm apply,
// LeakyReLU / ReLU and element-wise dropout -- while it stages the tile, optionally followed by the 2x2 max-pool of a
// DownBlock (the source is then [N][2H][2W][C]).  z is rounded to the storage dtype exactly as fi_bn_act_fwd rounds it,
// so the fused forward reproduces the unfused one bit for bit.  scale == NULL: the source is used as it is.
struct InXform {
  const float* scale;        // [groups][C] or NULL
  const float* shift;        // [groups][C]
  float slope;
  int drop_mode;             // FI_DROP_NONE or FI_DROP_RNG_ELEM (source 0 only)
  uint32_t thresh;
  float keep_scale;
  uint64_t seed, seed_gstride;   // group g draws with seed + g * seed_gstride
  const int32_t* seed_offset;
};

struct ConvArgs {
  InXform t0, t1;
  int xf;                    // host side: which loader instantiation launch_conv_fwd picks (0 / 1 / 2 = + max-pool)
  int bcast0;                // source 0 holds ONE group ([gimages] images) read by every group: source image = n % gimages
  int gimages;               // images per group (0: one group): coefficient / seed / statistics group of image n = n / gimages
  long stats_gstride;        // doubles between the statistics accumulators of consecutive groups
  const void* x0;
  const void* x1;
  const void* w;
  const float* bias;
  void* y0;
  void* y1;
  double* stats;
  int N, H, W;
  int c0, c1, co0, co1;
  int acc0, acc1, y_f32;
  int tilesX, tilesY, nct;
  int wrows;                 // conv_fwd_ws2_kernel: output rows per 16-channel chunk of the chunk-major operand `w` then points to
  int depth;                 // > 0 (wave-specialised form only): 3x3x3 convolution over volumes of `depth` consecutive slices --
                             // an "image" is a slice, the depth taps are three channel groups of the contraction (K =
                             // 9 taps x 3 depth taps x (c0 + c1) channels, filter [Cout][9][3][c0 + c1]); slices outside the
                             // volume read as zero
#ifdef FI_TRACE
  long long* trace;   // [workgroups][8] timestamps (tools/ktrace.py); debug builds only
#endif
};

#ifdef FI_TRACE
#define FI_TR(slot)                                                                                        \
  do {                                                                                                     \
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define FI_TR(slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// staging helpers
// ---------------------------------------------------------------------------------------------
// Load VG consecutive input channels [ci, ci+VG) of pixel `pix` (flat n*H*W index) from the
// two-source concatenation; zero beyond cin.  `vec_ok`: c0 % VG == 0 && c1 % VG == 0.
template <typename T>
__device__ __forceinline__ typename DT<T>::vec_t load_cat(const T* __restrict__ x0, const T* __restrict__ x1, int c0,
                                                          int c1, size_t pix, int ci, bool vec_ok) {
  constexpr int VG = DT<T>::VG;
  typedef typename DT<T>::vec_t vec_t;
  const int cin = c0 + c1;
  if (vec_ok) {
    if (ci >= cin) {
      vec_t z;
      memset(&z, 0, sizeof(z));
      return z;
    }
    const T* src = (ci < c0) ? x0 + pix * c0 + ci : x1 + pix * c1 + (ci - c0);
    return *reinterpret_cast<const vec_t*>(src);
  }
  union {
    vec_t v;
    T e[VG];
  } u;
#pragma unroll
  for (int j = 0; j < VG; ++j) {
    const int c = ci + j;
    T val = from_f32<T>(0.f);
    if (c < cin) val = (c < c0) ? x0[pix * c0 + c] : x1[pix * c1 + (c - c0)];
    u.e[j] = val;
  }
  return u.v;
}

// The same vector as load_cat for channel counts that are whole vectors, as an UNCONDITIONAL load from a clamped
// address that is zeroed afterwards when (gy, gx) lies outside the image, ci beyond cin or !live.  A load under a
// divergent branch makes the compiler wait for it at the merge point: the wgrad tile fetch (10 vectors per thread)
// then paid one memory round trip per vector -- 4-7 us per tile in tools/ktrace.py.
template <typename T>
__device__ __forceinline__ typename DT<T>::vec_t load_cat_clamped(const T* __restrict__ x0, const T* __restrict__ x1,
                                                                  int c0, int c1, int n, int gy, int gx, int H, int W,
                                                                  int ci, bool live) {
  typedef typename DT<T>::vec_t vec_t;
  const int cin = c0 + c1;
  const bool ok = live && gy >= 0 && gy < H && gx >= 0 && gx < W && ci < cin;
  const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
  const int cc = ci < cin ? ci : 0;
  const bool first = cc < c0;
  const T* base = first ? x0 + cc : x1 + (cc - c0);
  const int cs = first ? c0 : c1;
  vec_t v = *reinterpret_cast<const vec_t*>(base + ((size_t)(n * H + cy) * W + cx) * cs);
  if (!ok) memset(&v, 0, sizeof(v));
  return v;
}

// ---------------------------------------------------------------------------------------------
// forward / dgrad kernel
// ---------------------------------------------------------------------------------------------
// Register count, not LDS, decides how many workgroups share a CU here, and these kernels live off other workgroups
// hiding their staging round trips (tools/ktrace.py: 3-5 us of a 6-8 us workgroup is the wait for its tile).  Three
// things keep the count down (U-Net convs 355+281 -> 285+229 us fwd+dgrad):
//  * the tap / K-step loop around the MFMAs is NOT unrolled: unrolled, the compiler hoists every LDS fragment read
//    of the chunk to the top (25 fragments = 100 VGPRs for a 16-row tile);
//  * the element-wise loader for channel counts that are not whole 16-byte vectors exists in the narrowest-chunk
//    instantiation only (VEC below; host check), not as a never-taken branch in every kernel;
//  * PLAIN: one output dtype, no accumulate, whole 4-channel groups -> the epilogue is one vector store per fragment;
//    fp32 logits, accumulating dgrads and odd channel counts take the generic instantiation.
// Tried and not kept: persistent workgroups that keep the weight slab in LDS and prefetch their next tile into
// registers (single-chunk layers): the prefetch costs 40 VGPRs, residency drops, 256^2 16->16 21.6 us either way and
// 32->16 27.7 -> 52.8 us.
#ifndef FI_MMA_UNROLL
#define FI_MMA_UNROLL 1
#endif
// LDS strides of the forward kernel's tiles.  The MFMA operand reads are ds_read_b128 with lane = kg*16 + li: li walks
// pixels (or weight rows), kg the 16-byte contraction groups.  The hardware serves a b128 read in 16-lane groups that mix
// TWO kg values ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md), so a group is conflict-free iff the 16 start banks
// S*li (+4 for the second kg) are distinct multiples of 4 -- i.e. iff the stride is 32 bytes mod 64.  The old padding
// (one 16-byte vector: 48 / 80 / 144-byte pixels) measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.4.
template <typename T, int N> struct FiLdsStride {
  static constexpr int value =
      sizeof(T) == 2 ? N + ((32 - (N * 2) % 64 + 64) % 64) / 2 : N + DT<T>::VG;       // elements
};

// XF: 0 = plain loader, 1 = loader applies the InXform of each source, 2 = InXform + 2x2 max-pool of source 0 (c1 == 0)
template <typename T, int KS, int TH, int NF, int CK, bool PLAIN, int XF = 0>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvArgs a) {
  constexpr int HALO = KS / 2, XW = 16 + 2 * HALO, XH = TH + 2 * HALO, KK = KS * KS;
  constexpr int VG = DT<T>::VG, KSTEP = DT<T>::KSTEP, KV = DT<T>::KV;
  constexpr int CKP = FiLdsStride<T, CK>::value;                // padded pixel stride in LDS
  constexpr int KC = KK * CK;                                   // contraction length per chunk
  constexpr int KCP = ((KC + KSTEP - 1) / KSTEP) * KSTEP;       // rounded up to whole MFMAs
  constexpr int WKP = FiLdsStride<T, KCP>::value;               // padded weight-row stride
  constexpr int BN = NF * 16;
  constexpr int MF = TH / 4;
  constexpr int VPP = CK / VG;                                  // 16-byte vectors per pixel
  constexpr int NXV = XH * XW * VPP, NWV = BN * KK * VPP;       // vectors per staged chunk
  constexpr int NX = (NXV + 255) / 256, NW = (NWV + 255) / 256; // ... per thread
  typedef typename DT<T>::vec_t vec_t;
  typedef typename DT<T>::frag_t frag_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* xs = reinterpret_cast<T*>(smem);                           // [XH*XW][CKP]
  T* ws = xs + XH * XW * CKP;                                   // [BN][WKP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;
  const int cin = a.c0 + a.c1, cout = a.co0 + a.co1;
  // Workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  Re-number them so that one XCD walks
  // a contiguous run of (tile, cout-slab) pairs: the nct slabs that re-stage the SAME input tile (and the
  // neighbouring tiles that share its halo) then hit that XCD's L2 instead of fetching the tile once per XCD.
  int bid;
  {
    const unsigned B = gridDim.x, g = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const unsigned Bq = B >> 3, r = B & 7u;
    bid = (int)(g * Bq + (g < r ? g : r) + q);
  }
  const int ct = bid % a.nct;
  bid /= a.nct;
  const int tx = bid % a.tilesX;
  bid /= a.tilesX;
  const int ty = bid % a.tilesY;
  const int n = bid / a.tilesY;
  const int H = a.H, W = a.W;
  const T* x0 = reinterpret_cast<const T*>(a.x0);
  const T* x1 = reinterpret_cast<const T*>(a.x1);
  const T* wg = reinterpret_cast<const T*>(a.w);
  constexpr bool VEC = CK > VG;   // chunks wider than one vector: channel counts are whole vectors (host check)
  static_assert(XF == 0 || VEC, "the transforming loader exists for whole-vector channel counts only");
  const int grp = a.gimages > 0 ? n / a.gimages : 0;
  FI_TR(0);
#ifdef FI_TRACE
  if (a.trace && threadIdx.x == 0) {
    a.trace[(size_t)blockIdx.x * 8 + 6] = (long long)wall_clock64();
    a.trace[(size_t)blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                          ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
  }
#endif

  // acc[m][f][r] = out(channel ct*BN + f*16 + kg*4 + r ; pixel row wave*MF+m, col li): the weights are the
  // MFMA "A" operand and the pixels the "B" operand, so a lane ends up with 4 CONSECUTIVE channels of one
  // pixel -> one 8/16-byte NHWC store per fragment instead of four 2/4-byte ones.
  f32x4 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[m][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging.  A thread owns ONE vector column of the halo tile (pixel column px, channel vector v) and ONE
  //      (tap, channel vector) column of the weight slab for the whole kernel and walks rows with constant
  //      strides -- no div/mod, one bounds compare and one multiply-add per vector.  Loads are unconditional
  //      (clamped address, zeroed at the LDS write): a load under a divergent branch makes the compiler wait for
  //      it at the merge point, which serialises one HBM round trip per vector.
  //      Tried and measured slower on MI355X (tools/kbench.py, LOG.md section 6): prefetching the next chunk
  //      into registers (VGPRs 100 -> 180-240, occupancy halves: 354 -> 499 us over the U-Net forward convs) and
  //      persistent workgroups walking several tiles (354 -> 415 us): many independent workgroups hide the
  //      staging round trips better.
  constexpr int XCOLS = XW * VPP, XRPP = 256 / XCOLS, XPASS = (XH + XRPP - 1) / XRPP;
  constexpr int WCOLS = KK * VPP, WRPP = 256 / WCOLS, WPASS = (BN + WRPP - 1) / WRPP;
  static_assert(XRPP >= 1 && WRPP >= 1, "tile too wide for 256 threads");
  const int xcol = tid % XCOLS, xrow0 = tid / XCOLS;
  const int xpx = xcol / VPP, xv = xcol % VPP;
  const int xgx = tx * 16 + xpx - HALO;
  const bool xcolok = xrow0 < XRPP && xgx >= 0 && xgx < W;
  const int xcx = min(max(xgx, 0), W - 1);
  const int xlds0 = (xrow0 * XW + xpx) * CKP + xv * VG;
  const int wcol = tid % WCOLS, wrow0 = tid / WCOLS;
  const int wt = wcol / VPP, wv = wcol % VPP;
  const int wlds0 = wrow0 * WKP + wt * CK + wv * VG;
  auto stage = [&](int cb) {
    if constexpr (XF == 0) {
      const int ci = cb + xv * VG;
      const bool chok = ci < cin;
      const int cc = chok ? ci : 0;
      const bool first = cc < a.c0;
      const T* colbase = first ? x0 + cc : x1 + (cc - a.c0);
      const int cstride = first ? a.c0 : a.c1;
      vec_t r[XPASS];
      bool ok[XPASS];
#pragma unroll
      for (int p = 0; p < XPASS; ++p) {
        const int py = xrow0 + p * XRPP;
        const int gy = ty * TH + py - HALO;
        ok[p] = xcolok && chok && py < XH && gy >= 0 && gy < H;
        const int cy = min(max(gy, 0), H - 1);
        const int pix = (n * H + cy) * W + xcx;   // < 2^31 pixels per tensor (host-checked)
        if constexpr (VEC) {
          r[p] = *reinterpret_cast<const vec_t*>(colbase + (size_t)pix * cstride);
        } else {
          r[p] = load_cat<T>(x0, x1, a.c0, a.c1, (size_t)pix, ci, false);   // narrow tensors (image, logits grad)
        }
      }
#pragma unroll
      for (int p = 0; p < XPASS; ++p) {
        const int py = xrow0 + p * XRPP;
        if (xrow0 < XRPP && py < XH) {
          vec_t val = r[p];
          if (!ok[p]) memset(&val, 0, sizeof(val));
          *reinterpret_cast<vec_t*>(&xs[xlds0 + p * (XRPP * XW * CKP)]) = val;
        }
      }
    } else {
      // transforming loader: this thread's channel vector keeps ONE set of coefficients per chunk in registers
      const int ci = cb + xv * VG;
      const bool chok = ci < cin;
      const int cc = chok ? ci : 0;
      const bool first = cc < a.c0;
      const int csrc = first ? cc : cc - a.c0;
      const int cstride = first ? a.c0 : a.c1;
      const T* colbase = (first ? x0 : x1) + csrc;
      const float* scp = first ? a.t0.scale : a.t1.scale;
      const float* shp = first ? a.t0.shift : a.t1.shift;
      const float slope = first ? a.t0.slope : a.t1.slope;
      const bool xf = scp != nullptr;
      float sc[VG], sh[VG];
#pragma unroll
      for (int j = 0; j < VG; ++j) {
        sc[j] = xf ? scp[(size_t)grp * cstride + csrc + j] : 1.f;
        sh[j] = xf ? shp[(size_t)grp * cstride + csrc + j] : 0.f;
      }
      const bool drop = XF == 1 && first && a.t0.drop_mode == FI_DROP_RNG_ELEM;
      uint64_t seed = 0;
      if (drop) {
        seed = a.t0.seed + (uint64_t)grp * a.t0.seed_gstride;
        if (a.t0.seed_offset) seed += 0xD1B54A32D192ED03ull * (uint64_t)(uint32_t)a.t0.seed_offset[0];
      }
      auto xform = [&](const vec_t& raw, size_t vecidx) -> vec_t {
        union {
          vec_t v;
          T e[VG];
        } u;
        u.v = raw;
        float f[VG];
#pragma unroll
        for (int j = 0; j < VG; ++j) {
          const float v = to_f32(u.e[j]) * sc[j] + sh[j];
          f[j] = fmaxf(v, v * slope);                  // = v > 0 ? v : v * slope for 0 <= slope <= 1 (host-checked), one select less
        }
        if (drop) {
#pragma unroll
          for (int g4 = 0; g4 < VG / 4; ++g4) {
            uint32_t rr[4];
            fi_rand32x4(seed, vecidx * (VG / 4) + g4, rr);
#pragma unroll
            for (int j = 0; j < 4; ++j) f[g4 * 4 + j] *= rr[j] >= a.t0.thresh ? a.t0.keep_scale : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < VG; ++j) u.e[j] = from_f32<T>(f[j]);
        return u.v;
      };
      // source image: every group reads the same [gimages] images when the source is shared (first conv of the probe batch)
      const int nl = n - grp * a.gimages;
      const int ns = (first && a.bcast0) ? nl : n;
      constexpr int NS = XF == 2 ? 4 : 1;
      vec_t r[XPASS][NS];
      bool ok[XPASS];
      unsigned vix[XPASS];
#pragma unroll
      for (int p = 0; p < XPASS; ++p) {
        const int py = xrow0 + p * XRPP;
        const int gy = ty * TH + py - HALO;
        ok[p] = xcolok && chok && py < XH && gy >= 0 && gy < H;
        const int cy = min(max(gy, 0), H - 1);
        if constexpr (XF == 2) {
          const int pix = (ns * 2 * H + 2 * cy) * (2 * W) + 2 * xcx;     // source is [N][2H][2W][C]
          const T* src = colbase + (size_t)pix * cstride;
          r[p][0] = *reinterpret_cast<const vec_t*>(src);
          r[p][1] = *reinterpret_cast<const vec_t*>(src + cstride);
          r[p][2] = *reinterpret_cast<const vec_t*>(src + (size_t)2 * W * cstride);
          r[p][3] = *reinterpret_cast<const vec_t*>(src + (size_t)(2 * W + 1) * cstride);
          vix[p] = 0;
        } else {
          const int pix = (ns * H + cy) * W + xcx;
          r[p][0] = *reinterpret_cast<const vec_t*>(colbase + (size_t)pix * cstride);
          // dropout index = vector index inside the group's dense [gimages][H][W][C] tensor (what fi_bn_act_fwd uses)
          const int pixl = (nl * H + cy) * W + xcx;
          vix[p] = (unsigned)pixl * (unsigned)(cstride / VG) + (unsigned)(csrc / VG);
        }
      }
#pragma unroll
      for (int p = 0; p < XPASS; ++p) {
        const int py = xrow0 + p * XRPP;
        if (xrow0 < XRPP && py < XH) {
          vec_t val;
          if (!ok[p]) {
            memset(&val, 0, sizeof(val));
          } else if constexpr (XF == 2) {
            union {
              vec_t v;
              T e[VG];
            } q[4], m;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k].v = xf ? xform(r[p][k], 0) : r[p][k];
#pragma unroll
            for (int j = 0; j < VG; ++j) {
              // scan order, strict > : the element fi_maxpool2_fwd keeps
              float best = to_f32(q[0].e[j]);
              T bq = q[0].e[j];
#pragma unroll
              for (int k = 1; k < 4; ++k) {
                const float v = to_f32(q[k].e[j]);
                if (v > best) {
                  best = v;
                  bq = q[k].e[j];
                }
              }
              m.e[j] = bq;
            }
            val = m.v;
          } else {
            val = xf ? xform(r[p][0], vix[p]) : r[p][0];
          }
          *reinterpret_cast<vec_t*>(&xs[xlds0 + p * (XRPP * XW * CKP)]) = val;
        }
      }
    }
    {
      const int ci = cb + wv * VG;
      const bool chok = ci < cin && wrow0 < WRPP;
      const T* colbase = wg + (size_t)wt * cin + (ci < cin ? ci : 0);
      vec_t r[WPASS];
      bool ok[WPASS];
#pragma unroll
      for (int p = 0; p < WPASS; ++p) {
        const int co = wrow0 + p * WRPP;
        const int gco = ct * BN + co;
        ok[p] = chok && co < BN && gco < cout;
        const T* src = colbase + (size_t)(gco < cout ? gco : 0) * (KK * cin);
        if constexpr (VEC) {
          r[p] = *reinterpret_cast<const vec_t*>(src);
        } else {
          union {
            vec_t vv;
            T e[VG];
          } u;
          memset(&u, 0, sizeof(u));
          if (ok[p]) {
#pragma unroll
            for (int q = 0; q < VG; ++q)
              if (ci + q < cin) u.e[q] = src[q];
          }
          r[p] = u.vv;
        }
      }
#pragma unroll
      for (int p = 0; p < WPASS; ++p) {
        const int co = wrow0 + p * WRPP;
        if (wrow0 < WRPP && co < BN) {
          vec_t val = r[p];
          if (!ok[p]) memset(&val, 0, sizeof(val));
          *reinterpret_cast<vec_t*>(&ws[wlds0 + p * (WRPP * WKP)]) = val;
        }
      }
    }
  };

  if (KCP > KC) {  // zero the K padding once: clamped A reads then multiply by 0 (never overwritten by stage)
    constexpr int PV = (KCP - KC) / VG;
    for (int i = tid; i < BN * PV; i += 256) {
      vec_t z;
      memset(&z, 0, sizeof(z));
      *reinterpret_cast<vec_t*>(&ws[(i / PV) * WKP + KC + (i % PV) * VG]) = z;
    }
  }
  for (int cb = 0; cb < cin; cb += CK) {
    __syncthreads();  // everyone finished reading the previous chunk
    stage(cb);
    __syncthreads();
    if (cb == 0) FI_TR(1);

    // ---- MFMA over this chunk
    if constexpr (CK >= KSTEP) {
#pragma unroll FI_MMA_UNROLL
      for (int t = 0; t < KK; ++t) {
        const int r = t / KS, s = t % KS;
#pragma unroll
        for (int ks = 0; ks < CK / KSTEP; ++ks) {
          frag_t b[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f)
            b[f] = *reinterpret_cast<const frag_t*>(&ws[(f * 16 + li) * WKP + t * CK + ks * KSTEP + kg * KV]);
#pragma unroll
          for (int m = 0; m < MF; ++m) {
            const int row = wave * MF + m;
            const frag_t av =
                *reinterpret_cast<const frag_t*>(&xs[((row + r) * XW + li + s) * CKP + ks * KSTEP + kg * KV]);
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[m][f] = mfma16(b[f], av, acc[m][f]);
          }
        }
      }
    } else {
      // CK < KSTEP (bf16 with 8 or 16 staged channels): one MFMA spans several taps; each lane's
      // 8 contraction elements stay inside one tap because CK % 8 == 0.
#pragma unroll FI_MMA_UNROLL
      for (int ks = 0; ks < KCP / KSTEP; ++ks) {
        const int k0 = ks * KSTEP + kg * KV;
        int t = k0 / CK;
        const int cil = k0 % CK;
        if (t > KK - 1) t = KK - 1;  // padded K: weights are zero there, any valid address will do
        const int r = t / KS, s = t % KS;
        frag_t b[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) b[f] = *reinterpret_cast<const frag_t*>(&ws[(f * 16 + li) * WKP + k0]);
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          const int row = wave * MF + m;
          const frag_t av = *reinterpret_cast<const frag_t*>(&xs[((row + r) * XW + li + s) * CKP + cil]);
#pragma unroll
          for (int f = 0; f < NF; ++f) acc[m][f] = mfma16(b[f], av, acc[m][f]);
        }
      }
    }
  }

  FI_TR(3);
  // ---- epilogue
  float ssum[NF][4], ssq[NF][4];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[f][r] = ssq[f][r] = 0.f;
  const int gx = tx * 16 + li;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int cg = ct * BN + f * 16 + kg * 4;  // first of this lane's 4 channels
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (a.bias && cg + r < cout) ? a.bias[cg + r] : 0.f;
    const bool second = cg >= a.co0;
    const int cdst = second ? a.co1 : a.co0;
    const int cofs = second ? cg - a.co0 : cg;
    void* ybase = second ? a.y1 : a.y0;
    const int accum = second ? a.acc1 : a.acc0;
    // whole 4-channel group inside one destination and 4-aligned -> one vector store
    const bool vec4 = PLAIN || ((cg + 3 < cout) && (cdst % 4 == 0) && (cofs % 4 == 0) && (second || cg + 3 < a.co0));
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int gy = ty * TH + wave * MF + m;
      if (gy < H && gx < W && cg < cout) {
        const size_t pixo = ((size_t)n * H + gy) * W + gx;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[m][f][r] + bv[r];
        if (vec4) {
          const size_t o = pixo * cdst + cofs;
          if (!PLAIN && a.y_f32) {
            float4* yp = reinterpret_cast<float4*>(reinterpret_cast<float*>(ybase) + o);
            if (accum && ybase) {
              const float4 old = *yp;
              v[0] += old.x;
              v[1] += old.y;
              v[2] += old.z;
              v[3] += old.w;
            }
            if (ybase) *yp = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            typedef typename Quad<T>::q_t q_t;
            T* yp = reinterpret_cast<T*>(ybase) + o;
            if (!PLAIN && accum && ybase) {
              float old4[4];
              Quad<T>::unpack(*reinterpret_cast<const q_t*>(yp), old4);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += old4[r];
            }
            const q_t q = Quad<T>::pack(v);                              // v := the values as stored
            if (ybase) *reinterpret_cast<q_t*>(yp) = q;                  // NULL destination: statistics-only launch
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ssum[f][r] += v[r];
            ssq[f][r] += v[r] * v[r];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = cg + r;
            if (co < cout) {
              const bool sec = co >= a.co0;
              const int cd = sec ? a.co1 : a.co0;
              void* yb = sec ? a.y1 : a.y0;
              const size_t o = pixo * cd + (sec ? co - a.co0 : co);
              float vv = v[r];
              const int ac = sec ? a.acc1 : a.acc0;
              if (a.y_f32) {
                float* yp = reinterpret_cast<float*>(yb) + o;
                if (ac && yb) vv += *yp;
                if (yb) *yp = vv;
              } else {
                T* yp = reinterpret_cast<T*>(yb) + o;
                if (ac && yb) vv += to_f32(*yp);
                const T q = from_f32<T>(vv);
                if (yb) *yp = q;
                vv = to_f32(q);
              }
              ssum[f][r] += vv;
              ssq[f][r] += vv * vv;
            }
          }
        }
      }
    }
  }
  FI_TR(4);
  if (a.stats) {
    // LDS reuse: all MFMA reads are done.  LDS-only barriers here: __syncthreads() would also wait (vmcnt(0)) until the
    // output stores just issued are acknowledged -- ~1 us during which the workgroup holds its registers and LDS.
    fi_lds_barrier();
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][BN][2]
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = fi_row16_sum(ssum[f][r]), q = fi_row16_sum(ssq[f][r]);   // over the 16 pixel lanes of this kg group
        if (li == 0) {
          red[(wave * BN + f * 16 + kg * 4 + r) * 2 + 0] = s;
          red[(wave * BN + f * 16 + kg * 4 + r) * 2 + 1] = q;
        }
      }
    fi_lds_barrier();
    if (tid < BN * 2) {
      const int c = tid >> 1, which = tid & 1;
      const int co = ct * BN + c;
      if (co < cout) {
        double tot = 0.0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) tot += (double)red[(wv * BN + c) * 2 + which];
        // FI_STATS_SLOTS copies of the accumulator: thousands of workgroups adding to the same 2*C
        // addresses serialise at the fabric; spreading them over 32 slots (summed by the BN kernel) removes that.
        const int slot = blockIdx.x & (FI_STATS_SLOTS - 1);
        atomicAdd(&a.stats[(size_t)grp * a.stats_gstride + ((size_t)slot * cout + co) * 2 + which], tot);
      }
    }
  }
  FI_TR(5);
#ifdef FI_TRACE
  if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 8 + 2] = (long long)wall_clock64();
#endif
}

template <typename T, int KS, int TH, int NF, int CK>
static int launch_conv_fwd(const ConvArgs& a, hipStream_t st) {
  const int xf = a.xf;
  constexpr int HALO = KS / 2, XW = 16 + 2 * HALO, XH = TH + 2 * HALO, KK = KS * KS;
  constexpr int VG = DT<T>::VG, KSTEP = DT<T>::KSTEP;
  constexpr int KC = KK * CK, KCP = ((KC + KSTEP - 1) / KSTEP) * KSTEP;
  constexpr int CKP = FiLdsStride<T, CK>::value, WKP = FiLdsStride<T, KCP>::value;
  constexpr int BN = NF * 16;
  size_t lds = (size_t)(XH * XW * CKP + BN * WKP) * sizeof(T);
  const size_t red = (size_t)4 * BN * 2 * sizeof(float);
  if (lds < red) lds = red;
  const long blocks = (long)a.N * a.tilesX * a.tilesY * a.nct;
  const bool plain = !a.y_f32 && !a.acc0 && !a.acc1 && a.co0 % 4 == 0 && a.co1 % 4 == 0;
  if constexpr (CK > DT<T>::VG) {
    // fused-forward loaders: plain epilogue only (the probe forward stores one dtype, never accumulates)
    if (xf) {
      if (!plain) return FI_ERR_UNSUPPORTED;
      if (xf == 2) {
        if constexpr (KS == 3)
          hipLaunchKernelGGL((conv_fwd_kernel<T, KS, TH, NF, CK, true, 2>), dim3((unsigned)blocks), dim3(256), lds, st, a);
        else
          return FI_ERR_UNSUPPORTED;
      } else {
        hipLaunchKernelGGL((conv_fwd_kernel<T, KS, TH, NF, CK, true, 1>), dim3((unsigned)blocks), dim3(256), lds, st, a);
      }
      FI_CHECK_LAUNCH();
      return 0;
    }
  } else {
    if (xf) return FI_ERR_UNSUPPORTED;
  }
  if (plain)
    hipLaunchKernelGGL((conv_fwd_kernel<T, KS, TH, NF, CK, true>), dim3((unsigned)blocks), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((conv_fwd_kernel<T, KS, TH, NF, CK, false>), dim3((unsigned)blocks), dim3(256), lds, st, a);
  FI_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// forward / dgrad kernel, second form: PERSISTENT workgroups with the next stage in flight
// ---------------------------------------------------------------------------------------------
// The one-tile kernel above hides a tile's staging round trip behind OTHER resident workgroups, which needs many of them:
// small register tiles (most layers run one 16-channel output fragment per wave), the input tile re-staged once per
// 16-channel slab, 3-5 us of every 6-10 us workgroup life spent waiting.  This form turns that around for the layers
// where it pays (measured per layer, conv_api.hip picks): a workgroup walks MANY (tile, chunk) stages, and while the
// MFMAs of stage s run from LDS the global loads of stage s+1 are already in flight into registers (one register set:
// committed to LDS after the barrier that ends stage s, re-issued at once for stage s+2).  With the latency covered
// inside the workgroup it can afford the big register tile -- 64 pixels x 16*NF channels per wave, 16*NF MFMAs per
// 4 + NF LDS fragment reads -- and an unrolled tap loop, at 1-2 workgroups per CU.
//   * a workgroup keeps ONE output-channel slab for all its tiles: single-chunk layers (Cin <= CK) stage their weights
//     once per workgroup, not once per tile;
//   * the epilogue of a finished tile is issued AFTER the next stage's LDS commit and the re-issue of the loads behind
//     it: its global stores then sit behind those loads in the (in-order, shared) vmcnt queue and are a whole MFMA phase
//     old by the time anything waits on that queue -- the wait for the next tile never waits for store acknowledgements;
//   * XCD-aware: XCD x (= blockIdx % 8) owns the x-th eighth of the tiles and its workgroups stride through it together,
//     so the halo overlap of neighbouring tiles and the slabs re-reading one tile meet in that XCD's L2.
// Same LDS layouts, MFMA operand mapping, input transforms (XF) and plain epilogue (bias, one vector store per fragment,
// BatchNorm statistics) as conv_fwd_kernel; 16-row tiles only.
#ifndef FI_V2_UNROLL
#define FI_V2_UNROLL 3
#endif
template <typename T, int KS, int NF, int CK, int XF>
__global__ __launch_bounds__(256, 2) void conv_fwd_v2_kernel(ConvArgs a) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  constexpr int TH = 16;
  constexpr int HALO = KS / 2, XW = 16 + 2 * HALO, XH = TH + 2 * HALO, KK = KS * KS;
  constexpr int VG = DT<T>::VG, KSTEP = DT<T>::KSTEP, KV = DT<T>::KV;
  constexpr int CKP = FiLdsStride<T, CK>::value;
  constexpr int KC = KK * CK;
  constexpr int KCP = ((KC + KSTEP - 1) / KSTEP) * KSTEP;
  constexpr int WKP = FiLdsStride<T, KCP>::value;
  constexpr int BN = NF * 16;
  constexpr int MF = TH / 4;
  constexpr int VPP = CK / VG;
  static_assert(CK > VG, "whole-vector channel counts only");
  typedef typename DT<T>::vec_t vec_t;
  typedef typename DT<T>::frag_t frag_t;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef unsigned v2u __attribute__((ext_vector_type(2)));

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* xs = reinterpret_cast<T*>(smem);                           // [XH*XW][CKP]
  T* ws = xs + XH * XW * CKP;                                   // [BN][WKP]
  float* red = reinterpret_cast<float*>(ws + BN * WKP);         // [4 waves][BN][2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kg = lane >> 4;
  const int cin = a.c0 + a.c1, cout = a.co0 + a.co1;
  const int H = a.H, W = a.W;

  // ---- this workgroup's work: tiles t_first + k*tstep (k = 0, 1, ...) below t_end, output slab ct, nchunk chunks per tile
  const int ntile = a.N * a.tilesY * a.tilesX;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, gxd = gridDim.x >> 3;    // grid: a multiple of 8 * nct (host)
  const int tpx = (ntile + 7) >> 3;
  const int t_end = min(ntile, (xcd + 1) * tpx);
  const int ct = q % a.nct, tstep = gxd / a.nct;
  const int t_first = xcd * tpx + q / a.nct;
  if (t_first >= t_end) return;
  const int nchunk = (cin + CK - 1) / CK;
  const bool w_resident = nchunk == 1;

  // ---- buffer resources (raw, range-checked: an offset beyond num_records reads zeros / drops the store): no clamped
  //      addresses, no 64-bit address arithmetic, no zero-selects in the plain loader (see conv_thin_kernel)
  constexpr unsigned esz = sizeof(T);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const unsigned hw = (unsigned)H * (unsigned)W;
  const unsigned img0 = (XF != 0 && a.bcast0) ? (unsigned)a.gimages : (unsigned)a.N;
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.x0), 0, img0 * (XF == 2 ? 4u : 1u) * hw * (unsigned)a.c0 * esz, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.x1), 0, a.c1 ? (unsigned)a.N * hw * (unsigned)a.c1 * esz : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.w), 0, (unsigned)cout * KK * (unsigned)cin * esz, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry0 = __builtin_amdgcn_make_buffer_rsrc(
      a.y0, 0, a.y0 ? (unsigned)a.N * hw * (unsigned)a.co0 * esz : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry1 = __builtin_amdgcn_make_buffer_rsrc(
      a.y1, 0, (a.y1 && a.co1) ? (unsigned)a.N * hw * (unsigned)a.co1 * esz : 0u, 0x00020000);

  // ---- staging geometry (a thread owns one vector column of the halo tile / of the weight slab)
  constexpr int XCOLS = XW * VPP, XRPP = 256 / XCOLS, XPASS = (XH + XRPP - 1) / XRPP;
  constexpr int WCOLS = KK * VPP, WRPP = 256 / WCOLS, WPASS = (BN + WRPP - 1) / WRPP;
  static_assert(XRPP >= 1 && WRPP >= 1, "tile too wide for 256 threads");
  constexpr int NS = XF == 2 ? 4 : 1;
  // Every phase recomputes its thread geometry from an OPAQUE copy of the thread id: derived from `tid` directly, hipcc
  // hoists a few dozen per-thread invariants (row offsets, LDS addresses, masks) of all phases out of the stage loop and
  // keeps them live across it -- with the register set in flight and 64 accumulators that spills (NF = 4).
#define FI_V2_GEOMETRY()                                                     \
  int tid_ = tid;                                                            \
  asm volatile("" : "+v"(tid_));                                             \
  const int xcol = tid_ % XCOLS, xrow0 = tid_ / XCOLS;                       \
  const int xpx = xcol / VPP, xv = xcol % VPP;                               \
  const int xlds0 = (xrow0 * XW + xpx) * CKP + xv * VG;                      \
  const int wcol = tid_ % WCOLS, wrow0 = tid_ / WCOLS;                       \
  const int wt = wcol / VPP, wv = wcol % VPP;                                \
  const int wlds0 = wrow0 * WKP + wt * CK + wv * VG;                         \
  (void)xlds0; (void)wlds0; (void)xrow0; (void)wrow0; (void)xpx; (void)wt; (void)xv; (void)wv

  struct Tile {
    int tx, ty, n, grp;
  };
  auto tile_at = [&](int tile) {
    Tile c;
    c.tx = tile % a.tilesX;
    c.ty = (tile / a.tilesX) % a.tilesY;
    c.n = tile / (a.tilesX * a.tilesY);
    c.grp = a.gimages > 0 ? c.n / a.gimages : 0;
    return c;
  };

  vec_t xr[XPASS][NS];          // the ONE register set of the stage in flight
  vec_t wr[WPASS];

  // per-chunk source of this thread's channel vector
  struct Src {
    bool first, chok;
    unsigned cs, co;            // channels of the source tensor, first channel inside it
  };
  auto src_of = [&](int cb, int xv, int xrow0) {
    Src s;
    const int ci = cb + xv * VG;
    s.chok = ci < cin && xrow0 < XRPP;
    const int cc = ci < cin ? ci : 0;
    s.first = cc < a.c0;
    s.cs = (unsigned)(s.first ? a.c0 : a.c1);
    s.co = (unsigned)(s.first ? cc : cc - a.c0);
    return s;
  };

  auto issue = [&](const Tile& tc, int cb, bool with_w) __attribute__((always_inline)) {
    FI_V2_GEOMETRY();
    const Src s = src_of(cb, xv, xrow0);
    const int gx = tc.tx * 16 + xpx - HALO;
    const bool colok = s.chok && gx >= 0 && gx < W;
    const int ns = (XF != 0 && s.first && a.bcast0) ? tc.n - tc.grp * a.gimages : tc.n;
#pragma unroll
    for (int p = 0; p < XPASS; ++p) {
      const int gy = tc.ty * TH + xrow0 + p * XRPP - HALO;
      const bool ok = colok && gy >= 0 && gy < H;
      if constexpr (XF == 2) {
        const unsigned o = ((unsigned)((ns * 2 * H + 2 * gy) * (2 * W) + 2 * gx) * s.cs + s.co) * esz;
        const unsigned rowb = (unsigned)(2 * W) * s.cs * esz, pxb = s.cs * esz;
        xr[p][0] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(r0, ok ? o : OOB, 0, 0));
        xr[p][1] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(r0, ok ? o + pxb : OOB, 0, 0));
        xr[p][2] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(r0, ok ? o + rowb : OOB, 0, 0));
        xr[p][3] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(r0, ok ? o + rowb + pxb : OOB, 0, 0));
      } else {
        const unsigned o = ((unsigned)((ns * H + gy) * W + gx) * s.cs + s.co) * esz;
        // (a per-lane choice of buffer resource would be lowered to a readfirstlane loop around the load)
        if (a.c1 == 0) {
          xr[p][0] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(r0, ok ? o : OOB, 0, 0));
        } else {
          const T* const xb = reinterpret_cast<const T*>(s.first ? a.x0 : a.x1);
          xr[p][0] = *reinterpret_cast<const vec_t*>(xb + (ok ? o / esz : 0u));       // zeroed at commit when !ok
        }
      }
    }
    if (with_w) {
      const int ci = cb + wv * VG;
      const bool chok = ci < cin && wrow0 < WRPP;
#pragma unroll
      for (int p = 0; p < WPASS; ++p) {
        const int co = wrow0 + p * WRPP, gco = ct * BN + co;
        const unsigned o = ((unsigned)(gco * KK + wt) * (unsigned)cin + (unsigned)ci) * esz;
        wr[p] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rw, (chok && co < BN && gco < cout) ? o : OOB, 0, 0));
      }
    }
  };

  auto commit = [&](const Tile& tc, int cb, bool with_w) __attribute__((always_inline)) {
    FI_V2_GEOMETRY();
    const Src s = src_of(cb, xv, xrow0);
    float sc[VG], sh[VG];
    bool xf = false, drop = false;
    float slope = 1.f;
    uint64_t seed = 0;
    if constexpr (XF != 0) {
      const float* scp = s.first ? a.t0.scale : a.t1.scale;
      const float* shp = s.first ? a.t0.shift : a.t1.shift;
      slope = s.first ? a.t0.slope : a.t1.slope;
      xf = scp != nullptr;
#pragma unroll
      for (int j = 0; j < VG; ++j) {
        sc[j] = xf ? scp[(unsigned)tc.grp * s.cs + s.co + j] : 1.f;
        sh[j] = xf ? shp[(unsigned)tc.grp * s.cs + s.co + j] : 0.f;
      }
      drop = XF == 1 && s.first && a.t0.drop_mode == FI_DROP_RNG_ELEM;
      if (drop) {
        seed = a.t0.seed + (uint64_t)tc.grp * a.t0.seed_gstride;
        if (a.t0.seed_offset) seed += 0xD1B54A32D192ED03ull * (uint64_t)(uint32_t)a.t0.seed_offset[0];
      }
    }
    auto xform = [&](const vec_t& raw, size_t vecidx) __attribute__((always_inline)) -> vec_t {
      float f[VG];
      VecWords<T>::unpack(raw, f);
#pragma unroll
      for (int j = 0; j < VG; ++j) {
        const float v = f[j] * sc[j] + sh[j];
        f[j] = fmaxf(v, v * slope);                  // = v > 0 ? v : v * slope for 0 <= slope <= 1 (host-checked)
      }
      if (drop) {
#pragma unroll
        for (int g4 = 0; g4 < VG / 4; ++g4) {
          uint32_t rr[4];
          fi_rand32x4(seed, vecidx * (VG / 4) + g4, rr);
#pragma unroll
          for (int j = 0; j < 4; ++j) f[g4 * 4 + j] *= rr[j] >= a.t0.thresh ? a.t0.keep_scale : 0.f;
        }
      }
      return VecWords<T>::pack(f);
    };
    const int gx = tc.tx * 16 + xpx - HALO;
    const bool colok = s.chok && gx >= 0 && gx < W;
    const int nl = tc.n - tc.grp * a.gimages;
#pragma unroll
    for (int p = 0; p < XPASS; ++p) {
      const int py = xrow0 + p * XRPP;
      const int gy = tc.ty * TH + py - HALO;
      if (xrow0 < XRPP && py < XH) {
        vec_t val;
        if constexpr (XF == 0) {
          // one source: out-of-image / beyond-Cin vectors arrived as zeros (range-checked buffer loads)
          val = a.c1 == 0 ? xr[p][0] : fi_vec_select(colok && gy >= 0 && gy < H, xr[p][0]);
        } else {
          const bool ok = colok && gy >= 0 && gy < H;           // z of the padding is 0, not act(shift)
          if constexpr (XF == 2) {
            float best[VG], cand[VG];
            VecWords<T>::unpack(xf ? xform(xr[p][0], 0) : xr[p][0], best);
#pragma unroll
            for (int k = 1; k < 4; ++k) {
              VecWords<T>::unpack(xf ? xform(xr[p][k], 0) : xr[p][k], cand);
#pragma unroll
              for (int j = 0; j < VG; ++j) best[j] = cand[j] > best[j] ? cand[j] : best[j];
            }
            val = fi_vec_select(ok, VecWords<T>::pack(best));
          } else {
            const unsigned vix = (unsigned)((nl * H + gy) * W + gx) * (s.cs / VG) + s.co / VG;
            val = fi_vec_select(ok, xf ? xform(xr[p][0], vix) : xr[p][0]);
          }
        }
        *reinterpret_cast<vec_t*>(&xs[xlds0 + p * (XRPP * XW * CKP)]) = val;
      }
    }
    if (with_w) {
#pragma unroll
      for (int p = 0; p < WPASS; ++p) {
        const int co = wrow0 + p * WRPP;
        if (wrow0 < WRPP && co < BN) *reinterpret_cast<vec_t*>(&ws[wlds0 + p * (WRPP * WKP)]) = wr[p];
      }
    }
  };

  f32x4 acc[MF][NF];
  auto mma = [&]() __attribute__((always_inline)) {
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    const int li = lane_ & 15, kg = lane_ >> 4;
    if constexpr (CK >= KSTEP) {
#pragma unroll FI_V2_UNROLL
      for (int t = 0; t < KK; ++t) {
        const int r = t / KS, s = t % KS;
#pragma unroll
        for (int ks = 0; ks < CK / KSTEP; ++ks) {
          frag_t b[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f)
            b[f] = *reinterpret_cast<const frag_t*>(&ws[(f * 16 + li) * WKP + t * CK + ks * KSTEP + kg * KV]);
#pragma unroll
          for (int m = 0; m < MF; ++m) {
            const int row = wave * MF + m;
            const frag_t av =
                *reinterpret_cast<const frag_t*>(&xs[((row + r) * XW + li + s) * CKP + ks * KSTEP + kg * KV]);
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[m][f] = mfma16(b[f], av, acc[m][f]);
          }
        }
      }
    } else {
#pragma unroll 2
      for (int ks = 0; ks < KCP / KSTEP; ++ks) {
        const int k0 = ks * KSTEP + kg * KV;
        int t = k0 / CK;
        const int cil = k0 % CK;
        if (t > KK - 1) t = KK - 1;
        const int r = t / KS, s = t % KS;
        frag_t b[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) b[f] = *reinterpret_cast<const frag_t*>(&ws[(f * 16 + li) * WKP + k0]);
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          const int row = wave * MF + m;
          const frag_t av = *reinterpret_cast<const frag_t*>(&xs[((row + r) * XW + li + s) * CKP + cil]);
#pragma unroll
          for (int f = 0; f < NF; ++f) acc[m][f] = mfma16(b[f], av, acc[m][f]);
        }
      }
    }
  };

  // BatchNorm statistics of the stored values: per-lane partial sums over all tiles of the current group, one cross-lane
  // reduction and one round of fp64 atomics per group and workgroup
  // (the four-fragment instantiations have no 48 registers to spare for that and for the bias: they reduce per tile)
  constexpr bool DEFER = NF < 4;
  float ssum[NF][4], ssq[NF][4];
  auto stats_clear = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) ssum[f][r] = ssq[f][r] = 0.f;
  };
  auto stats_flush = [&](int grp) __attribute__((always_inline)) {
    if (!a.stats) return;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = fi_row16_sum(ssum[f][r]), qv = fi_row16_sum(ssq[f][r]);
        if (li == 0) {
          red[(wave * BN + f * 16 + kg * 4 + r) * 2 + 0] = s;
          red[(wave * BN + f * 16 + kg * 4 + r) * 2 + 1] = qv;
        }
      }
    fi_lds_barrier();
    if (tid < BN * 2) {
      const int c = tid >> 1, which = tid & 1;
      const int co = ct * BN + c;
      if (co < cout) {
        double tot = 0.0;
#pragma unroll
        for (int wv_ = 0; wv_ < 4; ++wv_) tot += (double)red[(wv_ * BN + c) * 2 + which];
        const int slot = blockIdx.x & (FI_STATS_SLOTS - 1);
        atomicAdd(&a.stats[(size_t)grp * a.stats_gstride + ((size_t)slot * cout + co) * 2 + which], tot);
      }
    }
    fi_lds_barrier();
  };

  float bv[DEFER ? NF : 1][4];
  if constexpr (DEFER) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = ct * BN + f * 16 + kg * 4 + r;
        bv[f][r] = (a.bias && co < cout) ? a.bias[co] : 0.f;
      }
  }

  auto epilogue = [&](const Tile& tc) __attribute__((always_inline)) {
    if constexpr (!DEFER) stats_clear();
    const int gx = tc.tx * 16 + li;
    const bool colok = gx < W;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int gy = tc.ty * TH + wave * MF + m;
      const bool ok = colok && gy < H;
      const float mk = ok ? 1.f : 0.f;
      const unsigned pix = (unsigned)((tc.n * H + gy) * W + gx);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int cg = ct * BN + f * 16 + kg * 4;                // whole 4-channel groups inside one destination (host)
        const bool second = cg >= a.co0;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float bias_r;
          if constexpr (DEFER)
            bias_r = bv[f][r];
          else
            bias_r = (a.bias && cg + r < cout) ? a.bias[cg + r] : 0.f;
          v[r] = acc[m][f][r] + bias_r;
        }
        const v2u qv = __builtin_bit_cast(v2u, Quad<T>::pack(v));        // v := the values as stored
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= mk;                         // tile overhang does not count
        const bool live = ok && cg < cout;
        if (second)
          __builtin_amdgcn_raw_buffer_store_b64(qv, ry1, live ? (pix * (unsigned)a.co1 + (unsigned)(cg - a.co0)) * esz : OOB, 0, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b64(qv, ry0, live ? (pix * (unsigned)a.co0 + (unsigned)cg) * esz : OOB, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ssum[f][r] += v[r];
          ssq[f][r] += v[r] * v[r];
        }
      }
    }
    if constexpr (!DEFER) stats_flush(tc.grp);
  };

  // ---- the stage loop.  `cur` = the stage whose operands are in LDS, `nxt` = the stage in flight / about to be committed.
  //      The first trip has no current stage: it only commits stage 0 and issues stage 1 (one copy of every phase in the
  //      code: the register allocator sees each of them once).
  if (KCP > KC) {  // zero the K padding once: never overwritten by commit
    constexpr int PV = (KCP - KC) / VG;
    for (int i = tid; i < BN * PV; i += 256)
      *reinterpret_cast<vec_t*>(&ws[(i / PV) * WKP + KC + (i % PV) * VG]) = make_uint4(0u, 0u, 0u, 0u);
  }
  int ctile = -1, cchunk = 0;                    // current stage (none yet)
  int ntile_ = t_first, nchunk_ = 0;             // next stage
  Tile tcur = tile_at(t_first), tnxt = tcur;
  int sgrp = tcur.grp;                           // group the statistics registers belong to
  bool first_w = true;
  stats_clear();
  issue(tnxt, 0, true);
  while (true) {
    if (ctile >= 0) {
      if (cchunk == 0) {
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int f = 0; f < NF; ++f) acc[m][f] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      mma();
      fi_lds_barrier();                          // every wave has read the current stage out of LDS (only LDS is ordered)
    }
    const bool done = ctile >= 0 && cchunk == nchunk - 1;
    const Tile tdone = tcur;
    const bool have_next = ntile_ < t_end;
    if (have_next) {
      commit(tnxt, nchunk_ * CK, first_w || !w_resident);       // waits for the loads issued a stage ago
      first_w = false;
      ctile = ntile_;
      cchunk = nchunk_;
      tcur = tnxt;
      nchunk_ = cchunk + 1;
      if (nchunk_ == nchunk) {
        nchunk_ = 0;
        ntile_ = ctile + tstep;
        if (ntile_ < t_end) tnxt = tile_at(ntile_);
      }
      if (ntile_ < t_end) issue(tnxt, nchunk_ * CK, !w_resident);
    }
    if (done) {
      // the finished tile's stores go out BEHIND the loads just issued (see the header comment)
      if (DEFER && tdone.grp != sgrp) {          // its sums start a new statistics group
        stats_flush(sgrp);
        stats_clear();
        sgrp = tdone.grp;
      }
      epilogue(tdone);
    }
    if (!have_next) break;
    fi_lds_barrier();                            // the next stage is in LDS (ds_write only; the loads just issued stay in flight)
  }
  if constexpr (DEFER) stats_flush(sgrp);
}

#undef FI_V2_GEOMETRY

// more than 64 KB of dynamic LDS per workgroup has to be allowed per kernel (once; gfx950 has 160 KB per CU)
static inline bool fi_allow_big_lds(const void* kernel) {
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
}

template <typename T, int KS, int NF, int CK>
static int launch_conv_fwd_v2(const ConvArgs& a, int wgs_per_cu, hipStream_t st) {
  constexpr int HALO = KS / 2, XW = 16 + 2 * HALO, XH = 16 + 2 * HALO, KK = KS * KS;
  constexpr int KSTEP = DT<T>::KSTEP;
  constexpr int KC = KK * CK, KCP = ((KC + KSTEP - 1) / KSTEP) * KSTEP;
  constexpr int CKP = FiLdsStride<T, CK>::value, WKP = FiLdsStride<T, KCP>::value;
  constexpr int BN = NF * 16;
  const size_t lds = (size_t)(XH * XW * CKP + BN * WKP) * sizeof(T) + (size_t)4 * BN * 2 * sizeof(float);
  const long ntile = (long)a.N * a.tilesX * a.tilesY;
  // grid: a multiple of 8 (XCDs) x nct, at most wgs_per_cu per CU, no more than one workgroup per (tile, slab)
  long per_xcd = 32L * wgs_per_cu / a.nct;
  const long tpx = (ntile + 7) / 8;
  if (per_xcd > tpx) per_xcd = tpx;
  if (per_xcd < 1) per_xcd = 1;
  const long blocks = 8 * per_xcd * a.nct;
  const dim3 g((unsigned)blocks), b(256);
  if (a.xf == 0) {
    static const bool big = fi_allow_big_lds((const void*)conv_fwd_v2_kernel<T, KS, NF, CK, 0>);
    (void)big;
    hipLaunchKernelGGL((conv_fwd_v2_kernel<T, KS, NF, CK, 0>), g, b, lds, st, a);
  } else if (a.xf == 1) {
    static const bool big = fi_allow_big_lds((const void*)conv_fwd_v2_kernel<T, KS, NF, CK, 1>);
    (void)big;
    hipLaunchKernelGGL((conv_fwd_v2_kernel<T, KS, NF, CK, 1>), g, b, lds, st, a);
  } else {
    if constexpr (KS == 3 && CK * (int)sizeof(T) <= 32) {        // pooled sources: 4 raw vectors per staged one -> narrow chunks only
      static const bool big = fi_allow_big_lds((const void*)conv_fwd_v2_kernel<T, KS, NF, CK, 2>);
      (void)big;
      hipLaunchKernelGGL((conv_fwd_v2_kernel<T, KS, NF, CK, 2>), g, b, lds, st, a);
    } else {
      return FI_ERR_UNSUPPORTED;
    }
  }
  FI_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// forward / dgrad kernel, fourth form: WAVE-SPECIALISED persistent workgroups (channel-rich layers)
// ---------------------------------------------------------------------------------------------
// Measured on the 64..256-channel layers (ISA counts + tools/kprobe.sh): per 32-channel chunk a wave of the one-tile kernel
// issues ~700 vector instructions of staging (addresses, BatchNorm + LeakyReLU of the producer, ~600 more with dropout) for
// 144 MFMAs, and the two phases alternate behind barriers -- the matrix cores idle through every staging phase of their
// own workgroup and the staging waits out its own load latency: 18-30 % MFMA busy.  Here the two jobs run CONCURRENTLY on
// every SIMD: a workgroup is 8 waves, waves 0-3 ("consumers") only read operand fragments from LDS and issue MFMAs, waves
// 4-7 ("producers") only load, transform and write the NEXT stage into the other half of a double-buffered LDS tile.  One
// barrier per stage.  The producers keep two register sets: the loads of stage s+2 are issued before stage s+1 is
// transformed, so a load has a full stage (~1 us) to land.  Persistent: a workgroup walks a contiguous run of (slab, tile)
// items, so the producers run ahead across tile boundaries while the consumers store a finished tile; BatchNorm statistics
// stay in consumer registers over the run and reach the fp64 accumulators once per (group, slab) with one atomic per lane.
// Same LDS layouts, operand mapping, transforms (XF) and rounding as the other forms; 3x3, 16-bit storage, 16-row tiles.
#ifndef FI_WS_DEBUG
#define FI_WS_DEBUG 0          // kernel A/B builds: 1 no MFMAs, 2 no loads, 4 no LDS commit, 8 no transform, 16 no epilogue
#endif
// PW = 8: 4 consumer waves (4 tile rows each) + one team of 8 producer waves with two register sets (above).
// PW = 44 ("duo"): 8 consumer waves (2 tile rows each) + TWO teams of 4 producer waves that alternate stages, one register
//   set each -- 16 waves, 128 registers.  Two consumer waves per SIMD share its matrix pipe, so one wave's fragment reads and
//   its epilogue (~750 instructions per tile: bias, rounding, stores, statistics) run under the other's MFMAs instead of
//   idling the pipe; a team issues stage s+2 in one iteration and transforms / commits it in the next, so its loads need no
//   wait-count bookkeeping (vmcnt(0) at commit is exact) and the coefficients are fetched at commit.
#ifdef FI_TRACE
// per-wave stage timeline of conv_fwd_ws_kernel (tools/ws_trace.py): [workgroup][12 waves][16 stages][8 stamps], stages 4 .. 19 of the run
#define FI_TWS(slot, stage)                                                                                              \
  do {                                                                                                                   \
    if (a.trace && lane == 0 && (stage) >= 4 && (stage) < 20)                                                            \
      a.trace[((((size_t)blockIdx.x * 12 + wave) * 16) + ((stage) - 4)) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define FI_TWS(slot, stage) do { } while (0)
#endif
template <typename T, int NF, int CK, int XF, int PW>
__global__ __launch_bounds__(PW == 44 ? 1024 : (4 + PW) * 64, 1) void conv_fwd_ws_kernel(ConvArgs a) {
  constexpr bool DUO = PW == 44;
  constexpr int CW = DUO ? 8 : 4;                                // consumer waves
  constexpr int NT = DUO ? 1024 : (4 + PW) * 64, PT = DUO ? 256 : PW * 64;        // threads: all, one producer team
  static_assert(sizeof(T) == 2, "16-bit storage");
  constexpr int KS = 3, TH = 16, HALO = 1, XW = 18, XH = 18, KK = 9;
  constexpr int VG = DT<T>::VG, KSTEP = DT<T>::KSTEP, KV = DT<T>::KV;
  constexpr int CKP = FiLdsStride<T, CK>::value;
  constexpr int KC = KK * CK;
  constexpr int KCP = ((KC + KSTEP - 1) / KSTEP) * KSTEP;
  constexpr int WKP = FiLdsStride<T, KCP>::value;
  constexpr int BN = NF * 16, MF = TH / CW, VPP = CK / VG;
  constexpr int XTILE = XH * XW * CKP, WTILE = BN * WKP, STAGE = XTILE + WTILE;     // elements
  static_assert(CK > VG, "whole-vector channel counts only");
  typedef typename DT<T>::vec_t vec_t;
  typedef typename DT<T>::frag_t frag_t;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef unsigned v2u __attribute__((ext_vector_type(2)));

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const lds = reinterpret_cast<T*>(smem);                    // [2 stages]{ xs[XH*XW][CKP], ws[BN][WKP] }, [consumer waves] strip[3 * BN] floats

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= CW;
  const int cr = a.c0 + a.c1;                                   // channels of one slice of the (concatenated) input
  const int cin = (a.depth > 0 ? 3 : 1) * cr, cout = a.co0 + a.co1;      // contraction channels per tap: depth taps x channels
  const int H = a.H, W = a.W;

  // ---- this workgroup's run of items; item = slab * ntile + tile (slab-major: a run keeps its slab and statistics group)
  const int ntile = a.N * a.tilesY * a.tilesX, tpi = a.tilesY * a.tilesX;
  const int nitem = ntile * a.nct;
  const int per = (nitem + (int)gridDim.x - 1) / (int)gridDim.x;
  const int i_begin = (int)blockIdx.x * per, i_end = min(nitem, i_begin + per);
  if (i_begin >= i_end) return;
  const int nchunk = (cin + CK - 1) / CK;
  const int nstage = (i_end - i_begin) * nchunk;

  constexpr unsigned esz = sizeof(T);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const unsigned hw = (unsigned)H * (unsigned)W;

  struct Item {
    int tx, ty, n, ct;
  };
  auto item_at = [&](int item) {
    Item c;
    c.ct = item / ntile;
    const int tile = item - c.ct * ntile;
    c.n = tile / tpi;
    const int r = tile - c.n * tpi;
    c.ty = r / a.tilesX;
    c.tx = r - c.ty * a.tilesX;
    return c;
  };
  auto item_next = [&](Item c) {
    if (++c.tx == a.tilesX) {
      c.tx = 0;
      if (++c.ty == a.tilesY) {
        c.ty = 0;
        if (++c.n == a.N) {
          c.n = 0;
          ++c.ct;
        }
      }
    }
    return c;
  };

  if (KCP > KC) {  // zero the K padding of both weight tiles once: never overwritten by the producers
    constexpr int PV = (KCP - KC) / VG;
    for (int i = tid; i < 2 * BN * PV; i += NT) {
      const int b = i / (BN * PV), j = i % (BN * PV);
      *reinterpret_cast<vec_t*>(&lds[b * STAGE + XTILE + (j / PV) * WKP + KC + (j % PV) * VG]) = make_uint4(0u, 0u, 0u, 0u);
    }
  }

  if (producer) {
    // =============================================================================================== producers
    // Loads here are branch-free and fixed in number per stage: the compiler's wait-count bookkeeping must be able to tell
    // "the older register set has landed" from "everything has landed" (a conditional or looped load -- e.g. the
    // readfirstlane loop a per-lane choice of buffer resource turns into -- makes every later wait a vmcnt(0), which
    // serialises a stage's load latency behind its transform).  Hence: per-lane 64-bit base pointers for the two sources
    // (plain global loads; lanes outside the image read element 0 of their tensor and are zeroed at commit), one buffer
    // resource for the weights, and the run's tail issues dead stages (`live` = false) instead of skipping them.
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.w), 0, (unsigned)cout * KK * (unsigned)cin * esz, 0x00020000);

    // staging geometry: a producer thread owns one vector column of the halo tile / of the weight slab
    constexpr int XCOLS = XW * VPP, XRPP = PT / XCOLS, XPASS = (XH + XRPP - 1) / XRPP;
    constexpr int WCOLS = KK * VPP, WRPP = PT / WCOLS, WPASS = (BN + WRPP - 1) / WRPP;
    static_assert(XRPP >= 1 && WRPP >= 1, "tile too wide for the producer threads");
    constexpr int NS = XF == 2 ? 4 : 1;
    const int team = DUO ? (wave - CW) >> 2 : 0;
    const int ptid = tid - CW * 64 - team * 256;
    const int xcol = ptid % XCOLS, xrow0 = ptid / XCOLS;
    const int xpx = xcol / VPP, xv = xcol % VPP;
    const int xlds0 = (xrow0 * XW + xpx) * CKP + xv * VG;
    const int wcol = ptid % WCOLS, wrow0 = ptid / WCOLS;
    const int wt = wcol / VPP, wv = wcol % VPP;
    const int wlds0 = XTILE + wrow0 * WKP + wt * CK + wv * VG;

    // a register set = everything stage k needs between its issue and its commit
    struct Set {
      vec_t x[XPASS][NS];
      vec_t w[WPASS];
      float sc[(XF != 0 && !DUO) ? VG : 1], sh[(XF != 0 && !DUO) ? VG : 1];
      unsigned cofs;            // (duo) element offset of this thread's coefficients, fetched at commit
      unsigned flags;           // bit p: pass p is inside the image; bit 16: source 0; bit 17: transform active
      unsigned vix0;            // dropout element-vector index of pass 0
    };
    Set S0, S1;                 // (duo: S0 only)
    const bool drop0 = XF == 1 && a.t0.drop_mode == FI_DROP_RNG_ELEM;
    uint64_t seed_base = 0;
    if (drop0) {
      seed_base = a.t0.seed;
      if (a.t0.seed_offset) seed_base += 0xD1B54A32D192ED03ull * (uint64_t)(uint32_t)a.t0.seed_offset[0];
    }
    const T* const x0p = reinterpret_cast<const T*>(a.x0);
    const T* const x1p = reinterpret_cast<const T*>(a.x1);
    const float* const dummy = reinterpret_cast<const float*>(a.w);        // >= 32 readable bytes for inactive lanes

    // addresses advance by a per-thread constant from pass to pass: one add and one select per load
    const unsigned wstep = (unsigned)(WRPP * KK) * (unsigned)cin * esz;
    auto issue = [&](Set& S, const Item& it, int cb, bool live) __attribute__((always_inline)) {
#if FI_WS_DEBUG & 2
      return;
#endif
      const int ci = cb + xv * VG;
      const bool chok = live && ci < cin && xrow0 < XRPP;
      int cc = ci < cin ? ci : 0;
      int kdi = 1;                                               // depth tap of this thread's channel vector (2D: the centre)
      if (a.depth > 0) {
        kdi = cc >= 2 * cr ? 2 : (cc >= cr ? 1 : 0);
        cc -= kdi * cr;
      }
      const bool first = cc < a.c0;
      const unsigned cs = (unsigned)(first ? a.c0 : a.c1), co = (unsigned)(first ? cc : cc - a.c0);
      const T* const xb = first ? x0p : x1p;
      const int grp = a.gimages > 0 ? it.n / a.gimages : 0;
      const int nl = it.n - grp * a.gimages;
      int ns = (XF != 0 && first && a.bcast0) ? nl : it.n;
      bool sliceok = true;
      if (a.depth > 0) {                                         // slice it.n of its volume + (kdi - 1): zero outside the volume
        const int d = it.n % a.depth + kdi - 1;
        sliceok = d >= 0 && d < a.depth;
        ns = it.n + kdi - 1;
      }
      const int gx = it.tx * 16 + xpx - HALO;
      const int gy0 = it.ty * TH + xrow0 - HALO;
      const bool colok = chok && sliceok && (unsigned)gx < (unsigned)W;
      unsigned flags = first ? 0x10000u : 0u;
      if constexpr (XF == 2) {
        const unsigned o0 = (unsigned)((ns * 2 * H + 2 * gy0) * (2 * W) + 2 * gx) * cs + co;
        const unsigned rowe = (unsigned)(2 * W) * cs, step = (unsigned)(2 * XRPP) * rowe;
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
          const bool ok = colok && (unsigned)(gy0 + p * XRPP) < (unsigned)H;
          flags |= ok ? (1u << p) : 0u;
          const unsigned o = ok ? o0 + (unsigned)p * step : 0u, re = ok ? rowe : 0u, pe = ok ? cs : 0u;
          S.x[p][0] = *reinterpret_cast<const vec_t*>(xb + o);
          S.x[p][1] = *reinterpret_cast<const vec_t*>(xb + o + pe);
          S.x[p][2] = *reinterpret_cast<const vec_t*>(xb + o + re);
          S.x[p][3] = *reinterpret_cast<const vec_t*>(xb + o + re + pe);
        }
      } else {
        const unsigned o0 = (unsigned)((ns * H + gy0) * W + gx) * cs + co;
        const unsigned step = (unsigned)(XRPP * W) * cs;
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
          const bool ok = colok && (unsigned)(gy0 + p * XRPP) < (unsigned)H;
          flags |= ok ? (1u << p) : 0u;
          S.x[p][0] = *reinterpret_cast<const vec_t*>(xb + (ok ? o0 + (unsigned)p * step : 0u));
        }
      }
      {
        const int wci = cb + wv * VG;
        const int gco0 = it.ct * BN + wrow0;
        const bool wok = live && wci < cin && wrow0 < WRPP;
        const int lim = min(BN, cout - it.ct * BN);              // uniform: rows of the slab that exist
        const unsigned o0 = ((unsigned)(gco0 * KK + wt) * (unsigned)cin + (unsigned)wci) * esz;
#pragma unroll
        for (int p = 0; p < WPASS; ++p) {
          const bool ok = wok && wrow0 + p * WRPP < lim;
          S.w[p] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? o0 + (unsigned)p * wstep : OOB, 0, 0));
        }
      }
      if constexpr (XF != 0) {
        const float* scp = first ? a.t0.scale : a.t1.scale;
        const float* shp = first ? a.t0.shift : a.t1.shift;
        const bool act = chok && scp != nullptr;
        flags |= act ? 0x20000u : 0u;
        const unsigned cofs = act ? (unsigned)grp * cs + co : 0u;
        S.cofs = cofs;
        if constexpr (!DUO) {
          scp = act ? scp : dummy;
          shp = act ? shp : dummy;
#pragma unroll
          for (int j = 0; j < VG; j += 4) {
            const float4 s4 = *reinterpret_cast<const float4*>(scp + cofs + j);
            const float4 h4 = *reinterpret_cast<const float4*>(shp + cofs + j);
            S.sc[j] = s4.x, S.sc[j + 1] = s4.y, S.sc[j + 2] = s4.z, S.sc[j + 3] = s4.w;
            S.sh[j] = h4.x, S.sh[j + 1] = h4.y, S.sh[j + 2] = h4.z, S.sh[j + 3] = h4.w;
          }
        }
        S.vix0 = (unsigned)((nl * H + gy0) * W + gx) * (cs / VG) + co / VG;
      }
      S.flags = flags;
    };

    auto commit = [&](Set& S, const Item& it, int buf) __attribute__((always_inline)) {
#if FI_WS_DEBUG & 4
      return;
#endif
      T* const xs = lds + buf * STAGE;
      const bool first = (S.flags & 0x10000u) != 0;
      float slope = 1.f;
      bool xf = false, drop = false;
      uint64_t seed = 0;
      unsigned vstep = 0;
      float csc[VG], csh[VG];                                   // this thread's 8 scale / shift values
      if constexpr (XF != 0) {
        slope = first ? a.t0.slope : a.t1.slope;
        xf = (S.flags & 0x20000u) != 0;
        if constexpr (DUO) {
          const float* const scp = xf ? (first ? a.t0.scale : a.t1.scale) : dummy;
          const float* const shp = xf ? (first ? a.t0.shift : a.t1.shift) : dummy;
#pragma unroll
          for (int j = 0; j < VG; j += 4) {
            const float4 s4 = *reinterpret_cast<const float4*>(scp + S.cofs + j);
            const float4 h4 = *reinterpret_cast<const float4*>(shp + S.cofs + j);
            csc[j] = s4.x, csc[j + 1] = s4.y, csc[j + 2] = s4.z, csc[j + 3] = s4.w;
            csh[j] = h4.x, csh[j + 1] = h4.y, csh[j + 2] = h4.z, csh[j + 3] = h4.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < VG; ++j) csc[j] = S.sc[j], csh[j] = S.sh[j];
        }
#if FI_WS_DEBUG & 8
        xf = false;
#endif
        if constexpr (XF == 1) {
          drop = drop0 && first;
          const int grp = a.gimages > 0 ? it.n / a.gimages : 0;
          seed = seed_base + (uint64_t)grp * a.t0.seed_gstride;
          vstep = (unsigned)(XRPP * W) * ((unsigned)(first ? a.c0 : a.c1) / VG);
        }
      }
      auto xform = [&](const vec_t& raw, size_t vecidx) __attribute__((always_inline)) -> vec_t {
        float f[VG];
        VecWords<T>::unpack(raw, f);
#pragma unroll
        for (int j = 0; j < VG; ++j) {
          const float v = f[j] * csc[j] + csh[j];
          f[j] = fmaxf(v, v * slope);                  // = v > 0 ? v : v * slope for 0 <= slope <= 1 (host-checked)
        }
        if (drop) {
#pragma unroll
          for (int g4 = 0; g4 < VG / 4; ++g4) {
            uint32_t rr[4];
            fi_rand32x4(seed, vecidx * (VG / 4) + g4, rr);
#pragma unroll
            for (int j = 0; j < 4; ++j) f[g4 * 4 + j] *= rr[j] >= a.t0.thresh ? a.t0.keep_scale : 0.f;
          }
        }
        return VecWords<T>::pack(f);
      };
#pragma unroll
      for (int p = 0; p < XPASS; ++p) {
        const int py = xrow0 + p * XRPP;
        if (xrow0 < XRPP && py < XH) {
          const bool ok = (S.flags >> p) & 1u;                  // outside the image (or beyond Cin): z = 0, not act(shift)
          vec_t val;
          if constexpr (XF == 0) {
            val = fi_vec_select(ok, S.x[p][0]);
          } else if constexpr (XF == 2) {
            float best[VG], cand[VG];
            VecWords<T>::unpack(xf ? xform(S.x[p][0], 0) : S.x[p][0], best);
#pragma unroll
            for (int k = 1; k < 4; ++k) {
              VecWords<T>::unpack(xf ? xform(S.x[p][k], 0) : S.x[p][k], cand);
#pragma unroll
              for (int j = 0; j < VG; ++j) best[j] = cand[j] > best[j] ? cand[j] : best[j];
            }
            val = fi_vec_select(ok, VecWords<T>::pack(best));
          } else {
            val = fi_vec_select(ok, xf ? xform(S.x[p][0], S.vix0 + (unsigned)p * vstep) : S.x[p][0]);
          }
          *reinterpret_cast<vec_t*>(&xs[xlds0 + p * (XRPP * XW * CKP)]) = val;
        }
      }
#pragma unroll
      for (int p = 0; p < WPASS; ++p) {
        const int col = wrow0 + p * WRPP;
        if (wrow0 < WRPP && col < BN) *reinterpret_cast<vec_t*>(&xs[wlds0 + p * (WRPP * WKP)]) = S.w[p];
      }
    };

    if constexpr (!DUO) {
      // cursors: `it_i` = the stage to issue next, `it_c` = the stage to commit next (one behind)
      Item it_i = item_at(i_begin), it_c = it_i;
      int ch_i = 0, ch_c = 0, k_i = 0;
      auto adv_i = [&]() __attribute__((always_inline)) {
        ++k_i;
        if (++ch_i == nchunk) {
          ch_i = 0;
          it_i = item_next(it_i);
        }
      };
      auto adv_c = [&]() __attribute__((always_inline)) {
        if (++ch_c == nchunk) {
          ch_c = 0;
          it_c = item_next(it_c);
        }
      };
      // prologue: stage 0 into buffer 0, stage 1 in flight
      issue(S0, it_i, ch_i * CK, true);
      adv_i();
      issue(S1, it_i, ch_i * CK, k_i < nstage);
      adv_i();
      commit(S0, it_c, 0);
      adv_c();
      fi_lds_barrier();
      for (int s = 0; s < nstage; s += 2) {
        // stage s is being consumed from buffer 0: issue s+2 into set 0, commit s+1 (set 1) into buffer 1
        FI_TWS(0, s);
        issue(S0, it_i, ch_i * CK, k_i < nstage);
        adv_i();
        FI_TWS(1, s);
        if (s + 1 < nstage) {
          commit(S1, it_c, 1);
          adv_c();
        }
        FI_TWS(2, s);
        fi_lds_barrier();
        FI_TWS(3, s);
        if (s + 1 >= nstage) break;
        // stage s+1 is being consumed from buffer 1: issue s+3 into set 1, commit s+2 (set 0) into buffer 0
        FI_TWS(0, s + 1);
        issue(S1, it_i, ch_i * CK, k_i < nstage);
        adv_i();
        FI_TWS(1, s + 1);
        if (s + 2 < nstage) {
          commit(S0, it_c, 0);
          adv_c();
        }
        FI_TWS(2, s + 1);
        fi_lds_barrier();
        FI_TWS(3, s + 1);
      }
    } else {
      // team g owns the stages of parity g: while stage s is consumed, team (s & 1) issues stage s+2 and the other team
      // commits stage s+1 into buffer (s+1) & 1
      Item it = item_at(i_begin);
      int ch = 0, k = 0;                                         // this team's current stage
      auto adv = [&]() __attribute__((always_inline)) {
        ++k;
        if (++ch == nchunk) {
          ch = 0;
          it = item_next(it);
        }
      };
      if (team == 0) {
        issue(S0, it, 0, true);
        commit(S0, it, 0);                                       // stage 0
        adv();
        adv();
      } else {
        adv();
        issue(S0, it, ch * CK, k < nstage);                      // stage 1 in flight
      }
      fi_lds_barrier();
      for (int s = 0; s < nstage; ++s) {
        if ((s & 1) == team) {
          issue(S0, it, ch * CK, k < nstage);                    // k == s + 2
        } else {
          if (s + 1 < nstage) commit(S0, it, (s + 1) & 1);       // k == s + 1
          adv();
          adv();
        }
        fi_lds_barrier();
      }
    }
  } else {
    // =============================================================================================== consumers
    const int li = lane & 15, kg = lane >> 4;
    const __amdgpu_buffer_rsrc_t ry0 = __builtin_amdgcn_make_buffer_rsrc(
        a.y0, 0, a.y0 ? (unsigned)a.N * hw * (unsigned)a.co0 * esz : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry1 = __builtin_amdgcn_make_buffer_rsrc(
        a.y1, 0, (a.y1 && a.co1) ? (unsigned)a.N * hw * (unsigned)a.co1 * esz : 0u, 0x00020000);

    f32x4 acc[MF][NF];
    // One consumer wave per SIMD: nothing else hides its LDS latency, so the fragment reads are software-pipelined by hand
    // and pinned with scheduling barriers -- the reads of contraction step k+1 are issued before the 4 x NF MFMAs of step k.
    auto mma = [&](int buf) __attribute__((always_inline)) {
      const T* const xs = lds + buf * STAGE;
      const T* const ws = xs + XTILE;
      constexpr int NKS = KCP / KSTEP;                           // contraction steps: taps (CK = 32) or tap pairs (CK = 16)
      frag_t A[2][MF], B[2][NF];
      auto fetch = [&](int ks, int q) __attribute__((always_inline)) {
        const int k0 = ks * KSTEP + kg * KV;
        int t = k0 / CK;
        const int cil = k0 % CK;
        if (t > KK - 1) t = KK - 1;                              // padded K: the weight fragment is zero there
        const int r = t / KS, sx = t % KS;
#pragma unroll
        for (int f = 0; f < NF; ++f) B[q][f] = *reinterpret_cast<const frag_t*>(&ws[(f * 16 + li) * WKP + k0]);
#pragma unroll
        for (int m = 0; m < MF; ++m)
          A[q][m] = *reinterpret_cast<const frag_t*>(&xs[((wave * MF + m + r) * XW + li + sx) * CKP + cil]);
      };
      fetch(0, 0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 1 < NKS) fetch(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int f = 0; f < NF; ++f) acc[m][f] = mfma16(B[ks & 1][f], A[ks & 1][m], acc[m][f]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // BatchNorm statistics and the bias live in a wave-private strip of LDS ([sum BN][sum of squares BN][bias BN] floats):
    // the consumer's registers are taken by the accumulators and two sets of operand fragments.  Per tile the wave folds
    // its 4 x NF fragments over rows and lanes (DPP) and adds the BN pairs to the strip; the strip reaches the fp64
    // accumulators once per (group, slab) of the run.  A wave's own LDS accesses execute in order: no barrier involved.
    float* const strip = reinterpret_cast<float*>(lds + 2 * STAGE) + wave * (3 * BN);
    auto stats_clear = [&]() __attribute__((always_inline)) {
      if (lane < BN) strip[lane] = strip[BN + lane] = 0.f;
    };
    auto stats_flush = [&](int grp, int ct) __attribute__((always_inline)) {
      if (!a.stats) return;
      const int slot = (blockIdx.x * CW + wave) & (FI_STATS_SLOTS - 1);
      const int co = ct * BN + lane;
      if (lane < BN && co < cout) {
        double* const dst = &a.stats[(size_t)grp * a.stats_gstride + ((size_t)slot * cout + co) * 2];
        atomicAdd(dst, (double)strip[lane]);
        atomicAdd(dst + 1, (double)strip[BN + lane]);
      }
    };
    auto load_bias = [&](int ct) __attribute__((always_inline)) {
      const int co = ct * BN + lane;
      if (lane < BN) strip[2 * BN + lane] = (a.bias && co < cout) ? a.bias[co] : 0.f;
    };
    // Stores: global stores are issue-bound (~7 B/clk/CU as 8-byte stores): two tile rows swap halves across the 16-lane
    // rows of the wave (v_permlane16_swap) so that a lane holds 8 consecutive channels of ONE pixel -- half as many, 16-byte
    // stores.  A statistics-only launch (no destination) issues none.
    auto epilogue = [&](const Item& it) __attribute__((always_inline)) {
      const int gx = it.tx * 16 + li;
      const bool colok = gx < W;
      float ps[NF][4], pq[NF][4];
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) ps[f][r] = pq[f][r] = 0.f;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const float4 bv = *reinterpret_cast<const float4*>(&strip[2 * BN + f * 16 + kg * 4]);
        const float bvr[4] = {bv.x, bv.y, bv.z, bv.w};
        const int cg = it.ct * BN + f * 16 + (kg >> 1) * 8;      // the 8 channels this lane stores (whole groups per destination: host)
        const bool second = cg >= a.co0;
#pragma unroll
        for (int mp = 0; mp < MF; mp += 2) {
          v2u q[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int m = mp + h;
            const bool okm = colok && it.ty * TH + wave * MF + m < H;
            const float mk = okm ? 1.f : 0.f;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[m][f][r] + bvr[r];
            q[h] = __builtin_bit_cast(v2u, Quad<T>::pack(v));    // v := the values as stored
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              ps[f][r] += v[r] * mk;                             // tile overhang does not count
              pq[f][r] += (v[r] * mk) * v[r];
            }
          }
          if (a.y0) {
            const v2u lo = __builtin_amdgcn_permlane16_swap(q[0].x, q[1].x, false, false);
            const v2u hi = __builtin_amdgcn_permlane16_swap(q[0].y, q[1].y, false, false);
            const v4u out = {lo.x, hi.x, lo.y, hi.y};            // channels cg .. cg+7 of pixel row mp + (kg & 1)
            const int gy = it.ty * TH + wave * MF + mp + (kg & 1);
            const bool live = colok && gy < H && cg < cout;
            const unsigned pix = (unsigned)((it.n * H + gy) * W + gx);
            if (a.co1 == 0) {
              __builtin_amdgcn_raw_buffer_store_b128(out, ry0, live ? (pix * (unsigned)a.co0 + (unsigned)cg) * esz : OOB, 0, 0);
            } else {
              __builtin_amdgcn_raw_buffer_store_b128(out, ry0, (live && !second) ? (pix * (unsigned)a.co0 + (unsigned)cg) * esz : OOB, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(out, ry1, (live && second) ? (pix * (unsigned)a.co1 + (unsigned)(cg - a.co0)) * esz : OOB, 0, 0);
            }
          }
        }
      }
      if (a.stats) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          float4 sv, qv;
          sv.x = fi_row16_sum(ps[f][0]), sv.y = fi_row16_sum(ps[f][1]), sv.z = fi_row16_sum(ps[f][2]), sv.w = fi_row16_sum(ps[f][3]);
          qv.x = fi_row16_sum(pq[f][0]), qv.y = fi_row16_sum(pq[f][1]), qv.z = fi_row16_sum(pq[f][2]), qv.w = fi_row16_sum(pq[f][3]);
          if (li == 0) {
            float4* const sp = reinterpret_cast<float4*>(&strip[f * 16 + kg * 4]);
            float4* const qp = reinterpret_cast<float4*>(&strip[BN + f * 16 + kg * 4]);
            const float4 s0 = *sp, q0 = *qp;
            *sp = make_float4(s0.x + sv.x, s0.y + sv.y, s0.z + sv.z, s0.w + sv.w);
            *qp = make_float4(q0.x + qv.x, q0.y + qv.y, q0.z + qv.z, q0.w + qv.w);
          }
        }
      }
    };

    Item it = item_at(i_begin);
    int ch = 0;
    int sgrp = a.gimages > 0 ? it.n / a.gimages : 0, sct = it.ct;
    stats_clear();
    load_bias(sct);
    int tstage = 0;
    auto consume = [&](int buf) __attribute__((always_inline)) {
      FI_TWS(0, tstage);
      if (ch == 0) {
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int f = 0; f < NF; ++f) acc[m][f] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#if !(FI_WS_DEBUG & 1)
      mma(buf);
#endif
      FI_TWS(1, tstage);
      if (++ch == nchunk) {
        ch = 0;
        const int grp = a.gimages > 0 ? it.n / a.gimages : 0;
        if (grp != sgrp || it.ct != sct) {                      // the finished tile starts a new (group, slab)
          stats_flush(sgrp, sct);
          stats_clear();
          if (it.ct != sct) load_bias(it.ct);
          sgrp = grp;
          sct = it.ct;
        }
#if !(FI_WS_DEBUG & 16)
        epilogue(it);
#endif
        it = item_next(it);
      }
    };
    fi_lds_barrier();                                            // stage 0 is in buffer 0
    for (int s = 0; s < nstage; s += 2) {
      consume(0);
      FI_TWS(2, tstage);
      fi_lds_barrier();
      FI_TWS(3, tstage);
      ++tstage;
      if (s + 1 >= nstage) break;
      consume(1);
      FI_TWS(2, tstage);
      fi_lds_barrier();
      FI_TWS(3, tstage);
      ++tstage;
    }
    stats_flush(sgrp, sct);
  }
}

template <typename T, int NF, int CK, int PW>
static int launch_conv_fwd_ws(const ConvArgs& a, int wgs_per_cu, hipStream_t st) {
  constexpr int XW = 18, XH = 18, KK = 9;
  constexpr int KSTEP = DT<T>::KSTEP;
  constexpr int KC = KK * CK, KCP = ((KC + KSTEP - 1) / KSTEP) * KSTEP;
  constexpr int CKP = FiLdsStride<T, CK>::value, WKP = FiLdsStride<T, KCP>::value;
  constexpr int BN = NF * 16;
  constexpr int CW = PW == 44 ? 8 : 4, NT = PW == 44 ? 1024 : (4 + PW) * 64;
  const size_t lds = (size_t)2 * (XH * XW * CKP + BN * WKP) * sizeof(T) + (size_t)CW * 3 * BN * sizeof(float);
  const long nitem = (long)a.N * a.tilesX * a.tilesY * a.nct;
  long blocks = 256L * (wgs_per_cu > 0 ? wgs_per_cu : 1);
  if (blocks > nitem) blocks = nitem;
  const dim3 g((unsigned)blocks), b(NT);
  if (a.xf == 0) {
    static const bool big = fi_allow_big_lds((const void*)conv_fwd_ws_kernel<T, NF, CK, 0, PW>);
    (void)big;
    hipLaunchKernelGGL((conv_fwd_ws_kernel<T, NF, CK, 0, PW>), g, b, lds, st, a);
  } else if (a.xf == 1) {
    static const bool big = fi_allow_big_lds((const void*)conv_fwd_ws_kernel<T, NF, CK, 1, PW>);
    (void)big;
    hipLaunchKernelGGL((conv_fwd_ws_kernel<T, NF, CK, 1, PW>), g, b, lds, st, a);
  } else {
    if constexpr (CK * (int)sizeof(T) <= 32) {                   // pooled sources: 4 raw vectors per staged one -> narrow chunks
      static const bool big = fi_allow_big_lds((const void*)conv_fwd_ws_kernel<T, NF, CK, 2, PW>);
      (void)big;
      hipLaunchKernelGGL((conv_fwd_ws_kernel<T, NF, CK, 2, PW>), g, b, lds, st, a);
    } else {
      return FI_ERR_UNSUPPORTED;
    }
  }
  FI_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// forward kernel, third form: the THIN layers (Cin <= 32, Cout <= 32: full and half resolution of the U-Net)
// ---------------------------------------------------------------------------------------------
// Measured (tools/kprobe.sh, ISA counts): at 16 or 32 channels the one-tile kernel is VALU-ISSUE bound, not memory bound --
// ~600 vector instructions per thread and tile for 20 MFMAs (~1100 with the transforming loader): clamped 64-bit
// addresses and zero-selects of the staging, the weight slab through LDS, AGPR copies, and a cross-lane statistics
// reduction per tile; 48 tiles per CU x 4 waves x ~700 instructions x 4 cycles IS the 60 us the 512^2 16->16 layer takes.
// This form strips the per-tile instruction stream down:
//   * the whole filter (9 taps x <= 32 channels) lives in REGISTERS as MFMA operand fragments for the life of the
//     workgroup: no weight staging, no weight fragment reads;
//   * staging through raw buffer loads: a 32-bit byte offset per vector, out-of-image vectors take an out-of-range offset and
//     the hardware returns zeros -- no clamps, no 64-bit address arithmetic, no zero-selects (plain loader);
//   * stores through raw buffer stores the same way (tile overhang = out-of-range offset, dropped by the hardware);
//   * BatchNorm statistics accumulate in registers over ALL tiles of the workgroup (a contiguous run of one statistics
//     group) and are reduced across lanes / waves and added to the fp64 accumulators once;
//   * persistent: the loads of tile t+1 are in flight during the MFMAs and the epilogue of tile t (conv_fwd_v2_kernel's
//     ordering: the epilogue's stores are issued behind the next loads).
// Same LDS tile layout, operand mapping, input transforms (XF) and rounding as the other two forms.
// Register budget: a 168-register cap (3 waves per SIMD) made the one-fragment instantiations 10-15 % faster than letting
// hipcc take 158-190 registers unasked (it schedules for the occupancy it is promised); the two-fragment ones spill under
// any cap worth having and are left alone (profiles/r02_h_thin_variants.txt).
// TWO: the input is a concatenation of two tensors (c1 != 0) -- a compile-time split, so that the single-source loader is
// straight-line buffer loads (a run-time branch per load measured 1.3-1.7x slower on the 16-channel layers).
// F32N: the logits convolution (out_conv: 16 -> n_class <= 4, /root/reference/code/networks/unet.py:228) -- fp32 outputs, 4 * Cout
// bytes per pixel, stored by the one k-group of lanes that holds real output channels (the generic one-tile kernel took 72 us for
// 12 x 512^2 x 16 -> 3 against 47 of this form's 16 -> 16).
template <typename T, int NF, int CK, int XF, bool TWO, bool F32N = false>
__global__ __launch_bounds__(256, (NF == 1 ? 3 : 1)) void conv_thin_kernel(ConvArgs a) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  constexpr int KS = 3, TH = 16, HALO = 1, XW = 18, XH = 18, KK = 9;
  constexpr int VG = 8, KSTEP = 32, KV = 8;
  constexpr int CKP = FiLdsStride<T, CK>::value;
  constexpr int KC = KK * CK, NKS = (KC + KSTEP - 1) / KSTEP;
  constexpr int BN = NF * 16, MF = 4, VPP = CK / VG;
  typedef typename DT<T>::vec_t vec_t;
  typedef typename DT<T>::frag_t frag_t;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef unsigned v2u __attribute__((ext_vector_type(2)));

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* xs = reinterpret_cast<T*>(smem);                           // [XH*XW][CKP]
  float* red = reinterpret_cast<float*>(xs + XH * XW * CKP);    // [4 waves][BN][2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kg = lane >> 4;
  const int cin = a.c0 + a.c1, cout = a.co0;
  const int H = a.H, W = a.W;

  // ---- this workgroup's tiles: one contiguous run
  const int ntile = a.N * a.tilesY * a.tilesX;
  const int per = (ntile + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per, t_end = min(ntile, t_begin + per);
  if (t_begin >= t_end) return;

  // ---- the filter as MFMA "A" fragments: wf[f][ks] = w[co = f*16 + li][k = ks*32 + kg*8 .. +8), k = tap*CK + channel
  frag_t wf[NF][NKS];
  {
    const T* wg = reinterpret_cast<const T*>(a.w);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int k0 = ks * KSTEP + kg * KV, t = k0 / CK, c = k0 % CK, co = f * 16 + li;
        vec_t v = make_uint4(0u, 0u, 0u, 0u);
        if (k0 < KC && c < cin && co < cout) v = *reinterpret_cast<const vec_t*>(wg + ((size_t)co * KK + t) * cin + c);
        wf[f][ks] = __builtin_bit_cast(frag_t, v);
      }
  }
  float bv[NF][4];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = f * 16 + kg * 4 + r;
      bv[f][r] = (a.bias && co < cout) ? a.bias[co] : 0.f;
    }

  // ---- buffer resources (raw, range-checked: an offset beyond num_records reads zeros / drops the store)
  const unsigned esz = sizeof(T);
  const unsigned src_images = (XF != 0 && a.bcast0) ? (unsigned)a.gimages : (unsigned)a.N;
  const unsigned hw_src = (XF == 2 ? 4u : 1u) * (unsigned)H * (unsigned)W;
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x0), 0,
                                                                      src_images * hw_src * (unsigned)a.c0 * esz, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.x1), 0, a.c1 ? (unsigned)a.N * (unsigned)H * (unsigned)W * (unsigned)a.c1 * esz : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      a.y0, 0, a.y0 ? (unsigned)a.N * (unsigned)H * (unsigned)W * (unsigned)cout * (F32N ? 4u : esz) : 0u, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;

  // ---- staging geometry: a thread owns one vector column (pixel column xpx, channel vector xv) of the halo tile
  constexpr int XCOLS = XW * VPP, XRPP = 256 / XCOLS, XPASS = (XH + XRPP - 1) / XRPP;
  constexpr int NS = XF == 2 ? 4 : 1;
  const int xcol = tid % XCOLS, xrow0 = tid / XCOLS;
  const int xpx = xcol / VPP, xv = xcol % VPP;
  const int xlds0 = (xrow0 * XW + xpx) * CKP + xv * VG;
  const int xc = xv * VG;                                       // first channel of this thread's vector
  const bool xfirst = !TWO || xc < a.c0;                        // which source it comes from
  const bool xchan = xc < cin && xrow0 < XRPP;
  const unsigned xcs = (unsigned)(xfirst ? a.c0 : a.c1), xco = (unsigned)(xfirst ? xc : xc - a.c0);

  // tile coordinates are wave-uniform and advance by one tile at a time: no divisions in the loop
  struct Tile {
    int tx, ty, n;
  };
  auto tile_at = [&](int tile) {
    Tile c;
    c.tx = tile % a.tilesX;
    c.ty = (tile / a.tilesX) % a.tilesY;
    c.n = tile / (a.tilesX * a.tilesY);
    return c;
  };
  auto tile_next = [&](Tile c) {
    if (++c.tx == a.tilesX) {
      c.tx = 0;
      if (++c.ty == a.tilesY) {
        c.ty = 0;
        ++c.n;
      }
    }
    return c;
  };
  vec_t xr[XPASS][NS];
  // offsets advance by a per-thread constant from pass to pass (one add, one compare, one select per load); the image
  // tests of the passes are kept as bits for the commit
  const unsigned xstep = (unsigned)(XRPP * (XF == 2 ? 4 : 1)) * (unsigned)W * xcs * esz;
  unsigned okbits = 0;
  auto issue = [&](const Tile& tc, int grp) __attribute__((always_inline)) {
    const int tx = tc.tx, ty = tc.ty, n = tc.n;
    const int ns = (XF != 0 && xfirst && a.bcast0) ? n - grp * a.gimages : n;
    const int gx = tx * 16 + xpx - HALO;
    const int gy0 = ty * TH + xrow0 - HALO;
    const bool colok = xchan && (unsigned)gx < (unsigned)W;
    unsigned bits = 0;
    if constexpr (XF == 2) {
      const unsigned rowb = (unsigned)(2 * W) * xcs * esz, pxb = xcs * esz;
#pragma unroll
      for (int p = 0; p < XPASS; ++p) {
        const int gy = gy0 + p * XRPP;
        const bool ok = colok && (unsigned)gy < (unsigned)H;
        bits |= ok ? (1u << p) : 0u;
        // (offsets recomputed per pass: carried in registers they cost the two-fragment instantiations their occupancy)
        const unsigned o = ((unsigned)((ns * 2 * H + 2 * gy) * (2 * W) + 2 * gx) * xcs + xco) * esz;
        xr[p][0] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(r0, ok ? o : OOB, 0, 0));
        xr[p][1] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(r0, ok ? o + pxb : OOB, 0, 0));
        xr[p][2] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(r0, ok ? o + rowb : OOB, 0, 0));
        xr[p][3] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(r0, ok ? o + rowb + pxb : OOB, 0, 0));
      }
    } else {
      const unsigned o0 = ((unsigned)((ns * H + gy0) * W + gx) * xcs + xco) * esz;
      // one resource per load instruction (a per-lane choice of resource is lowered to a readfirstlane loop): two sources
      // go through per-lane base pointers and plain global loads; their out-of-image lanes read element 0 and are
      // zeroed at commit
      const char* const xb = reinterpret_cast<const char*>(xfirst ? a.x0 : a.x1);
#pragma unroll
      for (int p = 0; p < XPASS; ++p) {
        const bool ok = colok && (unsigned)(gy0 + p * XRPP) < (unsigned)H;
        bits |= ok ? (1u << p) : 0u;
        const unsigned o = o0 + (unsigned)p * xstep;
        if constexpr (!TWO) {
          xr[p][0] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(r0, ok ? o : OOB, 0, 0));
        } else {
          xr[p][0] = *reinterpret_cast<const vec_t*>(xb + (ok ? o : 0u));
        }
      }
    }
    okbits = bits;
  };

  // transform state of the current statistics group (XF != 0): this thread's 8 scale / shift values, its dropout seed
  float sc[VG], sh[VG];
  float slope = 1.f;
  bool xf = false, drop = false;
  uint64_t seed = 0;
  auto load_coefs = [&](int grp) __attribute__((always_inline)) {
    if constexpr (XF != 0) {
      const float* scp = xfirst ? a.t0.scale : a.t1.scale;
      const float* shp = xfirst ? a.t0.shift : a.t1.shift;
      slope = xfirst ? a.t0.slope : a.t1.slope;
      xf = scp != nullptr && xc < cin;
#pragma unroll
      for (int j = 0; j < VG; ++j) {
        sc[j] = xf ? scp[(unsigned)grp * xcs + xco + j] : 1.f;
        sh[j] = xf ? shp[(unsigned)grp * xcs + xco + j] : 0.f;
      }
      drop = XF == 1 && xfirst && a.t0.drop_mode == FI_DROP_RNG_ELEM;
      if (drop) {
        seed = a.t0.seed + (uint64_t)grp * a.t0.seed_gstride;
        if (a.t0.seed_offset) seed += 0xD1B54A32D192ED03ull * (uint64_t)(uint32_t)a.t0.seed_offset[0];
      }
    }
  };
  auto xform = [&](const vec_t& raw, size_t vecidx) __attribute__((always_inline)) -> vec_t {
    float f[VG];
    VecWords<T>::unpack(raw, f);
#pragma unroll
    for (int j = 0; j < VG; ++j) {
      const float v = f[j] * sc[j] + sh[j];
      f[j] = fmaxf(v, v * slope);
    }
    if (drop) {
#pragma unroll
      for (int g4 = 0; g4 < VG / 4; ++g4) {
        uint32_t rr[4];
        fi_rand32x4(seed, vecidx * (VG / 4) + g4, rr);
#pragma unroll
        for (int j = 0; j < 4; ++j) f[g4 * 4 + j] *= rr[j] >= a.t0.thresh ? a.t0.keep_scale : 0.f;
      }
    }
    return VecWords<T>::pack(f);
  };

  auto commit = [&](const Tile& tc, int grp) __attribute__((always_inline)) {
    // dropout stream position of pass 0 (element-vector index inside the group's tensor), advancing by a constant per pass
    unsigned vix0 = 0, vstep = 0;
    if constexpr (XF == 1) {
      const int nl = tc.n - grp * a.gimages;
      vix0 = (unsigned)((nl * H + tc.ty * TH + xrow0 - HALO) * W + tc.tx * 16 + xpx - HALO) * (xcs / VG) + xco / VG;
      vstep = (unsigned)(XRPP * W) * (xcs / VG);
    }
#pragma unroll
    for (int p = 0; p < XPASS; ++p) {
      const int py = xrow0 + p * XRPP;
      if (xrow0 < XRPP && py < XH) {
        const bool ok = (okbits >> p) & 1u;                     // inside the image (and a real channel vector)
        vec_t val;
        if constexpr (XF == 0) {
          // one source: out-of-image vectors arrived as zeros (range-checked buffer loads)
          val = !TWO ? xr[p][0] : fi_vec_select(ok, xr[p][0]);
        } else if constexpr (XF == 2) {                         // z of the padding is 0, not act(shift)
          float best[VG], cand[VG];
          VecWords<T>::unpack(xf ? xform(xr[p][0], 0) : xr[p][0], best);
#pragma unroll
          for (int k = 1; k < 4; ++k) {
            VecWords<T>::unpack(xf ? xform(xr[p][k], 0) : xr[p][k], cand);
#pragma unroll
            for (int j = 0; j < VG; ++j) best[j] = cand[j] > best[j] ? cand[j] : best[j];
          }
          val = fi_vec_select(ok, VecWords<T>::pack(best));
        } else {
          val = fi_vec_select(ok, xf ? xform(xr[p][0], vix0 + (unsigned)p * vstep) : xr[p][0]);
        }
        *reinterpret_cast<vec_t*>(&xs[xlds0 + p * (XRPP * XW * CKP)]) = val;
      }
    }
  };

  f32x4 acc[MF][NF];
  auto mma = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[m][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      // this lane's 8 contraction elements: tap t, channels cil .. cil+8 (CK % 8 == 0 keeps them inside one tap)
      const int k0 = ks * KSTEP + kg * KV;
      int t = k0 / CK;
      const int cil = k0 % CK;
      if (t > KK - 1) t = KK - 1;                               // padded K: the filter fragment is zero there
      const int r = t / KS, s = t % KS;
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        const frag_t av = *reinterpret_cast<const frag_t*>(&xs[((wave * MF + m + r) * XW + li + s) * CKP + cil]);
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[m][f] = mfma16(wf[f][ks], av, acc[m][f]);
      }
#ifdef FI_THIN_SB
      if ((ks + 1) % FI_THIN_SB == 0) __builtin_amdgcn_sched_barrier(0);   // bound how many fragment reads are hoisted
#endif
    }
  };

  // BatchNorm statistics of the stored values: per-lane partial sums over all tiles of the current group
  float ssum[NF][4], ssq[NF][4];
  auto stats_clear = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) ssum[f][r] = ssq[f][r] = 0.f;
  };
  auto stats_flush = [&](int grp) __attribute__((always_inline)) {
    if (!a.stats) return;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = fi_row16_sum(ssum[f][r]), q = fi_row16_sum(ssq[f][r]);
        if (li == 0) {
          red[(wave * BN + f * 16 + kg * 4 + r) * 2 + 0] = s;
          red[(wave * BN + f * 16 + kg * 4 + r) * 2 + 1] = q;
        }
      }
    fi_lds_barrier();
    if (tid < BN * 2) {
      const int c = tid >> 1, which = tid & 1;
      if (c < cout) {
        double tot = 0.0;
#pragma unroll
        for (int wv_ = 0; wv_ < 4; ++wv_) tot += (double)red[(wv_ * BN + c) * 2 + which];
        const int slot = blockIdx.x & (FI_STATS_SLOTS - 1);
        atomicAdd(&a.stats[(size_t)grp * a.stats_gstride + ((size_t)slot * cout + c) * 2 + which], tot);
      }
    }
    fi_lds_barrier();                                            // `red` may be rewritten by the next flush
  };

  // Stores are issue-bound as 8-byte stores (~7 B/clk/CU): two tile rows swap halves across the wave's 16-lane rows
  // (v_permlane16_swap), a lane then holds 8 consecutive channels of ONE pixel -- half as many, 16-byte stores (Cout % 8 == 0: host).
  // (FULL: the tile lies inside the image -- the common case; no overhang mask on the statistics)
  auto epilogue_t = [&](const Tile& tc, auto full_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int tx = tc.tx, ty = tc.ty, n = tc.n;
    const int gx = tx * 16 + li;
    const bool colok = gx < W;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int mp = 0; mp < MF; mp += 2) {
        v2u q[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = mp + h;
          const float mk = (FULL || (colok && ty * TH + wave * MF + m < H)) ? 1.f : 0.f;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[m][f][r] + bv[f][r];
          q[h] = __builtin_bit_cast(v2u, Quad<T>::pack(v));      // v := the values as stored
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float vm = FULL ? v[r] : v[r] * mk;             // tile overhang does not count
            ssum[f][r] += vm;
            ssq[f][r] += vm * v[r];
          }
        }
        const v2u lo = __builtin_amdgcn_permlane16_swap(q[0].x, q[1].x, false, false);
        const v2u hi = __builtin_amdgcn_permlane16_swap(q[0].y, q[1].y, false, false);
        const v4u out = {lo.x, hi.x, lo.y, hi.y};                // channels cg .. cg+7 of pixel row mp + (kg & 1)
        const int gy = ty * TH + wave * MF + mp + (kg & 1);
        const int cg = f * 16 + (kg >> 1) * 8;
        const unsigned o = ((unsigned)((n * H + gy) * W + gx) * (unsigned)cout + (unsigned)cg) * esz;
        __builtin_amdgcn_raw_buffer_store_b128(out, ry, (colok && gy < H && cg < cout) ? o : OOB, 0, 0);
      }
    }
  };
  // F32N: lane (li, kg == 0) holds output channels 0 .. 3 of pixel column li for each of its wave's four rows
  auto epilogue_n = [&](const Tile& tc) __attribute__((always_inline)) {
    const int gx = tc.tx * 16 + li;
    const bool colok = gx < W && kg == 0;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int gy = tc.ty * TH + wave * MF + m;
      const bool ok = colok && gy < H;
      const unsigned o = (unsigned)((tc.n * H + gy) * W + gx) * (unsigned)cout * 4u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[m][0][r] + bv[0][r];
        const float vm = ok ? v : 0.f;
        ssum[0][r] += vm;
        ssq[0][r] += vm * v;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (ok && r < cout) ? o + 4u * r : OOB, 0, 0);
      }
    }
  };
  auto epilogue = [&](const Tile& tc) __attribute__((always_inline)) {
    if constexpr (F32N) {
      epilogue_n(tc);
      return;
    }
    // (two copies of the epilogue cost the two-fragment instantiations 10-15 %: they keep the masked form only)
    if (NF == 1 && tc.tx * 16 + 16 <= W && tc.ty * TH + TH <= H)
      epilogue_t(tc, std::true_type());
    else
      epilogue_t(tc, std::false_type());
  };


  // ---- segments = runs of tiles of one statistics group (coefficients, seeds and accumulators belong to the group); inside
  //      a segment the tile loop has conv_fwd_v2_kernel's ordering
  const int tpi = a.tilesX * a.tilesY;
  int t = t_begin;
  while (t < t_end) {
    const int grp = a.gimages > 0 ? (t / tpi) / a.gimages : 0;
    const int seg_end = a.gimages > 0 ? min(t_end, (grp + 1) * a.gimages * tpi) : t_end;
    load_coefs(grp);
    stats_clear();
    int cur = -1, nxt = t;
    Tile tcur = tile_at(t), tnxt = tcur;
    issue(tnxt, grp);
    while (true) {
      if (cur >= 0) {
        mma();
        fi_lds_barrier();                                        // every wave has read tile `cur` out of LDS
      }
      const int done = cur;
      const Tile tdone = tcur;
      if (nxt < seg_end) {
        commit(tnxt, grp);                                      // waits for the loads issued a tile ago
        cur = nxt;
        tcur = tnxt;
        ++nxt;
        tnxt = tile_next(tnxt);
        if (nxt < seg_end) issue(tnxt, grp);
      }
      if (done >= 0) epilogue(tdone);                           // its stores go out behind the loads just issued
      if (done == cur) break;                                   // nothing was committed: the segment is finished
      fi_lds_barrier();                                          // tile `cur` is in LDS
    }
    stats_flush(grp);
    t = seg_end;
  }
}

template <typename T, int CK>
static int launch_conv_thin_f32n(const ConvArgs& a, int wgs_per_cu, hipStream_t st) {
  constexpr int CKP = FiLdsStride<T, CK>::value;
  const size_t lds = (size_t)(18 * 18 * CKP) * sizeof(T) + (size_t)4 * 16 * 2 * sizeof(float);
  const long ntile = (long)a.N * a.tilesX * a.tilesY;
  long blocks = 256L * wgs_per_cu;
  if (blocks > ntile) blocks = ntile;
  hipLaunchKernelGGL((conv_thin_kernel<T, 1, CK, 0, false, true>), dim3((unsigned)blocks), dim3(256), lds, st, a);
  FI_CHECK_LAUNCH();
  return 0;
}

template <typename T, int NF, int CK>
static int launch_conv_thin(const ConvArgs& a, int wgs_per_cu, hipStream_t st) {
  constexpr int CKP = FiLdsStride<T, CK>::value;
  const size_t lds = (size_t)(18 * 18 * CKP) * sizeof(T) + (size_t)4 * NF * 16 * 2 * sizeof(float);
  const long ntile = (long)a.N * a.tilesX * a.tilesY;
  long blocks = 256L * wgs_per_cu;
  if (blocks > ntile) blocks = ntile;
  const dim3 g((unsigned)blocks), b(256);
  if (a.xf == 0) {
    if (a.c1 == 0) {
      hipLaunchKernelGGL((conv_thin_kernel<T, NF, CK, 0, false>), g, b, lds, st, a);
    } else {
      hipLaunchKernelGGL((conv_thin_kernel<T, NF, CK, 0, true>), g, b, lds, st, a);
    }
  } else if (a.xf == 1) {
    if (a.c1 == 0) {
      hipLaunchKernelGGL((conv_thin_kernel<T, NF, CK, 1, false>), g, b, lds, st, a);
    } else {
      hipLaunchKernelGGL((conv_thin_kernel<T, NF, CK, 1, true>), g, b, lds, st, a);
    }
  } else {
    if constexpr (CK == 16) {
      hipLaunchKernelGGL((conv_thin_kernel<T, NF, CK, 2, false>), g, b, lds, st, a);        // pooled sources: c1 == 0 (host)
    } else {
      return FI_ERR_UNSUPPORTED;
    }
  }
  FI_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// wgrad:  dw[co][t][ci] += sum_pix dy[pix][co] * x[pix + tap t][ci]
//   GEMM view M = Cout, N = Cin (per tap), K = pixels.  A workgroup owns a (16*NFO) x (16*NFI)
//   channel tile and walks spatial tiles (grid-stride), staging the x halo tile and the dy tile
//   in LDS.  The 4 waves split the tile's pixels (K), each keeping all 9*NFO*NFI accumulators;
//   they are combined through LDS atomics and flushed with one fp32 global atomic per element.
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
  const void* x0;
  const void* x1;
  const void* dy;
  float* dw;
  float* dbias;
  float* part;          // workspace [spatialBlocks][part_stride] for the two-stage reduction, or NULL (atomics)
  size_t part_stride;   // cout*KK*cin + cout (bias), in floats
  int N, H, W;
  int c0, c1, cout;
  int tilesX, tilesY, nco, nci, spatialBlocks;
  int depth;            // > 0: 3x3x3 filter gradient in ONE launch -- an "image" is a slice of a volume of `depth` slices, the three
                        // depth taps are channel groups of the input side (group kd of the c0 + c1 channels reads slice n + kd - 1,
                        // zero outside the volume): dw [cout][9][3][c0 + c1], the layout of the one-launch forward operand
#ifdef FI_TRACE
  long long* trace;
#endif
};

// load_cat_clamped for the input side of the one-launch 3D filter gradient: channel ci of the 3 (c0 + c1)-wide contraction
// belongs to depth tap kd = ci / (c0 + c1) and is read from slice n + kd - 1 of the same volume.
template <typename T>
__device__ __forceinline__ typename DT<T>::vec_t load_cat_depth(const T* __restrict__ x0, const T* __restrict__ x1, int c0,
                                                                int c1, int n, int gy, int gx, int H, int W, int ci,
                                                                bool live, int depth) {
  const int cr = c0 + c1;
  const int kd = ci >= 2 * cr ? 2 : (ci >= cr ? 1 : 0);
  const int d = n % depth + kd - 1;
  const bool ok = live && ci < 3 * cr && d >= 0 && d < depth;
  return load_cat_clamped<T>(x0, x1, c0, c1, ok ? n + kd - 1 : n, gy, gx, H, W, ok ? ci - kd * cr : 0, ok);
}

#ifdef FI_TRACE
#define FI_TR_BEGIN()                                                                                       \
  do {                                                                                                      \
    FI_TR(0);                                                                                               \
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 8 + 6] = (long long)wall_clock64();      \
  } while (0)
#define FI_TR_END()                                                                                         \
  do {                                                                                                      \
    FI_TR(5);                                                                                               \
    if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 8 + 2] = (long long)wall_clock64();      \
  } while (0)
#else
#define FI_TR_BEGIN() do { } while (0)
#define FI_TR_END() do { } while (0)
#endif

// Fragment loaders for the wgrad contraction (K = pixels).  `tile` is a row-major [pixel][stride] LDS
// image (pixel = row*rowlen + col); the fragment covers 16 channels starting at `chan` and the KSTEP
// pixels of k-step `ks`, displaced by the filter tap (r, s).
template <typename T> struct WgFrag;
template <> struct WgFrag<float> {
  // v_mfma_f32_16x16x4_f32: lane (li, kg) feeds channel chan+li of pixel ks*4+kg -- a plain 4-byte read.
  static __device__ __forceinline__ float load(const float* tile, int rowlen, int stride, int ks, int r, int s,
                                               int chan, int kg, int li) {
    const int p = ks * 4 + kg;
    return tile[(((p >> 4) + r) * rowlen + (p & 15) + s) * stride + chan + li];
  }
  static __device__ __forceinline__ float ones() { return 1.0f; }
};
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
template <> struct WgFrag<bf16_t> {
  // v_mfma_f32_16x16x32_bf16 wants 8 contraction (pixel) values per lane, but the tile is pixel-major.
  // ds_read_b64_tr_b16 is the gfx950 transpose read for exactly this: inside each 16-lane group every
  // lane fetches 8 bytes (4 channels of ONE pixel) and receives 4 PIXELS of ONE channel:
  //   out[lane i][j] = in[lane 4*j + (i>>2)][i&3]
  // so lane i addresses pixel (4*kg + (i>>2)), channels chan + 4*(i&3) .. +3, and gets channel chan+i at
  // pixels 4*kg .. 4*kg+3 (tests/test_ops_gpu.py::test_tr16_semantics pins this on hardware).
  // KSTEP = 32 pixels = tile rows 2ks and 2ks+1: two transpose reads (A and B use the same pixel order,
  // which is all the contraction needs).  Tap shifts are row/column offsets of 8-byte aligned addresses.
  static __device__ __forceinline__ bf16x8 load(const bf16_t* tile, int rowlen, int stride, int ks, int r, int s,
                                                int chan, int kg, int li) {
    const bf16_t* p0 = tile + ((2 * ks + r) * rowlen + 4 * kg + (li >> 2) + s) * stride + chan + 4 * (li & 3);
    const bf16_t* p1 = p0 + rowlen * stride;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    union {
      s16x4_t h[2];
      bf16x8 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
  }
  static __device__ __forceinline__ bf16x8 ones() {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)1.0f;
    return v;
  }
};

template <> struct WgFrag<f16_t> {
  // same transpose read as bf16: ds_read_b64_tr_b16 moves 16-bit lanes, whatever they encode
  static __device__ __forceinline__ f16x8 load(const f16_t* tile, int rowlen, int stride, int ks, int r, int s,
                                                int chan, int kg, int li) {
    const f16_t* p0 = tile + ((2 * ks + r) * rowlen + 4 * kg + (li >> 2) + s) * stride + chan + 4 * (li & 3);
    const f16_t* p1 = p0 + rowlen * stride;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    union {
      s16x4_t h[2];
      f16x8 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
  }
  static __device__ __forceinline__ f16x8 ones() {
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (f16_t)1.0f;
    return v;
  }
};

template <typename T, int KS, int TH, int NFO, int NFI, bool VECX, bool VECD>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int HALO = KS / 2, XW = 16 + 2 * HALO, XH = TH + 2 * HALO, KK = KS * KS;
  constexpr int VG = DT<T>::VG, KSTEP = DT<T>::KSTEP;
  constexpr int BCI = NFI * 16, BCO = NFO * 16;
  constexpr int XP = BCI + VG, DP = BCO + VG;  // padded LDS pixel strides (16-byte multiples)
  constexpr int VPX = BCI / VG, VPD = BCO / VG;
  constexpr int NKS = TH * 16 / KSTEP;         // k-steps per spatial tile
  typedef typename DT<T>::vec_t vec_t;
  typedef typename DT<T>::frag_t frag_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* xs = reinterpret_cast<T*>(smem);           // [XH*XW][XP]
  T* ds = xs + XH * XW * XP;                    // [TH*16][DP]
  float* red = reinterpret_cast<float*>(smem);  // reused at the end: [KK][BCO][BCI] then [BCO] (bias)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;
  const int cin = (a.depth > 0 ? 3 : 1) * (a.c0 + a.c1), cout = a.cout;
  // (cit, cot) workgroups of one spatial block read the same gradient / input tiles at the same time: XCD g (= blockIdx % 8, an L2
  // each) takes a CONTIGUOUS range of the (sb, cot, cit) order instead of every eighth workgroup
  int bid;
  {
    const unsigned B = gridDim.x, g = blockIdx.x & 7u, qq = blockIdx.x >> 3, Bq = B >> 3, r = B & 7u;
    bid = (int)(g * Bq + (g < r ? g : r) + qq);
  }
  const int cit = bid % a.nci;
  bid /= a.nci;
  const int cot = bid % a.nco;
  const int sb = bid / a.nco;
  const int H = a.H, W = a.W;
  const T* x0 = reinterpret_cast<const T*>(a.x0);
  const T* x1 = reinterpret_cast<const T*>(a.x1);
  const T* dyg = reinterpret_cast<const T*>(a.dy);
  const bool want_bias = a.dbias != nullptr && cit == 0;

  f32x4 acc[KK][NFO][NFI];
  f32x4 accb[NFO];                              // dbias via a ones operand: D[co][*] = sum_pix dy[pix][co]
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int o = 0; o < NFO; ++o)
#pragma unroll
      for (int i = 0; i < NFI; ++i) acc[t][o][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < NFO; ++o) accb[o] = f32x4{0.f, 0.f, 0.f, 0.f};
  const frag_t onesv = WgFrag<T>::ones();

  // register-staged tiles: all loads of a tile are issued back to back, and the NEXT tile's loads are in
  // flight while the MFMAs of the current one run (the staging loop used to expose one HBM latency per
  // 256-vector round).
  constexpr int NXV = XH * XW * VPX, NDV = TH * 16 * VPD;
  constexpr int NX = (NXV + 255) / 256, ND = (NDV + 255) / 256;
  vec_t xr[NX], dr[ND];
  auto fetch = [&](int tile) {
    const int tx = tile % a.tilesX, ty = (tile / a.tilesX) % a.tilesY, n = tile / (a.tilesX * a.tilesY);
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int i = tid + it * 256;
      vec_t val;
      if constexpr (VECX) {
        const int ii = i < NXV ? i : 0;
        const int v = ii % VPX, pix = ii / VPX;
        const int py = pix / XW, px = pix % XW;
        val = a.depth > 0 ? load_cat_depth<T>(x0, x1, a.c0, a.c1, n, ty * TH + py - HALO, tx * 16 + px - HALO, H, W,
                                              cit * BCI + v * VG, i < NXV, a.depth)
                          : load_cat_clamped<T>(x0, x1, a.c0, a.c1, n, ty * TH + py - HALO, tx * 16 + px - HALO, H, W,
                                                cit * BCI + v * VG, i < NXV);
      } else {
        memset(&val, 0, sizeof(val));
        if (i < NXV) {
          const int v = i % VPX, pix = i / VPX;
          const int py = pix / XW, px = pix % XW;
          const int gy = ty * TH + py - HALO, gx = tx * 16 + px - HALO;
          if (gy >= 0 && gy < H && gx >= 0 && gx < W)
            val = load_cat<T>(x0, x1, a.c0, a.c1, ((size_t)n * H + gy) * W + gx, cit * BCI + v * VG, false);
        }
      }
      xr[it] = val;
    }
#pragma unroll
    for (int it = 0; it < ND; ++it) {
      const int i = tid + it * 256;
      vec_t val;
      if constexpr (VECD) {
        const int ii = i < NDV ? i : 0;
        const int v = ii % VPD, pix = ii / VPD;
        val = load_cat_clamped<T>(dyg, dyg, cout, 0, n, ty * TH + pix / 16, tx * 16 + pix % 16, H, W,
                                  cot * BCO + v * VG, i < NDV);
      } else {
        memset(&val, 0, sizeof(val));
        if (i < NDV) {
          const int v = i % VPD, pix = i / VPD;
          const int gy = ty * TH + pix / 16, gx = tx * 16 + pix % 16;
          if (gy < H && gx < W)
            val = load_cat<T>(dyg, dyg, cout, 0, ((size_t)n * H + gy) * W + gx, cot * BCO + v * VG, false);
        }
      }
      dr[it] = val;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int i = tid + it * 256;
      if (i < NXV) *reinterpret_cast<vec_t*>(&xs[(i / VPX) * XP + (i % VPX) * VG]) = xr[it];
    }
#pragma unroll
    for (int it = 0; it < ND; ++it) {
      const int i = tid + it * 256;
      if (i < NDV) *reinterpret_cast<vec_t*>(&ds[(i / VPD) * DP + (i % VPD) * VG]) = dr[it];
    }
  };

  const int ntiles = a.N * a.tilesX * a.tilesY;
  FI_TR_BEGIN();
  if (sb < ntiles) fetch(sb);
  for (int tile = sb; tile < ntiles; tile += a.spatialBlocks) {
    __syncthreads();
    commit();
    __syncthreads();
    if (tile == sb) FI_TR(1);
    if (tile + a.spatialBlocks < ntiles) fetch(tile + a.spatialBlocks);

    for (int ks = wave; ks < NKS; ks += 4) {
      frag_t av[NFO];
#pragma unroll
      for (int o = 0; o < NFO; ++o) {
        av[o] = WgFrag<T>::load(ds, 16, DP, ks, 0, 0, o * 16, kg, li);
        if (want_bias) accb[o] = mfma16(av[o], onesv, accb[o]);
      }
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const int r = t / KS, s = t % KS;
#pragma unroll
        for (int i = 0; i < NFI; ++i) {
          const frag_t bv = WgFrag<T>::load(xs, XW, XP, ks, r, s, i * 16, kg, li);
#pragma unroll
          for (int o = 0; o < NFO; ++o) acc[t][o][i] = mfma16(av[o], bv, acc[t][o][i]);
        }
      }
    }
  }

  FI_TR(3);
  // ---- combine the 4 waves through LDS: the waves take turns (plain ds_read/ds_write, fixed order ->
  //      deterministic).  LDS float atomics cost ~220 cycles per wave-instruction here and made the kernel
  //      LDS-bound (SQ_WAIT_INST_LDS 43 % of wave cycles); the turn-taking costs 4 barriers instead.
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int o = 0; o < NFO; ++o)
#pragma unroll
          for (int i = 0; i < NFI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // D[row = co = kg*4+r][col = ci = li]
              float* dst = &red[(t * BCO + o * 16 + kg * 4 + r) * BCI + i * 16 + li];
              *dst = (w == 0) ? acc[t][o][i][r] : *dst + acc[t][o][i][r];
            }
      if (li == 0) {
#pragma unroll
        for (int o = 0; o < NFO; ++o)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* dst = &red[KK * BCO * BCI + o * 16 + kg * 4 + r];
            *dst = (w == 0) ? accb[o][r] : *dst + accb[o][r];
          }
      }
    }
  }
  __syncthreads();
  FI_TR(4);
  if (a.part) {
    // two-stage, deterministic: this workgroup's partial sums go to its own slice of the caller's
    // workspace with plain stores; wgrad_reduce_kernel then adds the slices in a fixed order.
    float* slice = a.part + (size_t)sb * a.part_stride;
    for (int i = tid; i < KK * BCO * BCI; i += 256) {
      const int ci = i % BCI, co = (i / BCI) % BCO, t = i / (BCI * BCO);
      const int gco = cot * BCO + co, gci = cit * BCI + ci;
      if (gco < cout && gci < cin) slice[((size_t)gco * KK + t) * cin + gci] = red[i];
    }
    if (a.dbias && cit == 0 && tid < BCO) {
      const int gco = cot * BCO + tid;
      if (gco < cout) slice[(size_t)cout * KK * cin + gco] = red[KK * BCO * BCI + tid];
    }
    FI_TR_END();
    return;
  }
  for (int i = tid; i < KK * BCO * BCI; i += 256) {
    const int ci = i % BCI, co = (i / BCI) % BCO, t = i / (BCI * BCO);
    const int gco = cot * BCO + co, gci = cit * BCI + ci;
    if (gco < cout && gci < cin) {
      const float v = red[i];
      if (v != 0.f) atomicAdd(&a.dw[((size_t)gco * KK + t) * cin + gci], v);
    }
  }
  if (want_bias && tid < BCO) {
    const int gco = cot * BCO + tid;
    if (gco < cout) atomicAdd(&a.dbias[gco], red[KK * BCO * BCI + tid]);
  }
}

// second stage of the deterministic wgrad: dw[i] += sum_s part[s][i]  (i < n_dw), dbias likewise
// 256 threads = 32 consecutive elements x 8 slice groups: group g sums slices g, g+8, ... (independent
// loads in flight), the 8 group sums are then added in a fixed order through LDS -> deterministic.
static __global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, size_t stride, int slices,
                                                           float* __restrict__ dw, size_t n_dw,
                                                           float* __restrict__ dbias, int cout) {
  __shared__ float sm[8][32];
  const size_t n = n_dw + (dbias ? (size_t)cout : 0);
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  for (size_t base = (size_t)blockIdx.x * 32; base < n; base += (size_t)gridDim.x * 32) {
    const size_t i = base + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
      int k = g;
      for (; k + 24 < slices; k += 32) {
        s0 += part[(size_t)k * stride + i];
        s1 += part[(size_t)(k + 8) * stride + i];
        s2 += part[(size_t)(k + 16) * stride + i];
        s3 += part[(size_t)(k + 24) * stride + i];
      }
      for (; k < slices; k += 8) s0 += part[(size_t)k * stride + i];
    }
    __syncthreads();
    sm[g][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < n) {
      float s = sm[0][e];
#pragma unroll
      for (int q = 1; q < 8; ++q) s += sm[q][e];
      if (i < n_dw)
        dw[i] += s;
      else
        dbias[i - n_dw] += s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad for the channel-rich layers (Cin, Cout >= 32): a workgroup owns a 32 x 32 channel tile and
// each of its 4 waves one 16 x 16 quadrant of it (all k*k taps, all pixels of the tile).  Compared
// with the pixel-split kernel above this doubles the MFMAs per staged tile, needs no cross-wave LDS
// reduction, and writes 4x fewer partial slices per channel for the same number of workgroups.
// ---------------------------------------------------------------------------------------------
template <typename T, int KS, int TH, bool VECX, bool VECD>
__global__ __launch_bounds__(256) void conv_wgrad_quad_kernel(WgradArgs a) {
  constexpr int HALO = KS / 2, XW = 16 + 2 * HALO, XH = TH + 2 * HALO, KK = KS * KS;
  constexpr int VG = DT<T>::VG, KSTEP = DT<T>::KSTEP;
  constexpr int BC = 32, XP = BC + VG, VPX = BC / VG;
  constexpr int NKS = TH * 16 / KSTEP;
  typedef typename DT<T>::vec_t vec_t;
  typedef typename DT<T>::frag_t frag_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* xs = reinterpret_cast<T*>(smem);           // [XH*XW][XP]
  T* ds = xs + XH * XW * XP;                    // [TH*16][XP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kg = lane >> 4;
  const int qo = wave >> 1, qi = wave & 1;
  const int cin = (a.depth > 0 ? 3 : 1) * (a.c0 + a.c1), cout = a.cout;
  // (cit, cot) workgroups of one spatial block read the same gradient / input tiles at the same time: XCD g (= blockIdx % 8, an L2
  // each) takes a CONTIGUOUS range of the (sb, cot, cit) order instead of every eighth workgroup
  int bid;
  {
    const unsigned B = gridDim.x, g = blockIdx.x & 7u, qq = blockIdx.x >> 3, Bq = B >> 3, r = B & 7u;
    bid = (int)(g * Bq + (g < r ? g : r) + qq);
  }
  const int cit = bid % a.nci;
  bid /= a.nci;
  const int cot = bid % a.nco;
  const int sb = bid / a.nco;
  const int H = a.H, W = a.W;
  const T* x0 = reinterpret_cast<const T*>(a.x0);
  const T* x1 = reinterpret_cast<const T*>(a.x1);
  const T* dyg = reinterpret_cast<const T*>(a.dy);
  const bool want_bias = a.dbias != nullptr && cit == 0 && qi == 0;

  f32x4 acc[KK];
  f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < KK; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const frag_t onesv = WgFrag<T>::ones();

  constexpr int NXV = XH * XW * VPX, NDV = TH * 16 * VPX;
  constexpr int NX = (NXV + 255) / 256, ND = (NDV + 255) / 256;
  vec_t xr[NX], dr[ND];
  auto fetch = [&](int tile) {   // register-staged, next tile prefetched during the MFMAs (see pixel-split kernel)
    const int tx = tile % a.tilesX, ty = (tile / a.tilesX) % a.tilesY, n = tile / (a.tilesX * a.tilesY);
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int i = tid + it * 256;
      vec_t val;
      if constexpr (VECX) {
        const int ii = i < NXV ? i : 0;
        const int v = ii % VPX, pix = ii / VPX;
        const int py = pix / XW, px = pix % XW;
        val = a.depth > 0 ? load_cat_depth<T>(x0, x1, a.c0, a.c1, n, ty * TH + py - HALO, tx * 16 + px - HALO, H, W,
                                              cit * BC + v * VG, i < NXV, a.depth)
                          : load_cat_clamped<T>(x0, x1, a.c0, a.c1, n, ty * TH + py - HALO, tx * 16 + px - HALO, H, W,
                                                cit * BC + v * VG, i < NXV);
      } else {
        memset(&val, 0, sizeof(val));
        if (i < NXV) {
          const int v = i % VPX, pix = i / VPX;
          const int py = pix / XW, px = pix % XW;
          const int gy = ty * TH + py - HALO, gx = tx * 16 + px - HALO;
          if (gy >= 0 && gy < H && gx >= 0 && gx < W)
            val = load_cat<T>(x0, x1, a.c0, a.c1, ((size_t)n * H + gy) * W + gx, cit * BC + v * VG, false);
        }
      }
      xr[it] = val;
    }
#pragma unroll
    for (int it = 0; it < ND; ++it) {
      const int i = tid + it * 256;
      vec_t val;
      if constexpr (VECD) {
        const int ii = i < NDV ? i : 0;
        const int v = ii % VPX, pix = ii / VPX;
        val = load_cat_clamped<T>(dyg, dyg, cout, 0, n, ty * TH + pix / 16, tx * 16 + pix % 16, H, W,
                                  cot * BC + v * VG, i < NDV);
      } else {
        memset(&val, 0, sizeof(val));
        if (i < NDV) {
          const int v = i % VPX, pix = i / VPX;
          const int gy = ty * TH + pix / 16, gx = tx * 16 + pix % 16;
          if (gy < H && gx < W)
            val = load_cat<T>(dyg, dyg, cout, 0, ((size_t)n * H + gy) * W + gx, cot * BC + v * VG, false);
        }
      }
      dr[it] = val;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int i = tid + it * 256;
      if (i < NXV) *reinterpret_cast<vec_t*>(&xs[(i / VPX) * XP + (i % VPX) * VG]) = xr[it];
    }
#pragma unroll
    for (int it = 0; it < ND; ++it) {
      const int i = tid + it * 256;
      if (i < NDV) *reinterpret_cast<vec_t*>(&ds[(i / VPX) * XP + (i % VPX) * VG]) = dr[it];
    }
  };

  const int ntiles = a.N * a.tilesX * a.tilesY;
  FI_TR_BEGIN();
  if (sb < ntiles) fetch(sb);
  for (int tile = sb; tile < ntiles; tile += a.spatialBlocks) {
    __syncthreads();
    commit();
    __syncthreads();
    if (tile == sb) FI_TR(1);
    if (tile + a.spatialBlocks < ntiles) fetch(tile + a.spatialBlocks);
#pragma unroll 2
    for (int ks = 0; ks < NKS; ++ks) {
      const frag_t av = WgFrag<T>::load(ds, 16, XP, ks, 0, 0, qo * 16, kg, li);
      if (want_bias) accb = mfma16(av, onesv, accb);
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const frag_t bv = WgFrag<T>::load(xs, XW, XP, ks, t / KS, t % KS, qi * 16, kg, li);
        acc[t] = mfma16(av, bv, acc[t]);
      }
    }
  }

  FI_TR(3);
  FI_TR(4);
  // ---- flush: D[row = co = kg*4+r][col = ci = li] of this wave's quadrant
  const size_t n_dw = (size_t)cout * KK * cin;
  float* slice = a.part ? a.part + (size_t)sb * a.part_stride : nullptr;
  const int gci = cit * BC + qi * 16 + li;
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gco = cot * BC + qo * 16 + kg * 4 + r;
      if (gco < cout && gci < cin) {
        const size_t o = ((size_t)gco * KK + t) * cin + gci;
        if (slice)
          slice[o] = acc[t][r];
        else
          atomicAdd(&a.dw[o], acc[t][r]);
      }
    }
  if (want_bias && li == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gco = cot * BC + qo * 16 + kg * 4 + r;
      if (gco < cout) {
        if (slice)
          slice[n_dw + gco] = accb[r];
        else
          atomicAdd(&a.dbias[gco], accb[r]);
      }
    }
  }
  FI_TR_END();
}

template <typename T, int KS, int TH>
static int launch_conv_wgrad_quad(const WgradArgs& a, hipStream_t st) {
  constexpr int HALO = KS / 2, XW = 16 + 2 * HALO, XH = TH + 2 * HALO;
  constexpr int XP = 32 + DT<T>::VG;
  const size_t lds = (size_t)(XH * XW * XP + TH * 16 * XP) * sizeof(T);
  const long blocks = (long)a.spatialBlocks * a.nco * a.nci;
  // the quadrant kernel is only chosen for cin, cout >= 32; odd channel counts take the generic loaders
  if (a.c0 % DT<T>::VG == 0 && a.c1 % DT<T>::VG == 0 && a.cout % DT<T>::VG == 0)
    hipLaunchKernelGGL((conv_wgrad_quad_kernel<T, KS, TH, true, true>), dim3((unsigned)blocks), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((conv_wgrad_quad_kernel<T, KS, TH, false, false>), dim3((unsigned)blocks), dim3(256), lds, st, a);
  FI_CHECK_LAUNCH();
  return 0;
}

template <typename T, int KS, int TH, int NFO, int NFI>
static int launch_conv_wgrad(const WgradArgs& a, hipStream_t st) {
  constexpr int HALO = KS / 2, XW = 16 + 2 * HALO, XH = TH + 2 * HALO, KK = KS * KS;
  constexpr int VG = DT<T>::VG;
  constexpr int BCI = NFI * 16, BCO = NFO * 16, XP = BCI + VG, DP = BCO + VG;
  size_t lds = (size_t)(XH * XW * XP + TH * 16 * DP) * sizeof(T);
  const size_t red = (size_t)(KK * BCO * BCI + BCO) * sizeof(float);
  if (lds < red) lds = red;
  const long blocks = (long)a.spatialBlocks * a.nco * a.nci;
  const bool vx = a.c0 % VG == 0 && a.c1 % VG == 0, vd = a.cout % VG == 0;
  const dim3 g((unsigned)blocks), b(256);
  if (vx && vd)
    hipLaunchKernelGGL((conv_wgrad_kernel<T, KS, TH, NFO, NFI, true, true>), g, b, lds, st, a);
  else if (vd)     // first layer: 1-channel image, 16-channel gradient
    hipLaunchKernelGGL((conv_wgrad_kernel<T, KS, TH, NFO, NFI, false, true>), g, b, lds, st, a);
  else if (vx)     // logits layer: 16-channel input, 2-channel gradient
    hipLaunchKernelGGL((conv_wgrad_kernel<T, KS, TH, NFO, NFI, true, false>), g, b, lds, st, a);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<T, KS, TH, NFO, NFI, false, false>), g, b, lds, st, a);
  FI_CHECK_LAUNCH();
  return 0;
}
