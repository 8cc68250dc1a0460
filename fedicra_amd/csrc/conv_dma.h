// Forward / dgrad kernel, sixth form: the GEMM-shaped tile for the channel-rich 3x3 layers -- 512 pixels x 128 output channels per
// workgroup, every operand byte brought into LDS by LDS-DMA (buffer_load_dwordx4 ... lds), no producer waves.
//
// Why (round 6; DESIGN.md 4.1, LOG.md (6) / (10), profiles/r03_h_ws2_trace.txt): conv_fwd_ws2_kernel's 256-pixel x 128-channel tile streams a 37 KB
// weight slab per 10 KB of pixels through the CU's ~20 B/clk L2 -> LDS fill path -- 2 350 cycles of fill for 2 304 cycles of MFMAs --
// and its BatchNorm / LeakyReLU / dropout input transform is the vector work of ONE 4-wave producer team at a time.  Here
//   * a weight slab serves TWICE the pixels: two 16 x 16-pixel sub-tiles (A, B: consecutive tiles of the launch) share the slab
//     stream, so a stage is 4 608 MFMA cycles per SIMD for 58 KB of fill (12 B/clk);
//   * nothing passes through registers on its way in: weights AND pixels are written to LDS by the DMA path (zero VGPRs, ~2 vector
//     instructions per KB for the per-lane source offset).  The LDS layouts are conv_fwd_ws2_kernel's conflict-free ones; the
//     DMA's destination is linear (M0 base + 16 lane), so the swizzles are applied on the SOURCE side -- lane L fetches the 16
//     bytes that belong at LDS slot L (tools/ldsdma_probe.hip); lanes outside the image fetch through an out-of-range offset and
//     the hardware writes zeros;
//   * all 16 waves are consumers (64 pixels x 64 channels each: 2 x 2 accumulator blocks of v_mfma_f32_32x32x16, 128 registers, 4
//     waves per SIMD) AND transformers: the fused loader's z = dropout(act(scale y + shift)) is applied IN PLACE in LDS, one
//     16-channel chunk per stage, by the 8 waves that will consume it -- sixteen waves' worth of vector issue instead of four.  The
//     waves of sub-tile A transform before their MFMAs and those of B after, so that on every SIMD two waves are on the vector pipe
//     while two are on the matrix pipe.
//
//   workgroup = 16 waves = 2 sub-tiles x (4 row groups x 2 channel halves); persistent over a run of tile PAIRS of one slab
//   stage     = one 16-channel chunk: 9 taps x 4 MFMAs per wave; one barrier per stage
//   pixels    = groups of 32 channels (64 bytes of a pixel: half a line), two group buffers: group g + 1 lands while g's first chunk is
//               consumed, its chunks are transformed during g's second / its own first stage
//   weights   = the chunk-major operand [Cin/16][Cout][9][16] (fi_pack_weights modes 2 / 3); two slab buffers, slab s + 1 lands during stage s
//   LDS       = 2 x 2 x 21 KB pixels + 2 x 36 KB weights + 3 KB statistics strips = 159 KB
// Same transforms (XF 0 / 1), rounding, statistics slots and group semantics as the other forms; the epilogue is conv_fwd_ws2_kernel's.
#pragma once
#include "conv_ws2.h"

#ifndef FI_DMA_DEBUG
#define FI_DMA_DEBUG 0         // A/B builds: 1 no MFMAs, 2 no DMA, 8 no transform, 16 no epilogue
#endif

// One LDS-DMA instruction: lane L's 16 bytes at buffer offset `voff` (zeros beyond the resource) go to LDS[lds_addr + 16 L].  Inline assembly,
// not __builtin_amdgcn_raw_ptr_buffer_load_lds: hipcc tracks the builtin as an LDS store and puts s_waitcnt vmcnt(0) in front of the next LDS
// read it cannot prove disjoint (here: every stage's first ds_read -- the DMA would land before the stage's MFMAs start instead of beside
// them).  The kernel orders its DMA itself: s_waitcnt vmcnt(0) + the stage barrier before anything reads a landed buffer.  (M0 is written
// here only; nothing else in the kernel uses it.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
typedef int fi_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fi_lds_dma16(fi_v4i rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ fi_v4i fi_raw_rsrc(const void* ptr, unsigned bytes) {      // raw buffer: stride 0, `bytes` records, 32-bit data format
  const unsigned long long p = (unsigned long long)ptr;
  return fi_v4i{__builtin_amdgcn_readfirstlane((int)(unsigned)p), __builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu)),
                __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}

template <typename T, int XF>
__global__ __launch_bounds__(1024, 1) void conv_fwd_dma_kernel(ConvArgs a) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  static_assert(XF == 0 || XF == 1, "plain or transforming loader");
  constexpr int BN = 128, CG = 2, RG = 4, WPS = RG * CG, NS = 2;  // waves per sub-tile; sub-tiles
  constexpr int NCB = 2;                                         // 32-channel accumulator blocks of a wave
  constexpr int XH = 18, XW = 18, NPX = XH * XW, KK = 9, CK = 16, VG = 8;
  constexpr int GC = 32, NP = GC / VG;                           // channels of a pixel group; 16-byte pieces per staged pixel
  constexpr int XI = (NPX * NP + 63) / 64;                       // DMA instructions (1 KB each) of a sub-tile's group: 21
  constexpr int XS = XI * 1024, XG = NS * XS;                    // bytes of a sub-tile's / a whole group buffer
  constexpr int WROW = KK * CK * 2, WT = BN * WROW, WI = WT / 1024;   // weight row (288 B, unpadded) / slab bytes / DMA instructions (36)
  constexpr int O_W = 2 * XG, O_S = O_W + 2 * WT;                // LDS map: [2] pixel groups, [2] weight slabs, [NS x CG] strips
  static_assert(WI * 1024 == WT, "whole DMA instructions");
  constexpr int O_T = O_S + NS * CG * 192 * 4;                   // [2 chunk parities][NS] x (16 scale + 16 shift floats): the transform's coefficients
  static_assert(O_T + 2 * NS * 128 <= 160 * 1024, "LDS");
  typedef typename DT<T>::vec_t vec_t;
  typedef typename DT<T>::frag_t frag_t;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) char* lds_cptr_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_cptr_t)smem;   // LDS byte address of smem[0]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int st = wave / WPS, wv = wave % WPS;                    // sub-tile of this wave, its index among the sub-tile's waves
  const int rg = wv % RG, cgp = wv / RG;
  const int cin = a.c0 + a.c1, cout = a.co0 + a.co1;
  const int H = a.H, W = a.W;
  auto uni = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
#ifdef FI_TRACE
  int tn = 0;
#endif

  // ---- this workgroup's run: slab ct, tiles [t_begin, t_end) of the launch; sub-tile A takes the even, B the odd positions
  const int ntile = a.N * a.tilesY * a.tilesX, tpi = a.tilesY * a.tilesX;
  const int g1 = (int)gridDim.x / a.nct;                         // workgroups per slab (host: gridDim.x == nct * g1)
  const int ct = (int)blockIdx.x / g1, r1 = (int)blockIdx.x - ct * g1;
  const int t_begin = (int)((long)ntile * r1 / g1), t_end = (int)((long)ntile * (r1 + 1) / g1);
  const int cnt = t_end - t_begin;
  if (cnt <= 0) return;
  const int nchunk = cin / CK, ngrp = cin / GC;
  const int npair = (cnt + 1) >> 1;                              // items of sub-tile A; B has cnt >> 1
  const int nmine = st == 0 ? npair : (cnt >> 1);
  const int nstage = npair * nchunk;

  constexpr unsigned esz = 2;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const unsigned hw = (unsigned)H * (unsigned)W;

  struct Item {
    int tx, ty, n, grp, nl;                                      // tile, image, its statistics / coefficient group and index within it
  };
  auto item_at = [&](int tile) {
    Item c;
    c.n = tile / tpi;
    const int r = tile - c.n * tpi;
    c.ty = r / a.tilesX;
    c.tx = r - c.ty * a.tilesX;
    c.grp = a.gimages > 0 ? c.n / a.gimages : 0;
    c.nl = c.n - c.grp * a.gimages;
    return c;
  };
  auto item_step = [&](Item c, int steps) {                      // `steps` (1 or 2) tiles on
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k >= steps) break;
      if (++c.tx == a.tilesX) {
        c.tx = 0;
        if (++c.ty == a.tilesY) {
          c.ty = 0;
          ++c.n;
          if (++c.nl == a.gimages) c.nl = 0, ++c.grp;            // (gimages == 0: one group, nl = n)
        }
      }
    }
    return c;
  };
  auto item_next2 = [&](Item c) { return item_step(c, 2); };

  auto rsrc = [&](const void* ptr, unsigned bytes) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
  };
  const unsigned wbytes = (unsigned)nchunk * (unsigned)a.wrows * (KK * CK * esz);
  const unsigned xbytes0 = (unsigned)((XF != 0 && a.bcast0) ? a.gimages : a.N) * hw * (unsigned)a.c0 * esz;
  const unsigned xbytes1 = (unsigned)a.N * hw * (unsigned)a.c1 * esz;
  const fi_v4i rw = fi_raw_rsrc(a.w, wbytes);

  // ---- All DMA instructions of a stage are issued by the 8 waves of sub-tile A, BEFORE their taps; the waves of sub-tile B start their taps
  // right behind the stage barrier.  (tools/dma_trace.py: with every wave issuing first, the matrix pipe idled ~1 000 cycles at the head of
  // every stage; the older half of a workgroup -- A -- wins the issue arbitration and ended its taps ~3 000 cycles before B anyway.)
  // ---- weight slab: 36 instructions (j = wv, wv + 8, ...).  LDS vector v = co * 18 + t * 2 + h holds source vector co * 18 + t * 2 +
  // (h ^ ((co >> 3) & 1)): rows of 288 bytes, the halves of a tap swapped on rows 8..15 mod 16.  (wbase: byte offset of the slab in the
  // chunk-major operand)
  const unsigned wstride = (unsigned)a.wrows * (unsigned)WROW, wbase0 = (unsigned)(ct * BN) * (unsigned)WROW;
  auto issue_w = [&](unsigned wbase, int par) __attribute__((always_inline)) {
#if FI_DMA_DEBUG & 2
    return;
#endif
    const unsigned dst = lds0 + (unsigned)(O_W + par * WT);
    int l = lane;
    asm volatile("" : "+v"(l));                                  // (per-lane constants recomputed per stage, not kept in registers across the run)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int j = wv + 8 * k;
      if (j < WI) {
        const int v = j * 64 + l;
        const int co = v / (KK * 2), q18 = v - co * (KK * 2);
        fi_lds_dma16(rw, dst + (unsigned)(j * 1024), wbase + (unsigned)((co * (KK * 2) + (q18 ^ ((co >> 3) & 1))) * 16));
      }
    }
  };

  // ---- pixel group of sub-tile `sub`: 21 instructions (i = wv, wv + 8, wv + 16).  LDS slot v = i * 64 + lane = pixel p = v / 4 of the
  // 18 x 18 halo list, slot v % 4, which holds source piece slot ^ ((col >> 1) & 3) of the group's four 8-channel pieces.
  // (it: the item of that sub-tile the group belongs to; cg: the group's index within the contraction)
  auto issue_x = [&](int sub, const Item& it, int cg_, int gbuf) __attribute__((always_inline)) {
#if FI_DMA_DEBUG & 2
    return;
#endif
    const int cg = uni(cg_);
    const bool first = cg * GC < a.c0;                           // a group never straddles the two sources (host)
    const unsigned cs = (unsigned)(first ? a.c0 : a.c1);
    const unsigned co = (unsigned)(cg * GC - (first ? 0 : a.c0));
    const int n = uni(it.n), ty = uni(it.ty), tx = uni(it.tx);
    const int ns = (XF != 0 && first && a.bcast0) ? uni(it.nl) : n;
    const unsigned pxb = cs * esz;
    const int y0 = ty * 16 - 1, x0 = tx * 16 - 1;
    const unsigned base = (unsigned)((ns * H + y0) * W + x0) * pxb + co * esz;    // (wraps for y0 / x0 = -1: only in-image lanes use it)
    // in-image rows / columns of the halo: [rlo, rlo + rspan) x [clo, clo + cspan)
    const unsigned rlo = y0 < 0 ? 1u : 0u, clo = x0 < 0 ? 1u : 0u;
    const unsigned rspan = (unsigned)min(XH, H - y0) - rlo, cspan = (unsigned)min(XW, W - x0) - clo;
    const unsigned dst = lds0 + (unsigned)(gbuf * XG + sub * XS);
    const fi_v4i rx = fi_raw_rsrc(first ? a.x0 : a.x1, first ? xbytes0 : xbytes1);   // (built from scalars: uniform)
    int l = lane;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = wv + 8 * k;
      if (i < XI) {
        int p = i * 16 + (l >> 2);
        p = p < NPX ? p : NPX;                                   // (beyond the list: row 18 -- fails the row test)
        const unsigned row = (unsigned)p / XW, col = (unsigned)p - row * XW;
        const bool ok = row - rlo < rspan && col - clo < cspan;
        const unsigned off = ok ? (row * (unsigned)W + col) * pxb + (base + (((unsigned)l & 3u) ^ ((col >> 1) & 3u)) * 16u) : OOB;
        fi_lds_dma16(rx, dst + (unsigned)(i * 1024), off);
      }
    }
  };

  // ---- in-place transform of one 16-channel chunk of this wave's sub-tile: 324 pixels x 2 halves over 8 waves; a wave's half (its 8
  // channels) is uniform: wave wv takes half wv & 1 of pixels (wv >> 1) * 64 + lane (+ 256 in a second pass: waves 0..3 only)
  const bool drop0 = XF == 1 && a.t0.drop_mode == FI_DROP_RNG_ELEM;
  uint64_t seed_base = 0;
  if (drop0) {
    seed_base = a.t0.seed;
    if (a.t0.seed_offset) seed_base += 0xD1B54A32D192ED03ull * (uint64_t)(uint32_t)a.t0.seed_offset[0];
  }
  // the scale / shift values of the 16 channels of chunk c of item `it`, staged in LDS one stage ahead of the transform that uses them by ONE
  // wave per sub-tile (a vector load: the compiler cannot prove the arrays unclobbered -- issued after the wave's MFMAs, where its wait, which
  // also waits for the stage's DMA, costs nothing), so that the transform itself has no wait on the vector-memory counter
  auto stage_coef = [&](const Item& it, int c_, int par) __attribute__((always_inline)) {
    if constexpr (XF == 0) return;
    const int c = uni(c_);
    const bool first = c * CK < a.c0;
    const float* const scp = first ? a.t0.scale : a.t1.scale;
    const float* const shp = first ? a.t0.shift : a.t1.shift;
    if (scp == nullptr) return;
    const int cofs = uni(uni(it.grp) * (first ? a.c0 : a.c1) + c * CK - (first ? 0 : a.c0));
    if (lane < 32) {
      const float v = lane < 16 ? scp[cofs + lane] : shp[cofs + lane - 16];
      reinterpret_cast<float*>(smem + O_T + (par * NS + st) * 128)[lane] = v;
    }
  };
  // The transform of a chunk, cut into SLICES that the stage issues between the filter taps' MFMA groups (a wave64 vector instruction holds the
  // SIMD's issue port 4 cycles and a SIMD hosts 4 waves: as separate phases -- all of a sub-tile's waves transforming, then all multiplying --
  // the matrix pipe idled half of every stage; tools/dma_trace.py).  Uniform part (Tctx, scalars) once per stage; per pass (= one 16-byte
  // vector per lane: pixel pass * 256 + (wv >> 1) * 64 + lane of the halo list, this wave's half of the chunk) two dropout-draw slices (the
  // draws need no data: they leave 8 keep bits) and the data slice (read, scale / shift / activation, keep bits, write back).  A lane
  // outside the image keeps the zeros the DMA wrote there (z = 0, not act(shift)).
  struct Tctx {
    bool on, drop;
    float slope;
    uint64_t seed;
    unsigned cs8, co8, rspan, cspan, xb, tb;
    int y0, x0, rlo, clo, piece, pix0;
  };
  struct Tpass {
    unsigned keep;                                               // bit j: element j of the vector is kept (dropout draws: they need no data)
  };
  auto t_setup = [&](const Item& it, int c_, int gbuf, bool live) __attribute__((always_inline)) {
    Tctx t;
    t.on = false;
    if constexpr (XF == 0) return t;
#if FI_DMA_DEBUG & 8
    return t;
#endif
    const int c = uni(c_);                                       // chunk of the contraction
    const int half = wv & 1;
    const int ch0 = c * CK + half * VG;
    const bool first = ch0 < a.c0;
    t.on = live && (first ? a.t0.scale : a.t1.scale) != nullptr; // (a NULL scale: this source is used as it is)
    const int cs = first ? a.c0 : a.c1;
    const int co = ch0 - (first ? 0 : a.c0);
    t.slope = first ? a.t0.slope : a.t1.slope;
    t.drop = drop0 && first;
    const int ty = uni(it.ty), tx = uni(it.tx), nl = uni(it.nl);
    t.seed = seed_base + (uint64_t)uni(it.grp) * a.t0.seed_gstride;
    t.cs8 = (unsigned)cs / VG, t.co8 = (unsigned)co / VG;
    t.y0 = ty * 16 - 1, t.x0 = tx * 16 - 1;
    t.rlo = t.y0 < 0 ? 1 : 0, t.clo = t.x0 < 0 ? 1 : 0;
    t.rspan = (unsigned)(min(XH, H - t.y0) - t.rlo), t.cspan = (unsigned)(min(XW, W - t.x0) - t.clo);
    t.piece = 2 * (c & 1) + half;
    t.xb = (unsigned)(gbuf * XG + st * XS);
    t.tb = (unsigned)(O_T + ((c & 1) * NS + st) * 128 + half * 32);
    t.pix0 = (nl * H + t.y0) * W + t.x0;
    return t;
  };
  auto t_pixel = [&](int k, int& p, int& row, int& col) __attribute__((always_inline)) {
    int l = lane;
    asm volatile("" : "+v"(l));                                  // (per-lane constants recomputed here, not kept in registers across the run)
    p = k * 256 + (wv >> 1) * 64 + l;
    p = p < NPX ? p : NPX;                                       // (beyond the list: row 18 -- fails the row test below)
    row = p / XW, col = p - row * XW;
  };
  auto t_draw = [&](const Tctx& t, int k, int g4, Tpass& q) __attribute__((always_inline)) {
    int p, row, col;
    t_pixel(k, p, row, col);
    const unsigned vecidx = (unsigned)(t.pix0 + row * W + col) * t.cs8 + t.co8;
    uint32_t rr[4];
    fi_rand32x4(t.seed, (size_t)vecidx * (VG / 4) + g4, rr);
    unsigned m = g4 ? q.keep : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) m |= rr[j] >= a.t0.thresh ? (1u << (g4 * 4 + j)) : 0u;
    q.keep = m;
  };
  auto t_apply = [&](const Tctx& t, int k, const Tpass& q) __attribute__((always_inline)) {
    int p, row, col;
    t_pixel(k, p, row, col);
    const bool ok = (unsigned)(row - t.rlo) < t.rspan && (unsigned)(col - t.clo) < t.cspan;   // outside the image: z = 0 (the DMA wrote zeros)
    vec_t* const ptr = reinterpret_cast<vec_t*>(smem + t.xb + (unsigned)((p * NP + (t.piece ^ ((col >> 1) & (NP - 1)))) * 16));
    if (ok) {
      const vec_t raw = *ptr;
      const float4* const tb = reinterpret_cast<const float4*>(smem + t.tb);
      const float4 s0 = tb[0], s1 = tb[1], h0 = tb[4], h1 = tb[5];
      const float sc[VG] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[VG] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
      float f[VG];
      VecWords<T>::unpack(raw, f);
#pragma unroll
      for (int j = 0; j < VG; ++j) {
        const float v = f[j] * sc[j] + sh[j];
        f[j] = fmaxf(v, v * t.slope);                            // = v > 0 ? v : v * slope for 0 <= slope <= 1 (host-checked)
      }
      if (t.drop) {
#pragma unroll
        for (int j = 0; j < VG; ++j) f[j] *= (q.keep >> j) & 1u ? a.t0.keep_scale : 0.f;
      }
      *ptr = VecWords<T>::pack(f);
    }
  };
  // (whole transform of one chunk at once: the prologue's chunk 0)
  auto transform = [&](const Item& it, int c_, int gbuf) __attribute__((always_inline)) {
    const Tctx t = t_setup(it, c_, gbuf, true);
    if (!t.on) return;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k == 1 && wv >= 4) break;                              // (uniform) pixels 256 + (wv >> 1) * 64 ..: beyond the list
      Tpass q;
      q.keep = 0xffu;
      if (t.drop) {
        t_draw(t, k, 0, q);
        t_draw(t, k, 1, q);
      }
      t_apply(t, k, q);
    }
  };

  // =============================================================================================== consumer side (conv_fwd_ws2_kernel's)
  const int n32 = lane & 31, hh = lane >> 5;
  const int prow = n32 >> 4, pcol = n32 & 15;
  const int rowbase = rg * 4, cobase = cgp * 64;
  unsigned pre[3];
#pragma unroll
  for (int sx = 0; sx < 3; ++sx)
    pre[sx] = (unsigned)(st * XS + (((rowbase + prow) * XW + pcol + sx) * NP + (hh ^ (((pcol + sx) >> 1) & (NP - 1)))) * 16);
  const unsigned wb = (unsigned)(O_W + (cobase + n32) * WROW + (((hh ^ (n32 >> 3)) & 1) << 4));
  const __amdgpu_buffer_rsrc_t ry0 = __builtin_amdgcn_make_buffer_rsrc(
      a.y0, 0, a.y0 ? (unsigned)a.N * hw * (unsigned)a.co0 * esz : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry1 = __builtin_amdgcn_make_buffer_rsrc(
      a.y1, 0, (a.y1 && a.co1) ? (unsigned)a.N * hw * (unsigned)a.co1 * esz : 0u, 0x00020000);

  f32x16 acc[2][NCB];                                            // [pixel pair: rows 0-1 / 2-3 of the wave][32-channel block]

  // a filter tap of the stage's chunk: 4 fragment reads (single-buffered: three other waves of the SIMD cover the LDS latency) + 4 MFMAs
  unsigned px[3];
  unsigned swb;
  auto taps_setup = [&](int wbuf, int gbuf, int cq) __attribute__((always_inline)) {
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) px[sx] = (pre[sx] ^ (unsigned)(cq << 5)) + (unsigned)(gbuf * XG);
    swb = wb + (unsigned)(wbuf * WT);
  };
  auto tap = [&](int t) __attribute__((always_inline)) {
    const int r = t / 3, sx = t % 3;
    frag_t P[2], Wf[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) Wf[cb] = *reinterpret_cast<const frag_t*>(smem + swb + cb * (32 * WROW) + t * 32);
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) P[pp] = *reinterpret_cast<const frag_t*>(smem + px[sx] + (2 * pp + r) * (XW * NP * 16));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[pp][cb] = mfma32(Wf[cb], P[pp], acc[pp][cb]);
    __builtin_amdgcn_sched_barrier(0);
  };

  // one strip per (sub-tile, 64-channel group), shared by its four row-group waves: [sum 64][sum of squares 64][bias 64] floats
  float* const strip = reinterpret_cast<float*>(smem + O_S) + (st * CG + cgp) * 192;
  auto stats_clear = [&]() __attribute__((always_inline)) { strip[lane] = strip[64 + lane] = 0.f; };
  auto stats_flush = [&](int grp) __attribute__((always_inline)) {
    if (!a.stats) return;
    const int slot = ((int)blockIdx.x * (NS * CG) + st * CG + cgp) & (FI_STATS_SLOTS - 1);
    const int co = ct * BN + cobase + lane;
    if (co < cout) {
      double* const dst = &a.stats[(size_t)grp * a.stats_gstride + ((size_t)slot * cout + co) * 2];
      atomicAdd(dst, (double)strip[lane]);
      atomicAdd(dst + 1, (double)strip[64 + lane]);
    }
  };
  auto load_bias = [&]() __attribute__((always_inline)) {
    const int co = ct * BN + cobase + lane;
    strip[128 + lane] = (a.bias && co < cout) ? a.bias[co] : 0.f;
  };

#define FI_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
  auto epilogue = [&](const Item& it) __attribute__((always_inline)) {
    int l = lane;
    asm volatile("" : "+v"(l));                                  // (the lane constants of the epilogue are recomputed per tile, not kept across the run)
    const int n32 = l & 31, hh = l >> 5;
    const int prow = n32 >> 4, pcol = n32 & 15;
    const int gx = it.tx * 16 + pcol;
    const bool colok = gx < W;
    const bool b3 = (l & 8) != 0, b2 = (l & 4) != 0, b1 = (l & 2) != 0;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        float4 bv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) bv[u] = *reinterpret_cast<const float4*>(&strip[128 + cb * 32 + 8 * (2 * jp + u) + 4 * hh]);
        f2 S[4], Q[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) S[m] = Q[m] = f2{0.f, 0.f};
        const int cg = ct * BN + cobase + cb * 32 + 8 * (2 * jp + hh);        // the 8 channels this lane stores
        const bool second = cg >= a.co0;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const int gy = it.ty * 16 + rowbase + 2 * pp + prow;
          const bool okp = colok && gy < H;
          const float mk1 = okp ? 1.f : 0.f;                     // tile overhang does not count (and is not stored)
          const f2 mk = {mk1, mk1};
          v2u q[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int k0 = 4 * (2 * jp + u);
            f2 v01 = {acc[pp][cb][k0], acc[pp][cb][k0 + 1]}, v23 = {acc[pp][cb][k0 + 2], acc[pp][cb][k0 + 3]};
            v01 = (v01 + f2{bv[u].x, bv[u].y}) * mk;
            v23 = (v23 + f2{bv[u].z, bv[u].w}) * mk;
            float v[4] = {v01.x, v01.y, v23.x, v23.y};
            q[u] = __builtin_bit_cast(v2u, Quad<T>::pack(v));    // v := the values as stored
            v01 = f2{v[0], v[1]}, v23 = f2{v[2], v[3]};
            S[2 * u] += v01, S[2 * u + 1] += v23;
            Q[2 * u] += v01 * v01, Q[2 * u + 1] += v23 * v23;
          }
          if (a.y0) {
            const v2u lo = __builtin_amdgcn_permlane32_swap(q[0].x, q[1].x, false, false);
            const v2u hi = __builtin_amdgcn_permlane32_swap(q[0].y, q[1].y, false, false);
            const v4u out = {lo.x, hi.x, lo.y, hi.y};
            const bool live = okp && cg < cout;
            const unsigned pix = (unsigned)((it.n * H + gy) * W + gx);
            if (a.co1 == 0) {
              __builtin_amdgcn_raw_buffer_store_b128(out, ry0, live ? (pix * (unsigned)a.co0 + (unsigned)cg) * esz : OOB, 0, 0);
            } else {
              __builtin_amdgcn_raw_buffer_store_b128(out, ry0, (live && !second) ? (pix * (unsigned)a.co0 + (unsigned)cg) * esz : OOB, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(out, ry1, (live && second) ? (pix * (unsigned)a.co1 + (unsigned)(cg - a.co0)) * esz : OOB, 0, 0);
            }
          }
        }
        if (a.stats) {
          float Y[8], Z[4], Wv[2];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const v2u s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(S[i].x), __float_as_uint(S[i].y), false, false);
            const v2u s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(Q[i].x), __float_as_uint(Q[i].y), false, false);
            Y[i] = __uint_as_float(s1.x) + __uint_as_float(s1.y);
            Y[4 + i] = __uint_as_float(s2.x) + __uint_as_float(s2.y);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float keep = b3 ? Y[2 * i + 1] : Y[2 * i], give = b3 ? Y[2 * i] : Y[2 * i + 1];
            Z[i] = keep + FI_DPP_F(give, 0x140);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float keep = b2 ? Z[2 * i + 1] : Z[2 * i], give = b2 ? Z[2 * i] : Z[2 * i + 1];
            Wv[i] = keep + FI_DPP_F(give, 0x141);
          }
          const float keep = b1 ? Wv[1] : Wv[0], give = b1 ? Wv[0] : Wv[1];
          float U = keep + FI_DPP_F(give, 0x4E);
          U += FI_DPP_F(U, 0xB1);
          if ((l & 1) == 0)
            atomicAdd(&strip[(b1 ? 64 : 0) + cb * 32 + 8 * (2 * jp + (b2 ? 1 : 0)) + 4 * hh + (b3 ? 2 : 0) + prow], U);
        }
      }
    }
  };
#undef FI_DPP_F

  // End of a stage: this wave's DMA has landed -- but NOT necessarily the epilogue's stores, which are younger than the stage's DMA
  // instructions and complete in order behind them: a counted wait leaves them in flight across the barrier (they drain beside the next
  // stage's MFMAs; waiting for them put the whole chip's store burst -- 32 MB, every workgroup in the same stage -- on the critical path).
  const int nst_epi = a.y0 ? (a.co1 ? 16 : 8) : 0;               // stores an epilogue issues per wave
  auto stage_end = [&](int nst) __attribute__((always_inline)) {
    if (nst == 0)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (nst == 8)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    fi_lds_barrier();
  };

  // =============================================================================================== the run
  // `it` = the item being consumed (index im in this sub-tile's list: live iff < nmine), `nx` = the one after it.  The chunk a stage
  // transforms (s + 1) and the group it fetches (chunks s + 2, s + 3) belong to `nx` at the end of an item.
  Item it = item_at(t_begin + st), nx = item_next2(it);
  int im = 0;
  int sgrp = -1, since_flush = 0;

  // prologue: slab 0 + group 0 land (and the coefficients of chunks 0 and 1), chunk 0 is transformed
  const int nB = cnt >> 1;                                        // items of sub-tile B
  unsigned wnext = wbase0 + wstride;                             // slab of the next stage (chunk c + 1, wrapping to chunk 0)
  if (st == 0) {
    issue_w(wbase0, 0);
    issue_x(0, it, 0, 0);
    if (nB > 0) issue_x(1, item_step(it, 1), 0, 0);
  }
  if (nmine > 0 && wv == WPS - 1) {
    stage_coef(it, 0, 0);
    stage_coef(it, 1, 1);
  }
  stage_end(0);
  if (nmine > 0) transform(it, 0, 0);
  if (rg == 0) {                                                 // strip housekeeping of this (sub-tile, channel group)
    stats_clear();
    load_bias();
  }
  __syncthreads();

  int c = 0;                                                     // chunk of the stage
  for (int s = 0; s < nstage; ++s) {
    FI_T2(1);                                                    // stage start
    int nst = 0;
    const bool live = im < nmine;
    const bool tnx = c + 1 == nchunk;                            // the chunk to transform is chunk 0 of the next item
    const bool tlive = im + (tnx ? 1 : 0) < nmine;
    const int tc = tnx ? 0 : c + 1;
    const int tgb = ((s + 1) >> 1) & 1;                          // group buffer of chunk s + 1
    Item ti;
    ti.tx = tnx ? nx.tx : it.tx, ti.ty = tnx ? nx.ty : it.ty, ti.n = tnx ? nx.n : it.n, ti.grp = tnx ? nx.grp : it.grp, ti.nl = tnx ? nx.nl : it.nl;
    if (st == 0) {                                               // (sub-tile A's waves: see issue_w)
      if (s + 1 < nstage) issue_w(wnext, (s + 1) & 1);
      if ((s & 1) == 0) {                                        // the group of chunks s + 2, s + 3 of both sub-tiles
        const bool xnx = c + 2 == nchunk;
        const int imx = im + (xnx ? 1 : 0);
        Item xi;
        xi.tx = xnx ? nx.tx : it.tx, xi.ty = xnx ? nx.ty : it.ty, xi.n = xnx ? nx.n : it.n, xi.grp = 0, xi.nl = xnx ? nx.nl : it.nl;
        const int xcg = xnx ? 0 : (c + 2) / 2, xgb = ((s >> 1) + 1) & 1;
        if (imx < npair) issue_x(0, xi, xcg, xgb);
        if (imx < nB) issue_x(1, item_step(xi, 1), xcg, xgb);
      }
    }
    wnext = c + 2 == nchunk ? wbase0 : wnext + wstride;
    if (c == 0) {
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[pp][cb][i] = 0.f;
      if (rg == 0 && live) {                                     // (>= 1 barrier after the previous tile's adds, >= 1 before this tile's)
        if (it.grp != sgrp || since_flush >= FI_WS2_FLUSH_TILES) {
          if (sgrp >= 0) {
            stats_flush(sgrp);
            stats_clear();
          }
          sgrp = it.grp;
          since_flush = 0;
        }
        ++since_flush;
      }
    }
    FI_T2(2);                                                    // DMA issued
    // the taps of chunk s with the transform of chunk s + 1 in slices between them: pass 0 behind taps 0..2, pass 1 (waves 0..3 of the
    // sub-tile: pixels 256..323) behind taps 3..5
    const Tctx tx_ = t_setup(ti, tc, tgb, tlive);
    const bool tdrop = tx_.on && tx_.drop, t2 = tx_.on && wv < 4;
    Tpass q;
    taps_setup(s & 1, (s >> 1) & 1, c & 1);
#if FI_DMA_DEBUG & 1
#define FI_DMA_TAP(t) do { } while (0)
#else
#define FI_DMA_TAP(t) tap(t)
#endif
    q.keep = 0xffu;
    FI_DMA_TAP(0);
    if (tdrop) t_draw(tx_, 0, 0, q);
    __builtin_amdgcn_sched_barrier(0);
    FI_DMA_TAP(1);
    if (tdrop) t_draw(tx_, 0, 1, q);
    __builtin_amdgcn_sched_barrier(0);
    FI_DMA_TAP(2);
    if (tx_.on) t_apply(tx_, 0, q);
    __builtin_amdgcn_sched_barrier(0);
    FI_DMA_TAP(3);
    if (t2 && tdrop) t_draw(tx_, 1, 0, q);
    __builtin_amdgcn_sched_barrier(0);
    FI_DMA_TAP(4);
    if (t2 && tdrop) t_draw(tx_, 1, 1, q);
    __builtin_amdgcn_sched_barrier(0);
    FI_DMA_TAP(5);
    if (t2) t_apply(tx_, 1, q);
    __builtin_amdgcn_sched_barrier(0);
    FI_DMA_TAP(6);
    FI_DMA_TAP(7);
    FI_DMA_TAP(8);
#undef FI_DMA_TAP
    FI_T2(4);                                                    // MFMAs issued
    if (wv == WPS - 1) {                                         // coefficients of chunk s + 2 (transformed in the next stage)
      const bool cnx = c + 2 >= nchunk;
      if (im + (cnx ? 1 : 0) < nmine) {
        Item ci;
        ci.tx = 0, ci.ty = 0, ci.n = 0, ci.nl = 0, ci.grp = cnx ? nx.grp : it.grp;
        stage_coef(ci, cnx ? c + 2 - nchunk : c + 2, s & 1);
      }
    }
    if (++c == nchunk) {
      c = 0;
#if !(FI_DMA_DEBUG & 16)
      if (live) {
        epilogue(it);
        nst = nst_epi;
      }
#endif
      FI_T2(6);                                                  // epilogue issued
      ++im;
      it = nx;
      nx = item_next2(nx);
    }
    FI_T2(7);
    stage_end(nst);
  }
  FI_T2(1);
  if (rg == 0 && sgrp >= 0) stats_flush(sgrp);
}

static int launch_conv_fwd_dma_geometry(const ConvArgs& a, int wgs_per_cu, long* blocks) {
  const long ntile = (long)a.N * a.tilesX * a.tilesY;
  long g1 = 256L * (wgs_per_cu > 0 ? wgs_per_cu : 1) / a.nct;
  if (g1 < 1) g1 = 1;
  if (g1 > (ntile + 1) / 2) g1 = (ntile + 1) / 2;                // at least one tile pair per workgroup
  *blocks = g1 * a.nct;
  return 0;
}

template <typename T>
static int launch_conv_fwd_dma(const ConvArgs& a, int wgs_per_cu, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * 2 * 21 * 1024 + (size_t)2 * (128 * 9 * 16 * 2) + (size_t)2 * 2 * 192 * sizeof(float) + 2 * 2 * 128;
  const int cin = a.c0 + a.c1, cout = a.co0 + a.co1;
  if (cin % 32 || a.c0 % 32 || a.c1 % 32 || cout != a.nct * 128 || a.tilesY != (a.H + 15) / 16 || a.tilesX != (a.W + 15) / 16)
    return FI_ERR_UNSUPPORTED;
  long blocks = 0;
  launch_conv_fwd_dma_geometry(a, wgs_per_cu, &blocks);
  const dim3 g((unsigned)blocks), b(1024);
  if (a.xf == 0) {
    static const bool big = fi_allow_big_lds((const void*)conv_fwd_dma_kernel<T, 0>);
    (void)big;
    hipLaunchKernelGGL((conv_fwd_dma_kernel<T, 0>), g, b, lds, st, a);
  } else if (a.xf == 1) {
    static const bool big = fi_allow_big_lds((const void*)conv_fwd_dma_kernel<T, 1>);
    (void)big;
    hipLaunchKernelGGL((conv_fwd_dma_kernel<T, 1>), g, b, lds, st, a);
  } else {
    return FI_ERR_UNSUPPORTED;
  }
  FI_CHECK_LAUNCH();
  return 0;
}
