// Forward / dgrad kernel, fifth form: wave-specialised, 64 x 64 WAVE tiles on v_mfma_f32_32x32x16 (channel-rich 3x3 layers).
//
// Why (round 3; profiles/r02_q_ws_kprobe_pmc.txt, MI355X_MICROARCH.md "LDS"): conv_fwd_ws_kernel gives each consumer wave a
// 32-pixel x 64-channel tile: per contraction step 6 ds_read_b128 feed 8 MFMAs, i.e. 0.75 LDS-array cycles per MFMA cycle
// for the fragment reads alone, plus the producers' writes -- the LDS pipe, not the matrix pipe, is what a bare consumer
// saturates (51 % of the bf16 peak).  LDS bytes per FLOP fall with the wave tile: here a consumer wave owns 4 image rows
// (64 pixels) x 64 output channels = 2 x 2 accumulator blocks of v_mfma_f32_32x32x16, so a filter tap is 4 fragment reads
// for 4 MFMAs of 32 cycles each: 0.5 LDS cycles per MFMA cycle (reads + writes: 0.66 against 0.99), and half the MFMA
// instructions.  K = 16 per MFMA also makes a 16-channel chunk exactly one MFMA per tap (9 per stage, no K padding), which
// halves a stage's footprint: a 256-pixel x 128-channel (or 512 x 64) workgroup tile double-buffers in 94 KB (76 KB).
//
//   workgroup = 8 consumer waves + 2 teams of 4 producer waves (1024 threads, 128 registers), persistent over a run of items
//   tile      = TR x 16 pixels x BN channels: (TR, BN) = (16, 128) -> consumers 4 row groups x 2 channel halves,
//                                                        (32,  64) -> consumers 8 row groups x 1
//   stage     = one 16-channel chunk: halo tile [TR+2][18][16 ch] + weight slab [BN][9 taps][16 ch]
//   weights   = CHUNK-MAJOR packed operand [Cin/16][Cout][9][16] (fi_pack_weights modes 2 / 3): a stage's slab is one
//               contiguous BN x 288-byte block -- with the tap-major layout a 16-channel chunk touches 32 of every 128-byte
//               line, four times the L2 -> L1 traffic, which at 128 output channels is the whole 64 B/clk of the vector L1.
// LDS layouts (bank = (byte / 4) mod 64; a ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32):
//   pixels : 32 bytes each, no padding; the two 16-byte channel halves of a pixel are SWAPPED on odd halo rows.  A pixel
//            fragment is two image rows x 16 pixels with lane = half * 32 + row * 16 + column, so a 16-lane group reads 8
//            pixels of an even and 8 of an odd row, all of the same half: the swap puts them on disjoint banks.
//   weights: rows of 304 bytes (288 + 16): 19 is odd, so the 16 rows of a lane group start 16 distinct 16-byte units apart.
// Epilogue: D[channel][pixel] with lane = pixel (32 lanes) and 4 x 4 consecutive channels per lane and block; a
// v_permlane32_swap per word pairs the two half-waves' 4-channel groups into 16-byte stores (8 consecutive channels of one
// pixel); BatchNorm statistics of the values as stored: DPP sum over the 16 lanes of a row, one v_permlane16_swap + add per
// PAIR of values over the two rows, then a wave-private LDS strip that is flushed to the fp64 accumulators every 8 tiles.
// Same transforms (XF 0 / 1), rounding, statistics slots and group semantics as the other forms.
#pragma once
#include "conv_impl.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#ifndef FI_WS2_DEBUG
#define FI_WS2_DEBUG 0         // A/B builds: 1 no MFMAs, 2 no loads, 4 no LDS commit, 8 no transform, 16 no epilogue, 32 no statistics
#endif
#define FI_WS2_FLUSH_TILES 8   // tiles between two flushes of the fp32 statistics strip into the fp64 accumulators

// -DFI_TRACE builds (tools/ws2_trace.py): every wave logs up to FI_WS2_TRN (s_memtime << 4 | tag) events of its first stages
#ifdef FI_TRACE
#define FI_WS2_TRN 250
#define FI_T2(tag)                                                                                                   \
  do {                                                                                                               \
    if (a.trace && lane == 0 && tn < FI_WS2_TRN)                                                                     \
      a.trace[((size_t)blockIdx.x * 16 + wave) * 256 + (tn++)] = ((long long)__builtin_amdgcn_s_memtime() << 4) | (tag); \
  } while (0)
#else
#define FI_T2(tag) do { } while (0)
#endif

// WR = 1 ("weights resident"): layers whose WHOLE filter fits beside two pixel groups (Cin * Cout * 18 bytes <= ~74 KB: 64 -> 64,
// 64 -> 32, 32 -> 32 ...) load it once per workgroup run instead of one slab per stage -- at 32 / 64 output channels the
// slab of a stage is as many bytes as its pixels, and the producers' fill rate is what bounds these layers.  A package is
// then a pixel part only.  BN = 32 (one 32-channel block per consumer wave) exists for this form only.
// MODE = 2 ("slab inner", SI): a layer whose whole contraction is ONE pixel group (Cin == 64) and that has several 128-channel
// slabs (the auxiliary head, 64 -> 512) walks the slabs INSIDE a tile: the tile's pixels are staged and transformed once
// instead of once per slab (a quarter of the loader's vector work -- what bounds the kernel), the weight slabs stream as before.
// An item is then a tile; its stages are slab-major; the next tile's pixel parts ride in the packages of the tile's last NQ stages.
template <typename T, int TR, int BN, int XF, int MODE>
__global__ __launch_bounds__(1024, 1) void conv_fwd_ws2_kernel(ConvArgs a) {
  constexpr int WR = MODE == 1 ? 1 : 0;
  constexpr bool SI = MODE == 2;
  static_assert(sizeof(T) == 2, "16-bit storage");
  static_assert((TR == 16 && BN == 128 && WR == 0) || (TR == 32 && BN == 64 && !SI) || (TR == 32 && BN == 32 && WR == 1),
                "8 consumer waves of 4 rows x 64 (32) channels");
  static_assert(XF == 0 || XF == 1, "plain or transforming loader");
  constexpr int CW = 8, PT = 256;                                // consumer waves; threads of one producer team
  constexpr int RG = TR / 4, CG = BN >= 64 ? BN / 64 : 1;        // consumer grid: row groups x channel groups
  constexpr int CWC = BN / CG, NCB = CWC / 32;                   // channels / 32-channel accumulator blocks of a consumer wave
  constexpr int XH = TR + 2, XW = 18, KK = 9, CK = 16, VG = 8;
  constexpr int GC = TR == 16 ? 64 : 32;                         // channels of a pixel GROUP (what the loader fetches per pixel)
  constexpr int NQ = GC / CK, NP = GC / VG;                      // stages per group; 16-byte pieces per staged pixel
  constexpr int XG = XH * XW * NP * 16;                          // bytes of a group's halo tile
  constexpr int WROW = KK * CK * 2, WT = BN * WROW;              // bytes of a staged weight row (288, unpadded) / slab
  constexpr int O_W = 2 * XG;                                    // LDS map: [2] pixel groups, [2 | all] weight slabs, [CG] strips
  static_assert(O_W + 2 * WT + CG * 192 * 4 <= 160 * 1024, "LDS");
  typedef typename DT<T>::vec_t vec_t;
  typedef typename DT<T>::frag_t frag_t;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= CW;
  const int cin = a.c0 + a.c1, cout = a.co0 + a.co1;
  const int H = a.H, W = a.W;
#ifdef FI_TRACE
  int tn = 0;
#endif

  // ---- this workgroup's run of items; item = slab * ntile + tile (slab-major: a run keeps its slab and statistics group)
  const int ntile = a.N * a.tilesY * a.tilesX, tpi = a.tilesY * a.tilesX;
  const int nitem = SI ? ntile : ntile * a.nct;
  const int per = (nitem + (int)gridDim.x - 1) / (int)gridDim.x;
  const int i_begin = (int)blockIdx.x * per, i_end = min(nitem, i_begin + per);
  if (i_begin >= i_end) return;
  const int nchunk = cin / CK, ngrp = cin / GC;
  const int spt = SI ? a.nct * nchunk : nchunk;                  // stages of an item
  const int nstage = (i_end - i_begin) * spt, ngroups = (i_end - i_begin) * ngrp;
  const int o_strip = O_W + (WR ? nchunk : 2) * WT;

  constexpr unsigned esz = 2;
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

  // Every workgroup streams the SAME weight slabs at about the same time.  Rotating the order in which a tile walks its chunks
  // by the tile's position (-DFI_WS2_ROT: groups by rot / NQ, the chunks inside a group by rot % NQ), so that workgroups on
  // different tiles are on different slabs, was measured and is NOT a gain (tools/ws2_trace.py: plain 64^2 128->128 stage
  // 3 484 -> 3 884 cycles, fused 4 960 -> 5 032): the L2 likes the lock step.  Left in as an A/B switch.
#ifdef FI_WS2_ROT
  auto rot_of = [&](const Item& c) { return (c.tx + 3 * c.ty) % nchunk; };
#else
  auto rot_of = [&](const Item&) { return 0; };
#endif

  if (producer) {
    // =============================================================================================== producers
    // Two teams of 4 waves.  PACKAGE k = the weight slab of stage k + one of the NQ parts of the pixel group that stage
    // k + NQ - 1 falls into (so that a group is complete one stage before its first use); team g owns the packages of parity
    // g: while stage s is consumed, team (s & 1) issues the loads of package s + 2 and the other team transforms / commits
    // package s + 1 -- a load has a whole stage to land and the commit's vmcnt(0) is exact.  Loads are branch-free and fixed
    // in number per package (conv_fwd_ws_kernel).
    const int team = (wave - CW) >> 2;
    const int ptid = tid - CW * 64 - team * PT;
#ifdef FI_WS2_PRIO
    __builtin_amdgcn_s_setprio(FI_WS2_PRIO);                    // A/B: producers ahead of the (older) consumer waves at the issue port
#endif
    // pixels of a part: PLQ consecutive halo pixels x NP pieces; thread j takes vectors j, j + 256, ...: a wave covers whole
    // 128-byte lines of the source, and a thread's piece (= its 8 channels within the group) never changes
    constexpr int PLQ = XH * XW / NQ, XV = PLQ * NP, XPASS = (XV + PT - 1) / PT;
    static_assert(PLQ * NQ == XH * XW && PT % NP == 0, "whole parts");
    const int piece = ptid % NP;
    constexpr int WV = BN * KK * 2, WPASS = (WV + PT - 1) / PT;    // weight slab: WV consecutive 16-byte vectors
    // Every load goes through a buffer resource built from SCALARS: which source a group comes from and whether the package
    // is live at all are uniform, and a uniform condition inside a per-lane select is what hipcc turns into a BRANCH around
    // the load -- its (path-insensitive) wait-count bookkeeping then puts s_waitcnt vmcnt(0) between the loads of one package
    // and they go out one memory round trip at a time (tools/ws2_trace.py: 4 300 cycles to issue 12 loads).  A dead package
    // loads through a zero-length resource (the hardware returns zeros); only the per-lane conditions select the offset.
    auto rsrc = [&](const void* ptr, unsigned bytes) __attribute__((always_inline)) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
    };
    const unsigned wbytes = (unsigned)nchunk * (unsigned)a.wrows * (KK * CK * esz);
    const unsigned xbytes0 = (unsigned)((XF != 0 && a.bcast0) ? a.gimages : a.N) * hw * (unsigned)a.c0 * esz;
    const unsigned xbytes1 = (unsigned)a.N * hw * (unsigned)a.c1 * esz;

    struct Set {
      vec_t x[XPASS];
      vec_t w[WR ? 1 : WPASS];
      float sc[XF != 0 ? VG : 1], sh[XF != 0 ? VG : 1];         // this thread's 8 scale / shift values (its piece of the group)
      unsigned flags;           // bit p: pass p is inside the image; bit 16: source 0; bit 17: transform active
    };
    Set S;
    const bool drop0 = XF == 1 && a.t0.drop_mode == FI_DROP_RNG_ELEM;
    uint64_t seed_base = 0;
    if (drop0) {
      seed_base = a.t0.seed;
      if (a.t0.seed_offset) seed_base += 0xD1B54A32D192ED03ull * (uint64_t)(uint32_t)a.t0.seed_offset[0];
    }

    // ---- pixel part q of the group `cg` of item `it`.  A thread's vectors are DPL halo pixels apart from pass to pass: (row,
    // col) and the source offset advance by one of two per-part constants (no multiply, no division per vector).
    constexpr int DPL = PT / NP, DR = DPL / XW, DC = DPL % XW;   // pixel-list step of a pass = DR rows + DC columns
    auto uni = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
    auto issue_x = [&](const Item& it, int cg_, int q_, bool live_) __attribute__((always_inline)) {
#if FI_WS2_DEBUG & 2
      return;
#endif
      // everything the buffer resource is made of is UNIFORM; say so, or hipcc wraps every load in a waterfall loop
      const int cg = uni(cg_), q = uni(q_);
      const bool live = uni(live_ ? 1 : 0) != 0;
      const bool first = cg * GC < a.c0;                         // a group never straddles the two sources (host)
      const unsigned cs = (unsigned)(first ? a.c0 : a.c1);
      const unsigned co = (unsigned)(cg * GC - (first ? 0 : a.c0)) + (unsigned)(piece * VG);
      const unsigned long long xptr = (unsigned long long)(first ? a.x0 : a.x1);
      const __amdgpu_buffer_rsrc_t rx = rsrc(
          reinterpret_cast<const void*>((unsigned long long)(unsigned)uni((int)(unsigned)xptr) |
                                        ((unsigned long long)(unsigned)uni((int)(unsigned)(xptr >> 32)) << 32)),
          (unsigned)uni((int)(live ? (first ? xbytes0 : xbytes1) : 0u)));
      const int n = uni(it.n), ty = uni(it.ty), tx = uni(it.tx);
      const int grp = a.gimages > 0 ? n / a.gimages : 0;
      const int nl = n - grp * a.gimages;
      const int ns = (XF != 0 && first && a.bcast0) ? nl : n;
      unsigned flags = first ? 0x10000u : 0u;
      const int pl0 = q * PLQ + ptid / NP;
      int row = pl0 / XW, col = pl0 - (pl0 / XW) * XW;
      const unsigned pxb = cs * esz;                             // bytes of a source pixel
      // byte offset of (row, col): ((ns H + ty TR - 1 + row) W + tx 16 - 1 + col) * pxb + co * 2
      unsigned off = ((unsigned)((ns * H + ty * TR - 1 + row) * W + tx * 16 - 1 + col)) * pxb + co * esz;
      const unsigned inc_a = (unsigned)(DR * W + DC) * pxb, inc_b = (unsigned)((DR + 1) * W + DC - XW) * pxb;
#pragma unroll
      for (int p = 0; p < XPASS; ++p) {
        const int gy = ty * TR + row - 1, gx = tx * 16 + col - 1;
        const bool ok = ptid + p * PT < XV && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;      // per lane
        flags |= ok ? (1u << p) : 0u;
        S.x[p] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : OOB, 0, 0));
        const bool wrap = col + DC >= XW;
        row += wrap ? DR + 1 : DR;
        col += wrap ? DC - XW : DC;
        off += wrap ? inc_b : inc_a;
      }
      if constexpr (XF != 0) {
        const float* const scp = first ? a.t0.scale : a.t1.scale;
        const float* const shp = first ? a.t0.shift : a.t1.shift;
        const bool act = live && scp != nullptr;
        flags |= act ? 0x20000u : 0u;
        const unsigned long long sp = (unsigned long long)scp, hp = (unsigned long long)shp;
        const unsigned nrec = (unsigned)uni((int)(act ? 0x7FFFFFF0u : 0u));
        const __amdgpu_buffer_rsrc_t rs = rsrc(reinterpret_cast<const void*>((unsigned long long)(unsigned)uni((int)(unsigned)sp) |
                                                                             ((unsigned long long)(unsigned)uni((int)(unsigned)(sp >> 32)) << 32)), nrec);
        const __amdgpu_buffer_rsrc_t rh = rsrc(reinterpret_cast<const void*>((unsigned long long)(unsigned)uni((int)(unsigned)hp) |
                                                                             ((unsigned long long)(unsigned)uni((int)(unsigned)(hp >> 32)) << 32)), nrec);
        const unsigned cofs = ((unsigned)grp * cs + co) * 4u;
#pragma unroll
        for (int j = 0; j < VG; j += 4) {
          const float4 s4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, cofs + j * 4, 0, 0));
          const float4 h4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rh, cofs + j * 4, 0, 0));
          S.sc[j] = s4.x, S.sc[j + 1] = s4.y, S.sc[j + 2] = s4.z, S.sc[j + 3] = s4.w;
          S.sh[j] = h4.x, S.sh[j + 1] = h4.y, S.sh[j + 2] = h4.z, S.sh[j + 3] = h4.w;
        }
      }
      S.flags = flags;
    };
    auto commit_x = [&](const Item& it, int cg_, int q_, int gbuf) __attribute__((always_inline)) {
#if FI_WS2_DEBUG & 4
      return;
#endif
      const int cg = uni(cg_), q = uni(q_);
      char* const xb = smem + gbuf * XG;
      const bool first = (S.flags & 0x10000u) != 0;
      float slope = 1.f;
      bool xf = false, drop = false;
      uint64_t seed = 0;
      unsigned cs8 = 0;
      unsigned vix = 0;
      const int pl0 = q * PLQ + ptid / NP;
      int pl = pl0, col = pl0 - (pl0 / XW) * XW;
      unsigned vinc_a = 0, vinc_b = 0;
      if constexpr (XF != 0) {
        slope = first ? a.t0.slope : a.t1.slope;
        xf = (S.flags & 0x20000u) != 0;
#if FI_WS2_DEBUG & 8
        xf = false;
#endif
        drop = drop0 && first;
        const int n = uni(it.n), ty = uni(it.ty), tx = uni(it.tx);
        const int grp = a.gimages > 0 ? n / a.gimages : 0;
        const int nl = n - grp * a.gimages;
        seed = seed_base + (uint64_t)grp * a.t0.seed_gstride;
        cs8 = (unsigned)(first ? a.c0 : a.c1) / VG;
        const unsigned co8 = (unsigned)(cg * GC - (first ? 0 : a.c0)) / VG + (unsigned)piece;
        const int row = pl0 / XW;
        // dropout element-vector index of (row, col) inside the group's tensor, stepped like the source offset
        vix = (unsigned)((nl * H + ty * TR - 1 + row) * W + tx * 16 - 1 + col) * cs8 + co8;
        vinc_a = (unsigned)(DR * W + DC) * cs8;
        vinc_b = (unsigned)((DR + 1) * W + DC - XW) * cs8;
      }
      auto xform = [&](const vec_t& raw, size_t vecidx) __attribute__((always_inline)) -> vec_t {
        float f[VG];
        VecWords<T>::unpack(raw, f);
#pragma unroll
        for (int j = 0; j < VG; ++j) {                 // (packed v_pk_fma_f32 / v_pk_mul_f32 measured 10 % SLOWER per launch)
          const float v = f[j] * S.sc[j] + S.sh[j];
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
        if (ptid + p * PT < XV) {
          const bool ok = (S.flags >> p) & 1u;                  // outside the image: z = 0, not act(shift)
          vec_t val;
          if constexpr (XF == 0)
            val = fi_vec_select(ok, S.x[p]);
          else
            val = fi_vec_select(ok, xf ? xform(S.x[p], vix) : S.x[p]);
          // piece slot within the pixel: xor by the column pair -> the consumers' fragment reads are conflict-free
          *reinterpret_cast<vec_t*>(xb + (pl * NP + (piece ^ ((col >> 1) & (NP - 1)))) * 16) = val;
        }
        const bool wrap = col + DC >= XW;
        pl += DPL;
        col += wrap ? DC - XW : DC;
        vix += wrap ? vinc_b : vinc_a;
      }
    };
    // ---- weight slab of chunk `chunk` (of the whole contraction) for the slab it.ct
    auto issue_w = [&](int ct, int chunk, bool live) __attribute__((always_inline)) {
#if FI_WS2_DEBUG & 2
      return;
#endif
      if constexpr (WR) return;
      // chunk-major operand: [chunk][wrows][9][16]: the slab is one contiguous block starting at row it.ct * BN
      const unsigned o0w = (((unsigned)chunk * (unsigned)a.wrows + (unsigned)(ct * BN)) * (KK * 2) + (unsigned)ptid) * 16u;
      const __amdgpu_buffer_rsrc_t rw = rsrc(a.w, (unsigned)__builtin_amdgcn_readfirstlane((int)(live ? wbytes : 0u)));
#pragma unroll
      for (int p = 0; p < WPASS; ++p) {
        const bool ok = (p + 1) * PT <= WV || ptid + p * PT < WV;          // compile-time true except in a ragged last pass
        S.w[p] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? o0w + (unsigned)(p * PT * 16) : OOB, 0, 0));
      }
    };
    auto commit_w = [&](int buf) __attribute__((always_inline)) {
#if FI_WS2_DEBUG & 4
      return;
#endif
      if constexpr (WR) return;
      char* const wb = smem + O_W + buf * WT;
#pragma unroll
      for (int p = 0; p < WPASS; ++p) {
        const int v = ptid + p * PT;
        if (v < WV) {
          const int co = v / (KK * 2), q18 = v - co * (KK * 2);
          // row of 288 bytes, no padding; the two 16-byte halves of a tap swap on rows 8..15 mod 16 (conflict-free reads)
          *reinterpret_cast<vec_t*>(wb + co * WROW + (q18 >> 1) * 32 + ((((q18 & 1) ^ (co >> 3)) & 1) << 4)) = S.w[p];
        }
      }
    };

    // package cursor: k = stage whose weights it carries; (gi, cgx, qx, itx) = pixel part of flat index k + NQ - 1
    // (slab-inner: ch counts the item's spt stages; the package of stage ch >= spt - NQ carries part ch - (spt - NQ) of the NEXT
    //  item's single group, the others carry weights only)
    Item itw = item_at(i_begin), itx = itw;
    int ch = 0, k = 0, gi = 0, cgx = 0, qx = NQ - 1, ti = 0;
    auto adv = [&]() __attribute__((always_inline)) {
      ++k;
      if (++ch == spt) {
        ch = 0;
        ++ti;
        itw = item_next(itw);
      }
      if constexpr (SI) return;
      if (++qx == NQ) {
        qx = 0;
        ++gi;
        if (++cgx == ngrp) {
          cgx = 0;
          itx = item_next(itx);
        }
      }
    };
    // position -> rotated index: group position gp of item it is group (gp + rot / NQ) % ngrp; stage position j is chunk
    // group(j / NQ) * NQ + (j % NQ + rot % NQ) % NQ
    auto grp_of = [&](const Item& it, int gp) __attribute__((always_inline)) {
      const int g = gp + rot_of(it) / NQ;
      return g >= ngrp ? g - ngrp : g;
    };
    auto chunk_of = [&](const Item& it, int j) __attribute__((always_inline)) {
      const int r = rot_of(it);
      int i = j % NQ + r % NQ;
      i = i >= NQ ? i - NQ : i;
      return grp_of(it, j / NQ) * NQ + i;
    };
    auto issue = [&]() __attribute__((always_inline)) {
      if constexpr (SI) {
        const int q = ch - (spt - NQ);
        issue_x(item_next(itw), 0, q >= 0 ? q : 0, q >= 0 && i_begin + ti + 1 < i_end);
        issue_w(ch / nchunk, ch % nchunk, k < nstage);
      } else {
        issue_x(itx, grp_of(itx, cgx), qx, gi < ngroups);
        issue_w(itw.ct, chunk_of(itw, ch), k < nstage);
      }
    };
    auto commit = [&]() __attribute__((always_inline)) {
      if constexpr (SI) {
        const int q = ch - (spt - NQ);
        if (q >= 0) commit_x(item_next(itw), 0, q, (ti + 1) & 1);   // (no next item: zeros into the idle buffer)
      } else {
        commit_x(itx, grp_of(itx, cgx), qx, gi & 1);
      }
      commit_w(k & 1);
    };
    if constexpr (WR) {
      // the whole filter, once: [chunk][BN rows][288 bytes], rows swizzled like a streamed slab (host: one slab, Cout == BN)
      const int pt = tid - CW * 64, total = nchunk * BN * (KK * 2);
      const __amdgpu_buffer_rsrc_t rw = rsrc(a.w, wbytes);
      for (int v0 = pt; v0 < total; v0 += 4 * 2 * PT) {
        vec_t t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int v = v0 + j * 2 * PT;
          const int chunk = v / (BN * KK * 2), r = v - chunk * (BN * KK * 2);
          const unsigned src = ((unsigned)chunk * (unsigned)a.wrows * (KK * 2) + (unsigned)r) * 16u;
          t[j] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rw, v < total ? src : OOB, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int v = v0 + j * 2 * PT;
          if (v < total) {
            const int chunk = v / (BN * KK * 2), r = v - chunk * (BN * KK * 2);
            const int co = r / (KK * 2), q18 = r - co * (KK * 2);
            *reinterpret_cast<vec_t*>(smem + O_W + chunk * WT + co * WROW + (q18 >> 1) * 32 + ((((q18 & 1) ^ (co >> 3)) & 1) << 4)) = t[j];
          }
        }
      }
    }
    if (team == 0) {
      if constexpr (SI) {
        issue_x(itw, 0, NQ - 1, true);
        issue_w(0, 0, true);
        commit_x(itw, 0, NQ - 1, 0);
        commit_w(0);
      } else {
        issue();
        commit();                                                // package 0: slab 0 + the LAST part of group 0
      }
      adv();
      adv();
    } else {
      for (int q = 0; q < NQ - 1; ++q) {                         // the other parts of group 0 (once per run)
        issue_x(itx, grp_of(itx, 0), q, true);
        commit_x(itx, grp_of(itx, 0), q, 0);
      }
      adv();
      issue();                                                   // package 1 in flight
    }
    fi_lds_barrier();
    for (int s = 0; s < nstage; ++s) {
      FI_T2(1);                                                  // iteration start
      if ((s & 1) == team) {
        issue();                                                 // k == s + 2
        FI_T2(2);                                                // loads issued
      } else {
        if (s + 1 < nstage) commit();                            // k == s + 1
        FI_T2(3);                                                // committed
        adv();
        adv();
      }
      fi_lds_barrier();
    }
    FI_T2(1);
  } else {
    // =============================================================================================== consumers
    const int rg = wave % RG, cgp = wave / RG;                   // row group, 64-channel group of this wave
    const int n32 = lane & 31, hh = lane >> 5;
    const int prow = n32 >> 4, pcol = n32 & 15;
    const int rowbase = rg * 4, cobase = cgp * CWC;
    // fragment addresses: lane constants (+ the chunk's xor, + the group buffer) + immediates.
    //   pixel (row, col), piece k = 2 * chunk + half:  ((row * 18 + col) * NP + (k ^ ((col >> 1) & (NP - 1)))) * 16
    //   weight row co, tap t, half h:                  co * 288 + t * 32 + ((h ^ (co >> 3)) & 1) * 16
    unsigned pre[3];
#pragma unroll
    for (int sx = 0; sx < 3; ++sx)
      pre[sx] = (unsigned)((((rowbase + prow) * XW + pcol + sx) * NP + (hh ^ (((pcol + sx) >> 1) & (NP - 1)))) * 16);
    const unsigned wb = (unsigned)(O_W + (cobase + n32) * WROW + (((hh ^ (n32 >> 3)) & 1) << 4));
    const __amdgpu_buffer_rsrc_t ry0 = __builtin_amdgcn_make_buffer_rsrc(
        a.y0, 0, a.y0 ? (unsigned)a.N * hw * (unsigned)a.co0 * esz : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry1 = __builtin_amdgcn_make_buffer_rsrc(
        a.y1, 0, (a.y1 && a.co1) ? (unsigned)a.N * hw * (unsigned)a.co1 * esz : 0u, 0x00020000);

    f32x16 acc[2][NCB];                                          // [pixel pair: rows 0-1 / 2-3 of the wave][32-channel block]

    auto mma = [&](int wbuf, int gbuf, int cq) __attribute__((always_inline)) {
      const char* const sw = smem + wbuf * WT;
      unsigned px[3];
#pragma unroll
      for (int sx = 0; sx < 3; ++sx) px[sx] = (pre[sx] ^ (unsigned)(cq << 5)) + (unsigned)(gbuf * XG);
      frag_t P[2][2], Wf[2][NCB];
      auto fetch = [&](int t, int q) __attribute__((always_inline)) {
        const int r = t / 3, sx = t % 3;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) Wf[q][cb] = *reinterpret_cast<const frag_t*>(sw + wb + cb * (32 * WROW) + t * 32);
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
          P[q][pp] = *reinterpret_cast<const frag_t*>(smem + px[sx] + (2 * pp + r) * (XW * NP * 16));
      };
      fetch(0, 0);
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        if (t + 1 < KK) fetch(t + 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[pp][cb] = mfma32(Wf[t & 1][cb], P[t & 1][pp], acc[pp][cb]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // one strip per 64-channel group, shared by its row-group waves: [sum 64][sum of squares 64][bias 64] floats.  The waves
    // ADD their per-tile totals (ds_add_f32); the row-group-0 wave clears / flushes it and loads the bias at the START of a
    // tile -- at least one stage barrier away from the adds on either side (a tile has >= 2 stages).
    float* const strip0 = reinterpret_cast<float*>(smem + o_strip) + cgp * 192;     // slab-inner: + slab * CG * 192
    float* strip = strip0;
    auto stats_clear = [&]() __attribute__((always_inline)) { strip[lane] = strip[64 + lane] = 0.f; };
    auto stats_flush = [&](int grp, int ct) __attribute__((always_inline)) {
      if (!a.stats) return;
      const int slot = (blockIdx.x * CG + cgp) & (FI_STATS_SLOTS - 1);
      const int co = ct * BN + cobase + lane;
      if (lane < CWC && co < cout) {
        double* const dst = &a.stats[(size_t)grp * a.stats_gstride + ((size_t)slot * cout + co) * 2];
        atomicAdd(dst, (double)strip[lane]);
        atomicAdd(dst + 1, (double)strip[64 + lane]);
      }
    };
    auto load_bias = [&](int ct) __attribute__((always_inline)) {
      const int co = ct * BN + cobase + lane;
      strip[128 + lane] = (a.bias && lane < CWC && co < cout) ? a.bias[co] : 0.f;
    };

    // Epilogue of a tile.  Per 32-channel block a lane holds, for its pixel column, 16 channels x 2 pixel rows (pp): bias,
    // rounding and the per-lane statistics partials are PACKED fp32 arithmetic (v_pk_add / v_pk_mul / v_pk_fma on register
    // pairs); the 16 sums + 16 sums of squares are then reduced over the 32 lanes of the half-wave by a TRANSPOSING
    // butterfly -- each level pairs two values, a lane keeps one of the pair and adds its partner's copy of it
    // (v_permlane16_swap for the row pair, then DPP row mirror / half-row mirror / quad xor 2 / quad xor 1), so the value
    // count halves as the lane sets double: per block 16 swaps + ~60 adds / selects instead of 128 DPP adds, and two
    // ds_add_f32 of 32 lanes instead of 16 of four.  (Half a block -- 8 registers -- at a time: the whole block's 32 partials
    // beside the accumulators do not fit 128 registers.)  (tools/ws2_trace.py: the epilogue was 3 800 cycles of a 10 900-cycle tile on the
    // 32 -> 32 layers, 7 300 of 16 800 at 64 -> 64.)
#define FI_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
    auto epilogue = [&](const Item& it) __attribute__((always_inline)) {
      const int gx = it.tx * 16 + pcol;
      const bool colok = gx < W;
      const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          // the two 4-channel groups j = 2 jp + u of this lane: channels cobase + cb * 32 + 8 j + 4 hh + (0..3), registers 4 j + r
          float4 bv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) bv[u] = *reinterpret_cast<const float4*>(&strip[128 + cb * 32 + 8 * (2 * jp + u) + 4 * hh]);
          f2 S[4], Q[4];                                         // [register pair 2 u + r / 2]: sums over the wave's pixel rows
#pragma unroll
          for (int m = 0; m < 4; ++m) S[m] = Q[m] = f2{0.f, 0.f};
          const int cg = it.ct * BN + cobase + cb * 32 + 8 * (2 * jp + hh);     // the 8 channels this lane stores
          const bool second = cg >= a.co0;
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            const int gy = it.ty * TR + rowbase + 2 * pp + prow;
            const bool okp = colok && gy < H;
            const float mk1 = okp ? 1.f : 0.f;                   // tile overhang does not count (and is not stored)
            const f2 mk = {mk1, mk1};
            v2u q[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int k0 = 4 * (2 * jp + u);
              f2 v01 = {acc[pp][cb][k0], acc[pp][cb][k0 + 1]}, v23 = {acc[pp][cb][k0 + 2], acc[pp][cb][k0 + 3]};
              v01 = (v01 + f2{bv[u].x, bv[u].y}) * mk;
              v23 = (v23 + f2{bv[u].z, bv[u].w}) * mk;
              float v[4] = {v01.x, v01.y, v23.x, v23.y};
              q[u] = __builtin_bit_cast(v2u, Quad<T>::pack(v));  // v := the values as stored
#if !(FI_WS2_DEBUG & 32)
              v01 = f2{v[0], v[1]}, v23 = f2{v[2], v[3]};
              S[2 * u] += v01, S[2 * u + 1] += v23;
              Q[2 * u] += v01 * v01, Q[2 * u + 1] += v23 * v23;
#endif
            }
            if (a.y0) {
              // lanes 0-31 end up with group 2 jp of both half-waves (8 consecutive channels), lanes 32-63 with group 2 jp + 1
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
#if !(FI_WS2_DEBUG & 32)
          if (a.stats) {
            // value list X = [S of register 0..7 | Q of register 0..7] (register = 4 u + r); level by level a lane keeps
            // X[2 i + its class bit]
            float Y[8], Z[4], Wv[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                        // rows of the half-wave (prow): even row <- .x, odd row <- .y
              const v2u s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(S[i].x), __float_as_uint(S[i].y), false, false);
              const v2u s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(Q[i].x), __float_as_uint(Q[i].y), false, false);
              Y[i] = __uint_as_float(s1.x) + __uint_as_float(s1.y);
              Y[4 + i] = __uint_as_float(s2.x) + __uint_as_float(s2.y);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                        // lane ^ 15 (row mirror), class = bit 3
              const float keep = b3 ? Y[2 * i + 1] : Y[2 * i], give = b3 ? Y[2 * i] : Y[2 * i + 1];
              Z[i] = keep + FI_DPP_F(give, 0x140);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {                        // lane ^ 7 (half-row mirror), class = bit 2
              const float keep = b2 ? Z[2 * i + 1] : Z[2 * i], give = b2 ? Z[2 * i] : Z[2 * i + 1];
              Wv[i] = keep + FI_DPP_F(give, 0x141);
            }
            const float keep = b1 ? Wv[1] : Wv[0], give = b1 ? Wv[0] : Wv[1];  // lane ^ 2, class = bit 1
            float U = keep + FI_DPP_F(give, 0x4E);
            U += FI_DPP_F(U, 0xB1);                              // lane ^ 1: both lanes of the pair hold the total
            // ... of X[8 b1 + 4 b2 + 2 b3 + prow]: b1 = sum | sum of squares, register 4 u + r = 4 b2 + 2 b3 + prow
            if ((lane & 1) == 0)
              atomicAdd(&strip[(b1 ? 64 : 0) + cb * 32 + 8 * (2 * jp + (b2 ? 1 : 0)) + 4 * hh + (b3 ? 2 : 0) + prow], U);
          }
#endif
        }
      }
    };
#undef FI_DPP_F

    Item it = item_at(i_begin);
    int ch = 0, cq = 0, gpar = 0, since_flush = 0;
    int sgrp = -1, sct = -1;
    auto consume = [&](int wbuf) __attribute__((always_inline)) {
      FI_T2(4);                                                  // stage start (barrier passed)
      const int cpos = SI ? ch % nchunk : ch;                    // chunk position within the slab's contraction
      if constexpr (SI) {
        it.ct = ch / nchunk;
        strip = strip0 + it.ct * (CG * 192);
      }
      if (cpos == 0) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[pp][cb][i] = 0.f;
      }
      if (ch == 0 && rg == 0) {                                  // strip housekeeping for this 64-channel group
        const int grp = a.gimages > 0 ? it.n / a.gimages : 0;
        if constexpr (SI) {
          if (grp != sgrp || since_flush >= FI_WS2_FLUSH_TILES) {   // new group, or time to leave fp32: every slab's strip
            for (int ct = 0; ct < a.nct; ++ct) {
              strip = strip0 + ct * (CG * 192);
              if (sgrp >= 0) stats_flush(sgrp, ct);
              stats_clear();
              if (sct < 0) load_bias(ct);
            }
            strip = strip0;
            sgrp = grp;
            sct = 0;
            since_flush = 0;
          }
        } else if (grp != sgrp || it.ct != sct || since_flush >= FI_WS2_FLUSH_TILES) {   // new (group, slab), or time to leave fp32
          if (sct >= 0) stats_flush(sgrp, sct);
          stats_clear();
          if (it.ct != sct) load_bias(it.ct);
          sgrp = grp;
          sct = it.ct;
          since_flush = 0;
        }
        ++since_flush;
      }
#if !(FI_WS2_DEBUG & 1)
      {
        int cr = cq + rot_of(it) % NQ;                           // the chunk of its group this stage holds (producers: chunk_of)
        cr = cr >= NQ ? cr - NQ : cr;
        mma(WR ? ch : wbuf, gpar, SI ? cpos : cr);
      }
#endif
      FI_T2(5);                                                  // MFMAs issued
      if constexpr (!SI) {
        if (++cq == NQ) {
          cq = 0;
          gpar ^= 1;
        }
      }
      if (cpos == nchunk - 1) {
#if !(FI_WS2_DEBUG & 16)
        epilogue(it);
#endif
        FI_T2(6);                                                // epilogue issued
      }
      if (++ch == spt) {
        ch = 0;
        if constexpr (SI) gpar ^= 1;
        it = item_next(it);
      }
    };
    fi_lds_barrier();                                            // package 0 and group 0 are in LDS
    for (int s = 0; s < nstage; s += 2) {
      consume(0);
      fi_lds_barrier();
      if (s + 1 >= nstage) break;
      consume(1);
      fi_lds_barrier();
    }
    if (rg == 0 && sct >= 0) {
      if constexpr (SI) {
        for (int ct = 0; ct < a.nct; ++ct) {
          strip = strip0 + ct * (CG * 192);
          stats_flush(sgrp, ct);
        }
      } else {
        stats_flush(sgrp, sct);
      }
    }
    FI_T2(4);
  }
}

template <typename T, int TR, int BN, int MODE>
static int launch_conv_fwd_ws2(const ConvArgs& a, int wgs_per_cu, hipStream_t st) {
  constexpr int XH = TR + 2, XW = 18, GC = TR == 16 ? 64 : 32;
  constexpr bool WR = MODE == 1, SI = MODE == 2;
  const int nchunk = (a.c0 + a.c1) / 16;
  const size_t lds = (size_t)2 * (XH * XW * GC * 2) + (size_t)(WR ? nchunk : 2) * (BN * 9 * 16 * 2) +
                     (size_t)(SI ? a.nct : 1) * (BN >= 64 ? BN / 64 : 1) * 192 * sizeof(float);
  if (lds > 160 * 1024 || (WR && (a.nct != 1 || a.co0 + a.co1 != BN)) || (SI && a.c0 + a.c1 != GC)) return FI_ERR_UNSUPPORTED;
  const long nitem = (long)a.N * a.tilesX * a.tilesY * (SI ? 1 : a.nct);
  long blocks = 256L * (wgs_per_cu > 0 ? wgs_per_cu : 1);
  if (blocks > nitem) blocks = nitem;
  const dim3 g((unsigned)blocks), b(1024);
  if (a.xf == 0) {
    static const bool big = fi_allow_big_lds((const void*)conv_fwd_ws2_kernel<T, TR, BN, 0, MODE>);
    (void)big;
    hipLaunchKernelGGL((conv_fwd_ws2_kernel<T, TR, BN, 0, MODE>), g, b, lds, st, a);
  } else if (a.xf == 1) {
    static const bool big = fi_allow_big_lds((const void*)conv_fwd_ws2_kernel<T, TR, BN, 1, MODE>);
    (void)big;
    hipLaunchKernelGGL((conv_fwd_ws2_kernel<T, TR, BN, 1, MODE>), g, b, lds, st, a);
  } else {
    return FI_ERR_UNSUPPORTED;
  }
  FI_CHECK_LAUNCH();
  return 0;
}
