// 3x3x3 convolution of the THIN full-resolution layers of unet_3D, streamed along the depth axis (gfx950 / CDNA4 only).
//
// /root/reference/code/networks/unet_3D.py:20-94 with utils.py:99-123 (UnetConv3) and :260-276 (UnetUp3_CT): at 128^3 the network runs
// 16 -> 16 twice, (16 skip + 32 up-sampled) -> 16 once, and backwards their input gradients 16 -> 16 and 16 -> (16 + 32) -- 2 x 128^3 x 16
// channels in or out, 67 MB a side in bf16, 34 / 70 us at the HBM roofline.  The one-launch form of rounds 3-5 (conv_fwd_ws_kernel with
// the depth taps as channel groups of the contraction) stages the three input slices of every output slice again for every output slice
// and fills half of a 32-channel MFMA slab: 220 / 447 us forward, 188 / 346 us dgrad -- 0.15-0.20 of the roofline, 1.6 of configs[3]'s
// 6.1 ms per iteration (profiles/r05_z_c4_per_layer_roofline.txt).
//
// Here a workgroup owns a (4 MF) x 16 (y, x) tile of ONE volume and walks a run of consecutive output slices; a RING of NR halo slices
// ((4 MF + 2) x 18 pixels x Cin) lives in LDS, so every input slice is staged once per tile (1.27 x its bytes for the halo, + 2 slices per run):
//   * staging by LDS-DMA (buffer_load_dwordx4 ... lds; conv_dma.h): zero registers, out-of-volume lanes fetch through an out-of-range
//     offset (the hardware writes zeros); the 64-byte pixels of the 32-channel source are swizzled on the SOURCE side (piece ^ ((col >> 1)
//     & 3)) so that the MFMA operand reads are bank-conflict-free; the 32-byte pixels of the 16-channel source need no swizzle;
//   * what bounds a workgroup is the LATENCY of its own staging chain, not bytes or flops (first version, ring of three: 100 us for
//     16 -> 16 where matrix pipe, LDS and HBM each ask for 30-37): the ring is as deep as LDS allows (NR = 5-8: two to five slices in
//     flight while one is computed, drained by COUNTED s_waitcnt vmcnt -- every wave issues the same number of DMA instructions per
//     slice); one slot is always free, so the next slice is issued at the top of an iteration and there is ONE barrier per slice;
//   * the whole filter sits in LDS as ready-made MFMA A fragments ([fragment][k-step][lane][16 bytes]: linear, conflict-free), built once
//     per workgroup from the library's own operand [Cout][9][3][Cin] (fi_pack_weights3d_multi) -- the contraction is re-ordered to
//     [depth tap][source][in-plane tap][channel] so that a 32-deep k-step never straddles a phase or a source;
//   * 4 waves x 4 rows x 16 pixels, v_mfma_f32_16x16x32: D[channel][pixel], a lane ends up with 4 consecutive channels of one pixel;
//     epilogue = conv_thin_kernel's (bias, round, per-lane statistics partials over the whole run, v_permlane16_swap -> 16-byte stores,
//     one destination or two).  The stores stay in flight across the barrier behind a counted s_waitcnt vmcnt.
// Statistics are per volume (InstanceNorm3d: utils.py:104-110), 8 fp64 slots per (volume, channel) like every other kernel.
#include <cstdlib>

#include "conv_dma.h"

namespace {

struct S3Args {
  const void* x0;
  const void* x1;
  const void* w;           // [Cout][9][3][c0 + c1]
  const float* bias;
  void* y0;
  void* y1;
  double* stats;
  long stats_stride;       // doubles between the accumulators of consecutive volumes
  int N, D, H, W;
  int co0, co1;
  int tilesX, tilesY, nseg, dseg;
};

// C1: channels of the second source (0 or 32; the first has 16); NF: 16-channel output fragments (1 or 3); NWV waves of MF tile rows each
// (the tile is NWV MF x 16 pixels); NR: slices of the LDS ring
template <typename T, int C1, int NF, int MF, int NR, int NWV, bool WREG>
__global__ __launch_bounds__(NWV * 64, 1) void conv3d_stream_kernel(S3Args a) {
  static_assert(sizeof(T) == 2, "16-bit storage");
  static_assert(C1 == 0 || C1 == 32, "second source: none or 32 channels");
  static_assert(NR >= 2 && (MF == 2 || MF == 4) && (NWV == 4 || NWV == 8), "ring of >= 2 slices; 2 or 4 rows per wave; 4 or 8 waves");
  static_assert(!WREG || NF == 1, "the filter fits the registers for 16 outputs only (60 / 168 registers)");
  constexpr int C0 = 16, CIN = C0 + C1;
  constexpr int TH = NWV * MF, NT = NWV * 64, XW = 18, XH = TH + 2, NPX = XW * XH;
  constexpr int KS0 = 5, KS1 = C1 ? 9 : 0, KSZ = KS0 + KS1;          // k-steps of one depth tap: 16-channel source (2 taps each) + 32-channel source
  constexpr int NKS = 3 * KSZ;
  // DMA instructions (1 KB each) of a slice: issued by the first NIW = 4 waves, the same number each (the surplus fetch out of range: zeros)
  constexpr int NIW = 4;
  constexpr int P0W = ((NPX * 2 + 63) / 64 + NIW - 1) / NIW, P1W = C1 ? ((NPX * 4 + 63) / 64 + NIW - 1) / NIW : 0;
  constexpr int NW = P0W + P1W;                                      // per issuing wave and slice: what the counted waits count
  constexpr int P0B = NIW * P0W * 1024, SLOT = P0B + NIW * P1W * 1024;   // bytes of the 16-channel plane / of a ring slot
  constexpr int O_W = NR * SLOT, WB = WREG ? 0 : NF * NKS * 1024;    // filter fragments behind the ring (or in registers)
  constexpr int O_R = O_W + WB;                                      // [NWV waves][NF * 16][2] floats: the statistics fold
  static_assert(O_R + NWV * NF * 16 * 2 * 4 <= 160 * 1024, "LDS");
  static_assert((NR - 2) * NW + 4 * NF <= 63, "the vector-memory counter has 6 bits");
  typedef typename DT<T>::vec_t vec_t;
  typedef typename DT<T>::frag_t frag_t;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) char* lds_cptr_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_cptr_t)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kg = lane >> 4;
  const int D = a.D, H = a.H, W = a.W;
  const int cout = a.co0 + a.co1;
  constexpr unsigned esz = 2, OOB = 0xFFFFFFF0u;

  // ---- this workgroup's run: volume n, tile (ty, tx), output slices [dlo, dhi).  Workgroup ids are dealt round-robin to the 8 XCDs (each
  // with its own L2): they are renumbered so that one XCD walks a CONTIGUOUS range of (tile, run) pairs -- neighbouring tiles re-read
  // each other's halo columns / rows (27 % of a tile's bytes) and neighbouring runs two slices, which then hit that XCD's L2
  int b = (int)blockIdx.x;
  {
    const int nb = (int)gridDim.x, per = nb / 8;
    if (b < per * 8) b = (b & 7) * per + (b >> 3);               // (the nb % 8 trailing ids keep their number)
  }
  const int seg = b % a.nseg;
  b /= a.nseg;
  const int tx = b % a.tilesX;
  b /= a.tilesX;
  const int ty = b % a.tilesY, n = b / a.tilesY;
  const int dlo = seg * a.dseg, dhi = min(D, dlo + a.dseg);
  if (dlo >= dhi) return;
  const int y0 = ty * TH - 1, x0 = tx * 16 - 1;

  // ---- the filter as MFMA A fragments.  K order: depth tap kz, then [5 k-steps of the 16-channel source: in-plane taps (2 j, 2 j + 1),
  // the tenth is zero] [9 k-steps of the 32-channel source: one in-plane tap each]; lane (li = output channel, kg) holds 8 consecutive
  // channels.  In LDS ([fragment][k-step][lane][16 bytes]) -- or, 16 -> 16, in 60 registers for the life of the workgroup.
  auto wfrag = [&](int f, int ks, int l) __attribute__((always_inline)) {
    const T* wg = reinterpret_cast<const T*>(a.w);
    const int co = f * 16 + (l & 15), g = l >> 4;
    const int kz = ks / KSZ, r = ks - kz * KSZ;
    int tap, c;
    if (r < KS0) {
      tap = 2 * r + (g >> 1);
      c = (g & 1) * 8;
    } else {
      tap = r - KS0;
      c = C0 + g * 8;
    }
    vec_t val = make_uint4(0u, 0u, 0u, 0u);
    if (tap < 9 && co < cout) val = *reinterpret_cast<const vec_t*>(wg + ((size_t)(co * 9 + tap) * 3 + kz) * CIN + c);
    return val;
  };
  frag_t wreg[WREG ? NKS : 1];
  if constexpr (WREG) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) wreg[ks] = __builtin_bit_cast(frag_t, wfrag(0, ks, lane));
  } else {
    for (int v = tid; v < NF * NKS * 64; v += NT)
      *reinterpret_cast<vec_t*>(smem + O_W + v * 16) = wfrag((v >> 6) / NKS, (v >> 6) % NKS, v & 63);
  }
  float bv[NF][4];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = f * 16 + kg * 4 + r;
      bv[f][r] = (a.bias && co < cout) ? a.bias[co] : 0.f;
    }

  // ---- staging: slice z of the volume (zeros outside it) into ring slot z mod NR; NW DMA instructions per wave
  const unsigned vox = (unsigned)D * (unsigned)H * (unsigned)W;
  const fi_v4i r0 = fi_raw_rsrc(a.x0, (unsigned)a.N * vox * C0 * esz);
  const fi_v4i r1 = fi_raw_rsrc(C1 ? a.x1 : a.x0, C1 ? (unsigned)a.N * vox * C1 * esz : 0u);
  const unsigned rlo = y0 < 0 ? 1u : 0u, clo = x0 < 0 ? 1u : 0u;
  const unsigned rspan = (unsigned)min(XH, H - y0) - rlo, cspan = (unsigned)min(XW, W - x0) - clo;
  auto slot_of = [&](int z) __attribute__((always_inline)) { return (z + NR) % NR; };    // (z >= -1)
  auto issue_slice = [&](int z) __attribute__((always_inline)) {
    if (NWV > NIW && wave >= NIW) return;                        // (uniform)
    const bool zin = (unsigned)z < (unsigned)D;
    const unsigned dst = lds0 + (unsigned)(slot_of(z) * SLOT);
    const unsigned pbase = (unsigned)(((n * D + z) * H + y0) * W + x0);      // (wraps for out-of-volume origins: only in-volume lanes use it)
    int l = lane;
    asm volatile("" : "+v"(l));                                  // (per-lane constants recomputed per slice, not kept in registers across the run)
#pragma unroll
    for (int k = 0; k < P0W; ++k) {                              // 16-channel source: 2 pieces per pixel
      const int i = wave + NIW * k;
      int p = i * 32 + (l >> 1);
      p = p < NPX ? p : NPX;                                     // (beyond the list: row XH -- fails the row test)
      const unsigned row = (unsigned)p / XW, col = (unsigned)p - row * XW;
      const bool ok = zin && row - rlo < rspan && col - clo < cspan;
      const unsigned off = ok ? (pbase + row * (unsigned)W + col) * (C0 * esz) + ((unsigned)l & 1u) * 16u : OOB;
      fi_lds_dma16(r0, dst + (unsigned)(i * 1024), off);
    }
    if constexpr (C1 != 0) {
#pragma unroll
      for (int k = 0; k < P1W; ++k) {                            // 32-channel source: 4 pieces per pixel, slot = piece ^ ((col >> 1) & 3)
        const int i = wave + NIW * k;
        int p = i * 16 + (l >> 2);
        p = p < NPX ? p : NPX;
        const unsigned row = (unsigned)p / XW, col = (unsigned)p - row * XW;
        const bool ok = zin && row - rlo < rspan && col - clo < cspan;
        const unsigned off = ok ? (pbase + row * (unsigned)W + col) * (C1 * esz) + ((((unsigned)l & 3u) ^ ((col >> 1) & 3u)) * 16u) : OOB;
        fi_lds_dma16(r1, dst + (unsigned)(P0B + i * 1024), off);
      }
    }
  };

  // ---- operand addresses.  Pixel fragment of tile row (wave * MF + m), in-plane tap (ky, kx): lane li = pixel column.
  //   16-channel plane: ((row + ky) * 18 + li + kx) * 32 + (kg & 1) * 16, the tap chosen by kg >> 1 (two taps per k-step)
  //   32-channel plane: ((row + ky) * 18 + li + kx) * 64 + (kg ^ (((li + kx) >> 1) & 3)) * 16
  const unsigned a0 = (unsigned)(((wave * MF) * XW + li) * 32 + (kg & 1) * 16);
  const unsigned hib = (unsigned)(kg >> 1);
  unsigned a1[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
    a1[kx] = (unsigned)(P0B + ((wave * MF) * XW + li + kx) * 64 + ((kg ^ (((li + kx) >> 1) & 3)) * 16));
  const unsigned wa = (unsigned)(O_W + lane * 16);

  // INPUT-stationary order (third version).  Input slice z feeds three output slices -- z + 1 through depth tap 0, z through tap 1,
  // z - 1 through tap 2 -- so a pixel fragment read from LDS once is multiplied by THREE filter fragments into three accumulator sets
  // (the output-stationary first versions read every fragment three times and were LDS-read-bound at 84-100 us for 16 -> 16: 1.5 reads
  // of 1 KB per 16-cycle MFMA).  acc[0]: output z - 1 (complete after this slice), acc[1]: output z, acc[2]: output z + 1.
  // The k-steps of a slice are software-pipelined: the fragments of k-step j + 1 are read while the MFMAs of k-step j run.
  f32x4 acc[3][MF][NF];
  auto slice_mma = [&](unsigned sb) __attribute__((always_inline)) {
    frag_t wf[2][3][NF], pv[2][MF];
    auto fetch = [&](int r, int q) __attribute__((always_inline)) {
      if constexpr (!WREG) {
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
          for (int f = 0; f < NF; ++f) wf[q][kz][f] = *reinterpret_cast<const frag_t*>(smem + wa + ((f * NKS + kz * KSZ + r) * 1024));
      }
      unsigned pa;
      if (r < KS0) {
        const int tA = 2 * r, tB = (2 * r + 1 < 9) ? 2 * r + 1 : 2 * r;    // (the tenth tap's filter fragment is zero: any valid address)
        const int oA = ((tA / 3) * XW + tA % 3) * 32, oB = ((tB / 3) * XW + tB % 3) * 32;
        pa = a0 + sb + (unsigned)oA + hib * (unsigned)(oB - oA);
      } else {
        const int t = r - KS0;
        pa = a1[t % 3] + sb + (unsigned)((t / 3) * XW * 64);
      }
      const int rowb = r < KS0 ? XW * 32 : XW * 64;
#pragma unroll
      for (int m = 0; m < MF; ++m) pv[q][m] = *reinterpret_cast<const frag_t*>(smem + pa + m * rowb);
    };
    fetch(0, 0);
#pragma unroll
    for (int r = 0; r < KSZ; ++r) {
      if (r + 1 < KSZ) fetch(r + 1, (r + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kz = 0; kz < 3; ++kz)
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            if constexpr (WREG)
              acc[2 - kz][m][f] = mfma16(wreg[kz * KSZ + r], pv[r & 1][m], acc[2 - kz][m][f]);
            else
              acc[2 - kz][m][f] = mfma16(wf[r & 1][kz][f], pv[r & 1][m], acc[2 - kz][m][f]);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue of one output slice (conv_thin_kernel's): statistics partials stay in registers over the run
  const __amdgpu_buffer_rsrc_t ry0 = __builtin_amdgcn_make_buffer_rsrc(a.y0, 0, (unsigned)a.N * vox * (unsigned)a.co0 * esz, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry1 = __builtin_amdgcn_make_buffer_rsrc(a.co1 ? a.y1 : a.y0, 0, a.co1 ? (unsigned)a.N * vox * (unsigned)a.co1 * esz : 0u,
                                                                      0x00020000);
  float ssum[NF][4], ssq[NF][4];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[f][r] = ssq[f][r] = 0.f;
  const int gx = tx * 16 + li;
  const bool colok = gx < W;
  auto epilogue = [&](int d) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int mp = 0; mp < MF; mp += 2) {
        v2u q[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = mp + h;
          const float mk = (colok && ty * TH + wave * MF + m < H) ? 1.f : 0.f;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[0][m][f][r] + bv[f][r];
          q[h] = __builtin_bit_cast(v2u, Quad<T>::pack(v));      // v := the values as stored
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float vm = v[r] * mk;                          // tile overhang does not count
            ssum[f][r] += vm;
            ssq[f][r] += vm * v[r];
          }
        }
        const v2u lo = __builtin_amdgcn_permlane16_swap(q[0].x, q[1].x, false, false);
        const v2u hi = __builtin_amdgcn_permlane16_swap(q[0].y, q[1].y, false, false);
        const v4u out = {lo.x, hi.x, lo.y, hi.y};                // channels cg .. cg + 7 of pixel row mp + (kg & 1)
        const int gy = ty * TH + wave * MF + mp + (kg & 1);
        const int cg = f * 16 + (kg >> 1) * 8;
        const bool live = colok && gy < H && cg < cout;
        const unsigned pix = (unsigned)(((n * D + d) * H + gy) * W + gx);
        if (a.co1 == 0) {
          __builtin_amdgcn_raw_buffer_store_b128(out, ry0, live ? (pix * (unsigned)a.co0 + (unsigned)cg) * esz : OOB, 0, 0);
        } else {
          const bool second = cg >= a.co0;
          __builtin_amdgcn_raw_buffer_store_b128(out, ry0, (live && !second) ? (pix * (unsigned)a.co0 + (unsigned)cg) * esz : OOB, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(out, ry1, (live && second) ? (pix * (unsigned)a.co1 + (unsigned)(cg - a.co0)) * esz : OOB, 0, 0);
        }
      }
    }
  };
  constexpr int NST = NF * (MF / 2);                             // stores of an epilogue per lane (one destination; twice that with two)

  // End of an output slice: slice d + 2 must have landed.  Memory operations complete in order, so "at most K outstanding" with K = the
  // DMA instructions of the KEEP younger slices + the epilogue's stores (youngest of all) waits for exactly the older ones.
  auto wait_keep = [&](int keep_) __attribute__((always_inline)) {
    int keep = keep_;
#define FI_S3_WAIT(K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory")
    const bool two = a.co1 != 0;
    if (NWV > NIW && wave >= NIW) keep = 0;                      // (a wave that issues no DMA has only its stores outstanding)
    if (keep == 0) {
      if (two) FI_S3_WAIT(2 * NST); else FI_S3_WAIT(NST);
    } else {
      if (two) FI_S3_WAIT((NR - 2) * NW + 2 * NST); else FI_S3_WAIT((NR - 2) * NW + NST);
    }
#undef FI_S3_WAIT
  };

  // ---- the run: input slices z = dlo - 1 .. dhi (zeros outside the volume).  At the top of iteration z the ring holds z .. z + NR - 2 and
  // ONE free slot (slice z - 1's, released at the last barrier), into which slice z + NR - 1 is issued right away; at the bottom slice
  // z + 1 must have landed: the NR - 2 younger slices and the epilogue's stores may stay in flight.  One barrier per slice.
  int znext = dlo - 1;                                           // next slice to stage
#pragma unroll
  for (int k = 0; k < NR - 1; ++k) {
    if (znext <= dhi) issue_slice(znext);
    ++znext;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[q][m][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                               // (also: the filter fragments are in LDS)
  for (int z = dlo - 1; z <= dhi; ++z) {
    const bool issued = znext <= dhi;
    if (issued) issue_slice(znext);
    ++znext;
    slice_mma((unsigned)(slot_of(z) * SLOT));
    const bool store = z - 1 >= dlo;                             // (the first two slices complete outputs of the run below this one: dropped)
    if (store) epilogue(z - 1);
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        acc[0][m][f] = acc[1][m][f];
        acc[1][m][f] = acc[2][m][f];
        acc[2][m][f] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    if (store)
      wait_keep(issued ? 1 : 0);
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fi_lds_barrier();
  }

  // ---- statistics of the run: 16-lane row sums, the four waves through LDS, one fp64 atomic per (channel, moment)
  if (a.stats) {
    float* red = reinterpret_cast<float*>(smem + O_R);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = fi_row16_sum(ssum[f][r]), q = fi_row16_sum(ssq[f][r]);
        if (li == 0) {
          red[(wave * NF * 16 + f * 16 + kg * 4 + r) * 2 + 0] = s;
          red[(wave * NF * 16 + f * 16 + kg * 4 + r) * 2 + 1] = q;
        }
      }
    __syncthreads();
    if (tid < NF * 16 * 2) {
      const int c = tid >> 1, which = tid & 1;
      if (c < cout) {
        double tot = 0.0;
#pragma unroll
        for (int wv_ = 0; wv_ < NWV; ++wv_) tot += (double)red[(wv_ * NF * 16 + c) * 2 + which];
        const int slot = blockIdx.x & (FI_STATS_SLOTS - 1);
        atomicAdd(&a.stats[(size_t)n * a.stats_stride + ((size_t)slot * cout + c) * 2 + which], tot);
      }
    }
  }
}

template <typename T, int C1, int NF, int MF, int NR, int NWV, bool WREG>
int launch_stream(S3Args a, hipStream_t st) {
  constexpr int KSZ = 5 + (C1 ? 9 : 0), NKS = 3 * KSZ, TH = NWV * MF, NPX = 18 * (TH + 2);
  constexpr int P0W = ((NPX * 2 + 63) / 64 + 3) / 4, P1W = C1 ? ((NPX * 4 + 63) / 64 + 3) / 4 : 0;
  constexpr size_t lds = (size_t)NR * 4 * (P0W + P1W) * 1024 + (size_t)(WREG ? 0 : NF * NKS * 1024) + (size_t)NWV * NF * 16 * 2 * 4;
  static const bool big = fi_allow_big_lds((const void*)conv3d_stream_kernel<T, C1, NF, MF, NR, NWV, WREG>);
  (void)big;
  a.tilesX = (a.W + 15) / 16, a.tilesY = (a.H + TH - 1) / TH;
  // runs of >= 16 slices, as many runs as it takes to have ~768 workgroups
  const long tiles = (long)a.N * a.tilesX * a.tilesY;
  long nseg = (768 + tiles - 1) / tiles;
  if (nseg > a.D / 16) nseg = a.D / 16;                          // (two slices of every run are multiplied for nothing)
  if (nseg < 1) nseg = 1;
  a.dseg = (int)((a.D + nseg - 1) / nseg);
  a.nseg = (a.D + a.dseg - 1) / a.dseg;
  const long blocks = tiles * a.nseg;
  hipLaunchKernelGGL((conv3d_stream_kernel<T, C1, NF, MF, NR, NWV, WREG>), dim3((unsigned)blocks), dim3(NWV * 64), lds, st, a);
  FI_CHECK_LAUNCH();
  return 0;
}

long g_stream3d = -1;       // fi_conv3d_tuning: -1 = the FI_CONV3D_STREAM environment default (1), 0 = off, 1 = on

}  // namespace

extern "C" int fi_conv3d_tuning(int stream_on) {
  g_stream3d = stream_on;
  return 0;
}

// The shapes the streaming kernel covers, tried by fi_conv3d_fwd_fused / fi_conv3d_dgrad_fused before the general one-launch form:
// 16-bit storage, sources of 16 [+ 32] channels, 16 or 48 (16 + 32, or one tensor of 48) outputs, byte offsets below 2^32.
// FI_ERR_UNSUPPORTED: not covered (the caller goes on to the general form).
int fi_conv3d_stream(int dtype, int N, int D, int H, int W, int c0, int c1, int co0, int co1, const void* x0, const void* x1, const void* w,
                     const float* bias, void* y0, void* y1, double* stats, long stats_stride, hipStream_t st) {
  static const long env_on = [] {
    const char* e = getenv("FI_CONV3D_STREAM");
    return e ? atol(e) : 1L;
  }();
  if (!(g_stream3d >= 0 ? g_stream3d != 0 : env_on != 0)) return FI_ERR_UNSUPPORTED;
  if (dtype != FI_BF16 && dtype != FI_F16) return FI_ERR_UNSUPPORTED;
  const int cout = co0 + co1;
  if (c0 != 16 || (c1 != 0 && c1 != 32) || (cout != 16 && cout != 48) || co0 % 8 || co1 % 8 || (c1 && cout != 16)) return FI_ERR_UNSUPPORTED;
  if (D < 4 || H < 8 || W < 16) return FI_ERR_UNSUPPORTED;
  const long vox = (long)N * D * H * W;
  if (vox * 48 * 2 >= (1L << 32) || vox >= (1L << 31)) return FI_ERR_UNSUPPORTED;
  S3Args a;
  a.x0 = x0, a.x1 = x1, a.w = w, a.bias = bias, a.y0 = y0, a.y1 = y1, a.stats = stats, a.stats_stride = stats_stride;
  a.N = N, a.D = D, a.H = H, a.W = W, a.co0 = co0, a.co1 = co1;
  a.tilesX = a.tilesY = a.nseg = a.dseg = 0;
  // Tile / ring shapes, all 8 waves x 2 rows (16-row tiles), ring of 4 -- the best of the measured ones (tools/c3s_bench.py, 2 x 128^3 bf16,
  // us per launch, general one-launch form in brackets; profiles/r06_b_conv3d_stream.txt):
  //   16 -> 16        filter in LDS   85 [236]   (filter in 60 registers: 111 -- spills at the 128-register cap; 4 waves x 4 rows: 94;
  //                                               ring of 8: 112 -- fewer workgroups per CU cost more than deeper prefetch buys)
  //   (16 + 32) -> 16 filter in 168 registers 204 [507]   (in LDS: 213; 8-row tiles, ring 5: 246; 4 waves x 4 rows: 235)
  //   16 -> (16 + 32) filter in LDS  231 [336]   (ring 8: 238; 4 waves x 4 rows: 244; 8-row tiles: 259)
  // What is left is not bytes (1.27 x halo + 1.125 x run overlap: ~65 us at 5 TB/s for 16 -> 16) but the per-slice chain of one
  // workgroup: MFMAs, epilogue, DMA issue, wait and ONE barrier in sequence with 2-4 waves per SIMD to overlap them.
  if (dtype == FI_BF16) {
    if (c1 == 0 && cout == 16) return launch_stream<bf16_t, 0, 1, 2, 4, 8, false>(a, st);
    if (c1 == 0 && cout == 48) return launch_stream<bf16_t, 0, 3, 2, 4, 8, false>(a, st);
    return launch_stream<bf16_t, 32, 1, 2, 4, 8, true>(a, st);
  }
  if (c1 == 0 && cout == 16) return launch_stream<f16_t, 0, 1, 2, 4, 8, false>(a, st);
  if (c1 == 0 && cout == 48) return launch_stream<f16_t, 0, 3, 2, 4, 8, false>(a, st);
  return launch_stream<f16_t, 32, 1, 2, 4, 8, true>(a, st);
}
