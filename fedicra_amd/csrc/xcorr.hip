// Statistics of a 3x3 convolution WITHOUT the convolution (gfx950 / CDNA4 only).
//
// FedICRA's K-1 no-grad LC forwards (/root/reference/code/flower_pCE_2D.py:128-139) run the auxiliary head
// Conv2d(64, 512, 3, padding 1) + BatchNorm2d (/root/reference/code/networks/unet.py:261-267) for ONE side effect: the batch
// statistics that move the BatchNorm's running mean / variance.  Nobody reads the 512-channel output.  Per output channel c,
// with y_c(p) = sum_t w_{c,t} . z(p + t) + b_c over the zero-padded input z (64 channels):
//
//   sum_p y_c    = sum_t w_{c,t} . S_t                            S_t = sum of z over the pixels tap t reaches
//   sum_p y_c^2  = sum_{t,t'} w_{c,t}^T M_{t,t'} w_{c,t'}         M_{t,t'} = sum_p z(p+t) z(p+t')^T   (64 x 64)
//
// Summed over ALL positions p of the plane (the image plus the one-pixel frame around it where some tap still reaches the
// image) M_{t,t'} is the input's AUTOCORRELATION at displacement d = t' - t,  A_d = sum_q z(q) z(q + d)^T, and A_{-d} = A_d^T:
// 13 matrices of 64 x 64 per statistics group instead of 512 output channels -- 53 K multiply-adds per pixel instead of
// 295 K.  What the frame contributes is taken off again by evaluating the convolution ON THE FRAME (4 edges x ~130 positions
// per image: 3 % of the positions, each reached by 3 taps at most) with the library's own 1x1 convolution over gathered edge
// strips.  Pipeline of fi_conv2d_stats_xcorr (all on the caller's stream, caller-owned workspace):
//
//   xcorr_partial_kernel   A_d partial sums per workgroup on the matrix pipe (v_mfma_f32_32x32x16), K = pixels: the BatchNorm
//                          + LeakyReLU of the producing layer applied while a row is staged (z rounded exactly as
//                          fi_bn_act_fwd rounds it), rows streamed through a 4-row LDS ring, operands by transposing LDS reads
//   xcorr_reduce_kernel    partials -> A[g][13][64][64], T[g][64] (channel sums) in fp64, fixed order
//   edge_gather_kernel     the four edge strips of every image as im2col rows [N][positions][3 x 64]  +  ring_weights_kernel
//   fi_conv2d_fwd_fused    x 4 (1x1, 192 -> Cout, statistics only): sum / sum of squares of the convolution on the frame
//   wpair_kernel           B[c][d] = sum_{t'-t=d} w_{c,t} (x) w_{c,t'}   (weights only; x 2 for d != 0: A_{-d} = A_d^T)
//   quadform_kernel        Q[g][c] = <A[g], B[c]> in fp64
//   combine_kernel         (sum, sum of squares) of y INSIDE the image, bias included -> the conv epilogue's stats layout
//
// The direct launch takes its statistics of the ROUNDED 16-bit outputs; this form of the exact fp32 ones: the two differ by
// the rounding noise of 2 x 10^5 outputs per channel (~1e-6 relative), which is the stated tolerance of the tests.  fp32
// parity mode keeps the direct launch.
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short xc_s16x4;

namespace {

constexpr int XC_C = 64;                 // input channels (the head reads ft_chns[2])
constexpr int XC_PIXB = 192;             // LDS bytes per pixel: 128 + 64 of padding.  ds_read_b64_tr_b16 is served 32 lanes at a time =
                                         // 4 consecutive pixels x 64 B: at 192 B they fall on the four disjoint quarters of the 64 banks
                                         // (a 144 B stride measured SQ_LDS_BANK_CONFLICT = 45 % of the LDS cycles)
#ifndef XC_PREFETCH
#define XC_PREFETCH 4
#endif
constexpr int XC_TAPS = 13;              // displacements (dr, dc) with dr > 0, or dr == 0 and dc >= 0, |dr|, |dc| <= 2
constexpr int XC_PART = XC_TAPS * XC_C * XC_C + XC_C;      // floats per group result: 13 matrices + 64 channel sums
constexpr int XC_RAW = 4 * 7 * 2 * 16 * 64;               // floats per workgroup partial: [wave][tap][a block][4 regs][lane][4] -- the
                                                         // accumulators as they sit in the registers, 16-byte stores

struct XcArgs {
  const void* x;             // [N][H][W][64] 16-bit: raw output of the producing convolution, or the activation itself
  const float* scale;        // [groups][64] or NULL (no transform)
  const float* shift;
  float slope;
  int N, H, W, gimages, groups, wpg, rpw;
  float* part;               // [groups * wpg][XC_RAW]
#ifdef XC_TRACE
  long long* trace;          // [workgroups][4]: s_memtime at kernel start, first row, after the last row, end
#endif
};
#ifdef XC_TRACE
static long long* g_xc_trace = nullptr;
extern "C" void fi_xcorr_debug_set_trace(long long* p) { g_xc_trace = p; }
#define XC_TR(i) do { if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 4 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define XC_TR(i) do {} while (0)
#endif

__device__ __forceinline__ f32x16 xc_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 xc_mfma(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// One 32-channel x 16-pixel MFMA operand from a pixel-major LDS row: two ds_read_b64_tr_b16.  In each 16-lane group lane i
// addresses 4 channels (8 bytes) of pixel (i >> 2) and receives channel i of 4 consecutive pixels (conv_impl.h WgFrag;
// semantics pinned on hardware by tests/test_ops_gpu.py::test_tr16_semantics).  `addr` already holds the lane's share:
//   ((8 * (lane >> 5) + ((lane & 15) >> 2)) * PIXB + ((lane >> 4) & 1) * 32 + (lane & 3) * 8.
template <typename T>
__device__ __forceinline__ typename DT<T>::frag_t xc_frag(const char* addr) {
  typedef __attribute__((address_space(3))) xc_s16x4 lds_v;
  const xc_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)addr);
  const xc_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(addr + 4 * XC_PIXB));
  union {
    xc_s16x4 h[2];
    typename DT<T>::frag_t v;
  } u;
  u.h[0] = lo;
  u.h[1] = hi;
  return u.v;
}

// one 32-bit word = two 16-bit channels
template <typename T> struct XcWord;
template <> struct XcWord<bf16_t> {
  static __device__ __forceinline__ void unpack(unsigned w, float& f0, float& f1) {
    f0 = __uint_as_float(w << 16), f1 = __uint_as_float(w & 0xffff0000u);
  }
  static __device__ __forceinline__ unsigned pack(float f0, float f1) {
    bf16x2_t p;
    p[0] = (bf16_t)f0, p[1] = (bf16_t)f1;
    return __builtin_bit_cast(unsigned, p);
  }
};
template <> struct XcWord<f16_t> {
  static __device__ __forceinline__ void unpack(unsigned w, float& f0, float& f1) {
    const f16x2_t p = __builtin_bit_cast(f16x2_t, w);
    f0 = (float)p[0], f1 = (float)p[1];
  }
  static __device__ __forceinline__ unsigned pack(float f0, float f1) {
    f16x2_t p;
    p[0] = (f16_t)f0, p[1] = (f16_t)f1;
    return __builtin_bit_cast(unsigned, p);
  }
};

// taps of the two tap halves (th = wave >> 1): (dr, dc); the 7th "tap" of half 1 reads a block of ones: its two MFMAs yield
// the channel sums T (D[i][j] = sum_k z[k][i] for every j)
__device__ __forceinline__ void xc_tap(int th, int i, int& dr, int& dc, bool& ones) {
  const int t = th * 7 + i;
  ones = t == 13;
  // 0:(0,0) 1:(0,1) 2:(0,2) 3:(1,-2) 4:(1,-1) 5:(1,0) 6:(1,1) 7:(1,2) 8:(2,-2) 9:(2,-1) 10:(2,0) 11:(2,1) 12:(2,2)
  dr = t < 3 ? 0 : (t < 8 ? 1 : 2);
  dc = t < 3 ? t : (t < 8 ? t - 5 : t - 10);
  if (ones) dr = 0, dc = 0;
}

template <typename T, int NV, bool XF>        // NV = W / 32: 16-byte vectors per thread and row; XF: the source is raw (transform)
__global__ __launch_bounds__(256, 1) void xcorr_partial_kernel(XcArgs a) {
  typedef typename DT<T>::frag_t frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = wave & 1, th = wave >> 1;
  const int W = a.W, H = a.H, HV = H + 2;                // HV: virtual rows per image = its rows + two zero rows behind it
  const int rowb = (W + 4) * XC_PIXB;
  char* const ones = smem + 4 * rowb;
  const int g = blockIdx.x / a.wpg, k = blockIdx.x % a.wpg;
  const int R0 = k * a.rpw;
  int R1 = R0 + a.rpw;
  if (R1 > a.gimages * H) R1 = a.gimages * H;
  if (R0 >= R1) return;                                  // (the reducer knows how many workgroups of a group hold rows)
  XC_TR(0);
  // ---- LDS: the two pad pixels on either side of every ring row are zeroed once (rows themselves are always written whole
  // before they are read); a row of ones
  for (int i = tid; i < 4 * 4 * (XC_PIXB / 16); i += 256) {
    const int q = i % (XC_PIXB / 16), pp = (i / (XC_PIXB / 16)) % 4, slot = i / (4 * (XC_PIXB / 16));
    const int px = pp < 2 ? pp : W + pp;
    reinterpret_cast<uint4*>(smem + slot * rowb + px * XC_PIXB)[q] = make_uint4(0u, 0u, 0u, 0u);
  }
  {
    T* o = reinterpret_cast<T*>(ones);
    for (int i = tid; i < rowb / 2; i += 256) o[i] = from_f32<T>(1.0f);     // a whole row of ones: the pseudo-tap strides like the others
  }
  // ---- staging: thread -> channel vector cv (8 channels) of pixels (tid >> 3) + 32 j
  const int cv = tid & 7, px0 = tid >> 3;
  float sc[8], sh[8];
  constexpr bool xf = XF;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = xf ? a.scale[(size_t)g * XC_C + cv * 8 + j] : 1.f;
    sh[j] = xf ? a.shift[(size_t)g * XC_C + cv * 8 + j] : 0.f;
  }
  const float slope = a.slope;
  const T* const xg = reinterpret_cast<const T*>(a.x) + (size_t)g * a.gimages * H * W * XC_C + cv * 8;
  // the row being staged: NV vectors as typed 32-bit words (two channels each); rn = the row after it, loaded while rw is
  // transformed and written
  unsigned rw[NV][4], rn[NV][4];
  bool rvalid = false, nvalid = false;
  // virtual row v = il * HV + lr: (il, lr) of the NEXT row to load are carried along (no division per row)
  int l_il = 0, l_lr = 0;
  auto seek_row = [&](int v) { l_il = v / HV, l_lr = v - l_il * HV; };
  auto load_row = [&]() {
    nvalid = l_lr < H && l_il < a.gimages;
    const size_t rowoff = ((size_t)(nvalid ? l_il : 0) * H + (nvalid ? l_lr : 0)) * W;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const uint4 q = *reinterpret_cast<const uint4*>(xg + (rowoff + px0 + 32 * j) * XC_C);
      rn[j][0] = q.x, rn[j][1] = q.y, rn[j][2] = q.z, rn[j][3] = q.w;
    }
    if (++l_lr == HV) l_lr = 0, ++l_il;
  };
  auto take_row = [&]() {
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) rw[j][k] = rn[j][k];
    rvalid = nvalid;
  };
  // z = act(scale * y + shift) of the two channels of word k of vector j, rounded like fi_bn_act_fwd rounds it
  auto xform_word = [&](int j, int k) __attribute__((always_inline)) {
    if constexpr (XF) {
      float f0, f1;
      XcWord<T>::unpack(rw[j][k], f0, f1);
      const float t0 = f0 * sc[2 * k] + sh[2 * k], t1 = f1 * sc[2 * k + 1] + sh[2 * k + 1];
      rw[j][k] = XcWord<T>::pack(fmaxf(t0, t0 * slope), fmaxf(t1, t1 * slope));
    }
  };
  float tf0 = 0.f, tf1 = 0.f;                 // one word in flight between the three parts of its transform
  unsigned so[4] = {0u, 0u, 0u, 0u};
  auto stage_part = [&](int j, int k, int part) __attribute__((always_inline)) {
    if constexpr (XF) {
      if (part == 0) {
        XcWord<T>::unpack(rw[j][k], tf0, tf1);
        tf0 = tf0 * sc[2 * k] + sh[2 * k];
      } else if (part == 1) {
        tf1 = tf1 * sc[2 * k + 1] + sh[2 * k + 1];
        tf0 = fmaxf(tf0, tf0 * slope);
      } else {
        tf1 = fmaxf(tf1, tf1 * slope);
        rw[j][k] = XcWord<T>::pack(tf0, tf1);
      }
    }
  };
  auto store_vec = [&](int v, int j) __attribute__((always_inline)) {
    char* const dst = smem + (v & 3) * rowb + (2 + px0 + 32 * j) * XC_PIXB + cv * 16;
    *reinterpret_cast<uint4*>(dst) = fi_vec_select(rvalid, make_uint4(rw[j][0], rw[j][1], rw[j][2], rw[j][3]));
  };
  auto write_row = [&](int v) {                              // prologue only
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
      for (int k = 0; k < 4; ++k) xform_word(j, k);
      store_vec(v, j);
    }
  };

  const int il0 = R0 / H, il1 = (R1 - 1) / H;
  const int v0 = il0 * HV + (R0 - il0 * H), v1 = il1 * HV + (R1 - 1 - il1 * H);

  f32x16 acc[7][2];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][ab][r] = 0.f;

  const int gq = lane >> 4, li = lane & 15;
  const int laneoff = (8 * (gq >> 1) + (li >> 2)) * XC_PIXB + (gq & 1) * 32 + (li & 3) * 8;
  int tdr[7], toff[7];
  bool tones[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    int dr, dc;
    xc_tap(th, i, dr, dc, tones[i]);
    tdr[i] = dr;
    toff[i] = (2 + dc) * XC_PIXB + laneoff + bh * 64;
  }

  __syncthreads();
  // prologue: rows v0 .. v0 + 2 into the ring, row v0 + 3 in registers
  seek_row(v0);
  for (int v = v0; v < v0 + 3; ++v) {
    load_row();
    take_row();
    write_row(v);
  }
  load_row();
  take_row();
  __syncthreads();

  // One segment = 16 pixels of the row = one K step: 2 A operands (64 channels of P) and, per tap, 1 B operand (this wave's 32
  // channels of Q at the tap's displacement) -> 2 MFMAs.  ONE wave per SIMD issues in order, so the instruction stream itself
  // has to keep the matrix pipe fed: a row is 7 * NSEG taps of straight-line code (every LDS offset an immediate), per tap
  //   read the B operand of tap + 2  |  MFMA 1  |  <= 9 vector instructions of staging  |  MFMA 2
  // -- an MFMA holds the pipe 32 cycles = 8 issue slots, and what stands between two of them runs in the first one's shadow.
  // The staging (the row after next: transform of NV vectors word by word, one LDS store per vector) is spread over the
  // taps; the next segment's A operands are read at tap 4 of a segment.  sched_barrier fences pin the order.  A zero row
  // behind an image is multiplied like any other (its products are zeros): no second code path, which would make hipcc move
  // the 224 accumulators between the register files.
  constexpr int NSEG = 2 * NV, NT = 7 * NSEG;
  XC_TR(1);
  for (int v = v0; v <= v1; ++v) {
    load_row();                              // row v + 4: lands during this row
    const char* tb[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) tb[i] = (tones[i] ? ones : smem + ((v + tdr[i]) & 3) * rowb) + toff[i];
    const char* const pa = smem + (v & 3) * rowb + 2 * XC_PIXB + laneoff;
    auto bread = [&](int s) __attribute__((always_inline)) {
      const int i = s % 7, seg = s / 7;
      return xc_frag<T>(tb[i] + seg * 16 * XC_PIXB);
    };
    constexpr int PD = XC_PREFETCH;            // B operands in flight ahead of the tap that uses them
    frag_t af[2][2], bf[PD + 1];
    af[0][0] = xc_frag<T>(pa), af[0][1] = xc_frag<T>(pa + 64);
#pragma unroll
    for (int q = 0; q < PD; ++q) bf[q] = bread(q);
#pragma unroll
    for (int s = 0; s < NT; ++s) {
      const int i = s % 7, seg = s / 7;
#if !defined(XC_NO_READS)
      if (s + PD < NT) bf[(s + PD) % (PD + 1)] = bread(s + PD);
#endif
#if !defined(XC_NO_READS)
      if (i == 3 && seg + 1 < NSEG) {
        af[(seg + 1) & 1][0] = xc_frag<T>(pa + (seg + 1) * 16 * XC_PIXB);
        af[(seg + 1) & 1][1] = xc_frag<T>(pa + (seg + 1) * 16 * XC_PIXB + 64);
      }
#else
      if (i == 3 && seg + 1 < NSEG) af[(seg + 1) & 1][0] = af[seg & 1][0], af[(seg + 1) & 1][1] = af[seg & 1][1];
#endif
      __builtin_amdgcn_sched_barrier(0);
#if !defined(XC_NO_MFMA)
      acc[i][0] = xc_mfma(af[seg & 1][0], bf[s % (PD + 1)], acc[i][0]);
#else
      asm volatile("" ::"v"(af[seg & 1][0]), "v"(bf[s % (PD + 1)]));
#endif
      __builtin_amdgcn_sched_barrier(0);
      // staging: word w of the row's 4 NV words at tap 3 w + 1; the store of a vector one tap after its last word
#if !defined(XC_NO_STAGE)
      // staging, one chunk of <= 3 vector instructions per tap (an MFMA's shadow hides ~5 issue slots of one wave): taps
      // 0 .. 12 NV - 1 transform the 4 NV words in three parts each, the last 2 NV taps select + store the NV vectors
      if (s < 12 * NV) {
        stage_part((s / 3) / 4, (s / 3) % 4, s % 3);
      } else {
        const int j = (s - 12 * NV) / 2;
        if ((s - 12 * NV) % 2 == 0) {
          so[0] = rvalid ? rw[j][0] : 0u, so[1] = rvalid ? rw[j][1] : 0u;
        } else {
          so[2] = rvalid ? rw[j][2] : 0u, so[3] = rvalid ? rw[j][3] : 0u;
          *reinterpret_cast<uint4*>(smem + ((v + 3) & 3) * rowb + (2 + px0 + 32 * j) * XC_PIXB + cv * 16) =
              make_uint4(so[0], so[1], so[2], so[3]);
        }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
#if !defined(XC_NO_MFMA)
      acc[i][1] = xc_mfma(af[seg & 1][1], bf[s % (PD + 1)], acc[i][1]);
#else
      asm volatile("" ::"v"(af[seg & 1][1]));
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    take_row();
    fi_lds_barrier();
  }

  XC_TR(2);
  // ---- partial sums of this workgroup, register layout (xcorr_reduce_kernel maps it back): one 16-byte store per 4 registers
  float4* const part = reinterpret_cast<float4*>(a.part + (size_t)blockIdx.x * XC_RAW) + (size_t)wave * (7 * 2 * 4 * 64) + lane;
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        part[((i * 2 + ab) * 4 + r4) * 64] = make_float4(acc[i][ab][4 * r4], acc[i][ab][4 * r4 + 1], acc[i][ab][4 * r4 + 2], acc[i][ab][4 * r4 + 3]);
  XC_TR(3);
}

// A[g][e] = sum over the workgroups of group g that hold rows, in fp64 and in workgroup order (deterministic); Af = the same
// values rounded to fp32 once (6e-8 relative: what the quadratic form reads -- half the bytes)
__global__ void xcorr_reduce_kernel(const float* part, int groups, int wpg, int used, double* A, float* Af) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)groups * XC_PART) return;
  const int g = (int)(e / XC_PART), i = (int)(e % XC_PART);
  // element (tap t, row, col) -> where xcorr_partial_kernel keeps it: wave (b half = col >> 5, tap half = t / 7), tap slot t % 7,
  // a block row >> 5, D register r = (row & 3) + 4 ((row & 31) >> 3), lane (col & 31) + 32 ((row >> 2) & 1); the channel
  // sums T[row] = the "ones" pseudo-tap (slot 6 of tap half 1), any column: column 0 of b half 0
  int t, row, col;
  if (i < XC_TAPS * XC_C * XC_C) {
    t = i >> 12, row = (i >> 6) & 63, col = i & 63;
  } else {
    t = XC_TAPS, row = i - XC_TAPS * XC_C * XC_C, col = 0;
  }
  const int wave = (col >> 5) + 2 * (t / 7), slot = t % 7, ab = row >> 5, r = (row & 3) + 4 * ((row & 31) >> 3);
  const int lane = (col & 31) + 32 * ((row >> 2) & 1);
  const size_t off = ((((size_t)wave * 7 + slot) * 2 + ab) * 4 + (r >> 2)) * 256 + (size_t)lane * 4 + (r & 3);
  const float* p = part + (size_t)g * wpg * XC_RAW + off;
  double s = 0.0;
  for (int k = 0; k < used; ++k) s += (double)p[(size_t)k * XC_RAW];
  A[e] = s;
  Af[e] = (float)s;
}

// ---- the frame: per image four edge strips as im2col rows of 3 taps x 64 channels
//   e = 0 top    p = (-1, pc), pc = pos in [0, W):        taps (+1, j-1) read z(0, pc + j - 1)          [W positions]
//   e = 1 bottom p = (H, pc):                             taps (-1, j-1) read z(H-1, pc + j - 1)
//   e = 2 left   p = (pr, -1), pr = pos - 1 in [-1, H]:   taps (j-1, +1) read z(pr + j - 1, 0)          [H + 2 positions, padded
//   e = 3 right  p = (pr, W):                             taps (j-1, -1) read z(pr + j - 1, W-1)          to 16: the corners are here]
template <typename T>
__global__ void edge_gather_kernel(XcArgs a, T* out_tb, T* out_lr, int plr) {     // [2][N][W][192], [2][N][plr][192]
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H, W = a.W;
  const long n_tb = 2L * a.N * W * 24, n_lr = 2L * a.N * plr * 24;
  if (idx >= n_tb + n_lr) return;
  const bool tb = idx < n_tb;
  long r = tb ? idx : idx - n_tb;
  const int cv = (int)(r & 7);
  r >>= 3;
  const int j = (int)(r % 3);
  r /= 3;
  const int np = tb ? W : plr;
  const int pos = (int)(r % np);
  r /= np;
  const int n = (int)(r % a.N), e2 = (int)(r / a.N);
  int y, x;
  bool ok;
  if (tb) {
    y = e2 == 0 ? 0 : H - 1;
    x = pos + j - 1;
    ok = x >= 0 && x < W;
  } else {
    x = e2 == 0 ? 0 : W - 1;
    y = pos - 1 + j - 1;
    ok = pos < H + 2 && y >= 0 && y < H;
  }
  uint4 o = make_uint4(0u, 0u, 0u, 0u);
  if (ok) {
    const int g = n / a.gimages;
    const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.x) + (((size_t)n * H + y) * W + x) * XC_C + cv * 8);
    o = raw;
    if (a.scale) {
      float f[8];
      VecWords<T>::unpack(raw, f);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float t = f[q] * a.scale[(size_t)g * XC_C + cv * 8 + q] + a.shift[(size_t)g * XC_C + cv * 8 + q];
        f[q] = fmaxf(t, t * a.slope);
      }
      o = VecWords<T>::pack(f);
    }
  }
  T* const out = tb ? out_tb : out_lr;
  *reinterpret_cast<uint4*>(out + ((((size_t)e2 * a.N + n) * np + pos) * 3 + j) * XC_C + cv * 8) = o;
}

// ring weights: wr[e][c][j][ci] = w[c][t_e(j)][ci]  (w = the forward operand [Cout][9][64])
template <typename T>
__global__ void ring_weights_kernel(const T* w, T* wr, int cout) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4L * cout * 3 * 8) return;
  const int cv = (int)(idx & 7);
  long r = idx >> 3;
  const int j = (int)(r % 3);
  r /= 3;
  const int c = (int)(r % cout), e = (int)(r / cout);
  const int t = e == 0 ? 6 + j : (e == 1 ? j : (e == 2 ? j * 3 + 2 : j * 3));
  *reinterpret_cast<uint4*>(wr + (((size_t)e * cout + c) * 3 + j) * XC_C + cv * 8) =
      *reinterpret_cast<const uint4*>(w + ((size_t)c * 9 + t) * XC_C + cv * 8);
}

// B[c][d][i][j] = m_d * sum over tap pairs (t, t') with t' - t = d of w[c][t][i] * w[c][t'][j];  m_0 = 1, else 2 (A_{-d} = A_d^T
// folds the mirrored displacement in); the tail B[c][13 * 4096 + i] = sum_t w[c][t][i] pairs with the channel sums T.
// One thread per 4 consecutive j: lanes of a row write 256 contiguous bytes.
template <typename T>
__global__ __launch_bounds__(256) void wpair_kernel(const T* w, float* B) {
  __shared__ __attribute__((aligned(16))) float ws[9 * XC_C];
  const int c = blockIdx.x;
  for (int i = threadIdx.x; i < 9 * XC_C; i += 256) ws[i] = to_f32(w[(size_t)c * 9 * XC_C + i]);
  __syncthreads();
  float* out = B + (size_t)c * XC_PART;
  for (int o = threadIdx.x; o < XC_TAPS * XC_C * XC_C / 4; o += 256) {
    const int j4 = (o & 15) * 4, i = (o >> 4) & 63, d = o >> 10;
    const int dr = d < 3 ? 0 : (d < 8 ? 1 : 2), dc = d < 3 ? d : (d < 8 ? d - 5 : d - 10);
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tr = -1; tr <= 1; ++tr) {
      const int tr2 = tr + dr;
#pragma unroll
      for (int tc = -1; tc <= 1; ++tc) {
        const int tc2 = tc + dc;
        if (tr2 <= 1 && tc2 >= -1 && tc2 <= 1) {
          const float av = ws[((tr + 1) * 3 + tc + 1) * XC_C + i];
          const float4 b = *reinterpret_cast<const float4*>(&ws[((tr2 + 1) * 3 + tc2 + 1) * XC_C + j4]);
          s4.x += av * b.x, s4.y += av * b.y, s4.z += av * b.z, s4.w += av * b.w;
        }
      }
    }
    const float m = d == 0 ? 1.f : 2.f;
    *reinterpret_cast<float4*>(out + (size_t)o * 4) = make_float4(m * s4.x, m * s4.y, m * s4.z, m * s4.w);
  }
  if (threadIdx.x < XC_C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) t += ws[k * XC_C + threadIdx.x];
    out[XC_TAPS * XC_C * XC_C + threadIdx.x] = t;
  }
}

// Qp[ks][g][c] = sum over K slice ks of Af[g][e] * B[c][e]  (e < 13 * 4096: the quadratic form), fp32 products, fp64 sums.  One
// workgroup per (8 output channels, K slice): every A value it loads meets 8 channels.
constexpr int XC_MAXG = 8, XC_KS = 8, XC_QC = 8;
__device__ __forceinline__ double xc_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void quadform_kernel(const float* Af, const float* B, int groups, int cout, double* Qp) {
  __shared__ double red[4][XC_MAXG * XC_QC];
  constexpr int NE = XC_TAPS * XC_C * XC_C, SL = NE / XC_KS;
  const int c0 = (blockIdx.x / XC_KS) * XC_QC, ks = blockIdx.x % XC_KS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s[XC_MAXG][XC_QC];
#pragma unroll
  for (int g = 0; g < XC_MAXG; ++g)
#pragma unroll
    for (int q = 0; q < XC_QC; ++q) s[g][q] = 0.0;
  for (int e = ks * SL + threadIdx.x * 4; e < (ks + 1) * SL; e += 1024) {       // 4 consecutive elements per thread and pass
    float4 b[XC_QC], av[XC_MAXG];
#pragma unroll
    for (int q = 0; q < XC_QC; ++q)
      b[q] = c0 + q < cout ? *reinterpret_cast<const float4*>(B + (size_t)(c0 + q) * XC_PART + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < XC_MAXG; ++g)
      av[g] = g < groups ? *reinterpret_cast<const float4*>(Af + (size_t)g * XC_PART + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < XC_MAXG; ++g)
#pragma unroll
      for (int q = 0; q < XC_QC; ++q)
        s[g][q] += ((double)(av[g].x * b[q].x) + (double)(av[g].y * b[q].y)) + ((double)(av[g].z * b[q].z) + (double)(av[g].w * b[q].w));
  }
#pragma unroll
  for (int g = 0; g < XC_MAXG; ++g)
#pragma unroll
    for (int q = 0; q < XC_QC; ++q) {
      const double t = xc_wave_sum(s[g][q]);
      if (lane == 0) red[wave][g * XC_QC + q] = t;
    }
  __syncthreads();
  if (threadIdx.x < XC_MAXG * XC_QC) {
    const int g = threadIdx.x / XC_QC, q = threadIdx.x % XC_QC;
    if (g < groups && c0 + q < cout)
      Qp[((size_t)ks * groups + g) * cout + c0 + q] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// (sum, sum of squares) of y = conv + bias over the pixels INSIDE the images of group g, into slot 0 of the conv epilogue's
// statistics layout [slot][Cout][2] (the caller zeroed it; the other slots stay zero)
__global__ __launch_bounds__(128) void combine_kernel(const float* bias, const double* Qp, const double* A, const float* B,
                                                      const double* ring, long ring_estride, long ring_gstride, int groups,
                                                      int cout, double count, double* stats, long stats_gstride) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= groups * cout) return;
  const int g = idx / cout, c = idx % cout;
  double q = 0.0;
#pragma unroll 4
  for (int ks = 0; ks < XC_KS; ++ks) q += Qp[((size_t)ks * groups + g) * cout + c];
  // the convolution summed over the whole plane: every tap sees every pixel once -> (sum_t w_t) . T
  const double* Tg = A + (size_t)g * XC_PART + XC_TAPS * XC_C * XC_C;
  const float* wsum = B + (size_t)c * XC_PART + XC_TAPS * XC_C * XC_C;
  double s1 = 0.0;
#pragma unroll 4
  for (int i = 0; i < XC_C; ++i) s1 += Tg[i] * (double)wsum[i];
  double r1 = 0.0, r2 = 0.0;                           // ... minus what the frame around the images holds
#pragma unroll 1
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int slot = 0; slot < FI_STATS_SLOTS; ++slot) {
      const double* rs = ring + (size_t)e * ring_estride + (size_t)g * ring_gstride + ((size_t)slot * cout + c) * 2;
      r1 += rs[0];
      r2 += rs[1];
    }
  const double b = bias ? (double)bias[c] : 0.0;
  const double conv1 = s1 - r1, conv2 = q - r2;
  double* dst = stats + (size_t)g * stats_gstride + (size_t)c * 2;
  dst[0] = conv1 + count * b;
  dst[1] = conv2 + 2.0 * b * conv1 + count * b * b;
}

struct XcPlan {
  int groups, wpg, rpw, used, plr;
  size_t o_part, o_A, o_Af, o_B, o_xe, o_wr, o_ring, o_Q, total;
  long ring_estride, ring_gstride;
};

int xc_plan(const FiConv* d, int group_images, XcPlan* p) {
  if (!d) return FI_ERR_NULL;
  if (d->dtype != FI_BF16 && d->dtype != FI_F16) return FI_ERR_UNSUPPORTED;
  if (d->ksize != 3 || d->c0 != XC_C || d->c1 != 0 || d->co1 != 0 || d->accumulate0 || d->y_f32) return FI_ERR_UNSUPPORTED;
  if (d->W != 64 && d->W != 128) return FI_ERR_UNSUPPORTED;       // whole 32-pixel vector rounds; W + 2 frame positions <= 144
  if (d->H < 4 || d->H > 4096 || d->co0 % 8 || d->co0 < 8) return FI_ERR_UNSUPPORTED;
  const int gi = group_images > 0 ? group_images : d->N;
  if (d->N < 1 || d->N % gi) return FI_ERR_SHAPE;
  p->groups = d->N / gi;
  if (p->groups > XC_MAXG) return FI_ERR_UNSUPPORTED;
  p->wpg = 256 / p->groups;
  const int rows = gi * d->H;
  p->rpw = (rows + p->wpg - 1) / p->wpg;
  p->used = (rows + p->rpw - 1) / p->rpw;
  const size_t es = 2;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = 0;
  p->o_part = o, o += al((size_t)p->groups * p->wpg * XC_RAW * 4);
  p->o_A = o, o += al((size_t)p->groups * XC_PART * 8);
  p->o_Af = o, o += al((size_t)p->groups * XC_PART * 4);
  p->o_B = o, o += al((size_t)d->co0 * XC_PART * 4);
  p->plr = ((d->H + 2 + 15) / 16) * 16;
  p->o_xe = o, o += al((size_t)2 * d->N * (d->W + p->plr) * 3 * XC_C * es);
  p->o_wr = o, o += al((size_t)4 * d->co0 * 3 * XC_C * es);
  p->ring_gstride = (long)FI_STATS_SLOTS * d->co0 * 2;
  p->ring_estride = p->ring_gstride * p->groups;
  p->o_ring = o, o += al((size_t)4 * p->ring_estride * 8);
  p->o_Q = o, o += al((size_t)XC_KS * p->groups * d->co0 * 8);
  p->total = o;
  return 0;
}


template <typename T>
int xc_run_tail(const FiConv* d, const XcArgs& a, const XcPlan& p, const void* w, const float* bias, double* stats,
                long stats_gstride, char* ws, hipStream_t st, hipStream_t ss);
template <typename T>
int xc_launch_main(const FiConv* d, const XcArgs& a, const XcPlan& p, char* ws, hipStream_t st);

template <typename T>
int xc_run(const FiConv* d, const FiInXform* t0, int group_images, const void* x0, const void* w, const float* bias,
           double* stats, long stats_gstride, char* ws, const XcPlan& p, hipStream_t st) {
  XcArgs a;
  a.x = x0;
  a.scale = t0 ? t0->scale : nullptr;
  a.shift = t0 ? t0->shift : nullptr;
  a.slope = t0 ? t0->slope : 1.f;
  a.N = d->N, a.H = d->H, a.W = d->W;
  a.gimages = group_images > 0 ? group_images : d->N;
  a.groups = p.groups, a.wpg = p.wpg, a.rpw = p.rpw;
  a.part = reinterpret_cast<float*>(ws + p.o_part);
#ifdef XC_TRACE
  a.trace = g_xc_trace;
#endif
  // (The frame and the weight-pair tensor need nothing of the autocorrelation kernel; running them on a second stream beside it
  // measured 303 us either way in round 4 -- the one-workgroup-per-CU kernel leaves the small launches no room -- and the form was
  // removed in round 6: everything in line.)
  return xc_run_tail<T>(d, a, p, w, bias, stats, stats_gstride, ws, st, st);
}

template <typename T>
int xc_launch_main(const FiConv* d, const XcArgs& a, const XcPlan& p, char* ws, hipStream_t st) {
  // ring of 4 rows + a row of ones + slack
  const size_t lds = (size_t)5 * (d->W + 4) * XC_PIXB + 8192;
  const dim3 grid((unsigned)(p.groups * p.wpg)), blk(256);
  if (d->W == 64) {
    if (a.scale)
      hipLaunchKernelGGL((xcorr_partial_kernel<T, 2, true>), grid, blk, lds, st, a);
    else
      hipLaunchKernelGGL((xcorr_partial_kernel<T, 2, false>), grid, blk, lds, st, a);
  } else {
    if (a.scale)
      hipLaunchKernelGGL((xcorr_partial_kernel<T, 4, true>), grid, blk, lds, st, a);
    else
      hipLaunchKernelGGL((xcorr_partial_kernel<T, 4, false>), grid, blk, lds, st, a);
  }
  FI_CHECK_LAUNCH();
  double* A = reinterpret_cast<double*>(ws + p.o_A);
  float* Af = reinterpret_cast<float*>(ws + p.o_Af);
  hipLaunchKernelGGL(xcorr_reduce_kernel, dim3(fi_cdiv((long)p.groups * XC_PART, 256)), dim3(256), 0, st, a.part, p.groups, p.wpg,
                     p.used, A, Af);
  FI_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int xc_run_tail(const FiConv* d, const XcArgs& a, const XcPlan& p, const void* w, const float* bias, double* stats,
                long stats_gstride, char* ws, hipStream_t st, hipStream_t ss) {
  hipError_t he;
  {
    // the big kernel goes to its queue FIRST: it takes one workgroup slot (135 KB of LDS) on every CU, and the small launches
    // of the side stream fill in beside it
    const int rc = xc_launch_main<T>(d, a, p, ws, st);
    if (rc) return rc;
  }
  T* xe_tb = reinterpret_cast<T*>(ws + p.o_xe);
  T* xe_lr = xe_tb + (size_t)2 * d->N * d->W * 3 * XC_C;
  T* wr = reinterpret_cast<T*>(ws + p.o_wr);
  double* ring = reinterpret_cast<double*>(ws + p.o_ring);
  hipLaunchKernelGGL((edge_gather_kernel<T>), dim3(fi_cdiv(2L * d->N * (d->W + p.plr) * 24, 256)), dim3(256), 0, ss, a, xe_tb, xe_lr,
                     p.plr);
  FI_CHECK_LAUNCH();
  hipLaunchKernelGGL((ring_weights_kernel<T>), dim3(fi_cdiv(4L * d->co0 * 3 * 8, 256)), dim3(256), 0, ss,
                     reinterpret_cast<const T*>(w), wr, d->co0);
  FI_CHECK_LAUNCH();
  if ((he = hipMemsetAsync(ring, 0, (size_t)4 * p.ring_estride * 8, ss)) != hipSuccess) return (int)he;
  for (int e = 0; e < 4; ++e) {
    FiConv e1;
    memset(&e1, 0, sizeof(e1));
    const int np = e < 2 ? d->W : p.plr;
    e1.dtype = d->dtype, e1.N = d->N, e1.H = np / 16, e1.W = 16, e1.ksize = 1, e1.c0 = 3 * XC_C, e1.co0 = d->co0;
    const T* xe = e < 2 ? xe_tb + (size_t)e * d->N * np * 3 * XC_C : xe_lr + (size_t)(e - 2) * d->N * np * 3 * XC_C;
    const int rc = fi_conv2d_fwd_fused(&e1, nullptr, nullptr, a.gimages, 0, xe, nullptr, wr + (size_t)e * d->co0 * 3 * XC_C, nullptr,
                                       nullptr, ring + (size_t)e * p.ring_estride, p.ring_gstride, ss);
    if (rc) return rc;
  }
  float* B = reinterpret_cast<float*>(ws + p.o_B);
  hipLaunchKernelGGL((wpair_kernel<T>), dim3(d->co0), dim3(256), 0, ss, reinterpret_cast<const T*>(w), B);
  FI_CHECK_LAUNCH();
  const double* A = reinterpret_cast<const double*>(ws + p.o_A);
  const float* Af = reinterpret_cast<const float*>(ws + p.o_Af);
  double* Q = reinterpret_cast<double*>(ws + p.o_Q);
  hipLaunchKernelGGL(quadform_kernel, dim3(fi_cdiv(d->co0, XC_QC) * XC_KS), dim3(256), 0, st, Af, B, p.groups, d->co0, Q);
  FI_CHECK_LAUNCH();
  hipLaunchKernelGGL(combine_kernel, dim3(fi_cdiv((long)p.groups * d->co0, 128)), dim3(128), 0, st, bias, Q, A, B, ring,
                     p.ring_estride, p.ring_gstride, p.groups, d->co0, (double)a.gimages * d->H * d->W, stats, stats_gstride);
  FI_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" long fi_conv2d_stats_xcorr_workspace(const FiConv* d, int group_images) {
  XcPlan p;
  const int rc = xc_plan(d, group_images, &p);
  if (rc) return rc;
  return (long)p.total;
}

extern "C" int fi_conv2d_stats_xcorr_layout(const FiConv* d, int group_images, long* offsets) {
  if (!offsets) return FI_ERR_NULL;
  XcPlan p;
  const int rc = xc_plan(d, group_images, &p);
  if (rc) return rc;
  const long o[8] = {(long)p.o_part, (long)p.o_A, (long)p.o_B, (long)p.o_xe, (long)p.o_wr, (long)p.o_ring, (long)p.o_Q, (long)p.used};
  for (int i = 0; i < 8; ++i) offsets[i] = o[i];
  return 0;
}

extern "C" int fi_conv2d_stats_xcorr(const FiConv* d, const FiInXform* t0, int group_images, const void* x0, const void* w,
                                     const float* bias, double* stats, long stats_group_stride, void* workspace,
                                     long workspace_bytes, void* stream) {
  if (!d || !x0 || !w || !stats || !workspace) return FI_ERR_NULL;
  XcPlan p;
  const int rc = xc_plan(d, group_images, &p);
  if (rc) return rc;
  if (t0 && (t0->pool || t0->drop_mode != FI_DROP_NONE)) return FI_ERR_UNSUPPORTED;
  if (t0 && t0->scale && (t0->slope < 0.f || t0->slope > 1.f)) return FI_ERR_UNSUPPORTED;
  if (workspace_bytes < (long)p.total) return FI_ERR_SHAPE;
  if (p.groups > 1 && stats_group_stride < (long)d->co0 * 2) return FI_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  return d->dtype == FI_F16 ? xc_run<f16_t>(d, t0, group_images, x0, w, bias, stats, stats_group_stride, (char*)workspace, p, st)
                            : xc_run<bf16_t>(d, t0, group_images, x0, w, bias, stats, stats_group_stride, (char*)workspace, p, st);
}
