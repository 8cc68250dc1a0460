// Shared device helpers for the fedicra_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fedicra_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef _Float16 f16_t;                  // IEEE half: the storage type of the reference's autocast path
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define FI_WAVE 64

// ---- per-dtype traits: VG = elements per 16-byte global/LDS vector,
//      KSTEP = contraction depth of one MFMA (16x16x4 f32 / 16x16x32 bf16),
//      KV = contraction elements each lane feeds per MFMA.
template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int VG = 4, KSTEP = 4, KV = 1;
  typedef float4 vec_t;
  typedef float frag_t;
};
template <> struct DT<bf16_t> {
  static constexpr int VG = 8, KSTEP = 32, KV = 8;
  typedef uint4 vec_t;
  typedef bf16x8 frag_t;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).  __syncthreads() additionally waits
// for every global load AND store of the wave (vmcnt(0)); after an epilogue's stores that is ~1 us of store-ack latency.
__device__ __forceinline__ void fi_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <> struct DT<f16_t> {
  static constexpr int VG = 8, KSTEP = 32, KV = 8;
  typedef uint4 vec_t;
  typedef f16x8 frag_t;
};
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }      // overflow -> inf (GradScaler's signal)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// ---- 16-byte vectors handled as four 32-bit words.  A vec_t that lives across a loop (the register set a persistent
//      kernel keeps in flight) must never be touched through memset or a union with T[VG]: hipcc's SROA then carries it as
//      16 separate BYTES (16 VGPRs instead of 4, reassembled with shifts at every use).  These helpers only ever use typed
//      32-bit components.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint4 fi_vec_select(bool keep, const uint4& v) {
  return make_uint4(keep ? v.x : 0u, keep ? v.y : 0u, keep ? v.z : 0u, keep ? v.w : 0u);
}
__device__ __forceinline__ float4 fi_vec_select(bool keep, const float4& v) {
  return make_float4(keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f);
}
template <typename T> struct VecWords;
template <> struct VecWords<float> {
  static __device__ __forceinline__ void unpack(const float4& v, float (&f)[4]) {
    f[0] = v.x, f[1] = v.y, f[2] = v.z, f[3] = v.w;
  }
  static __device__ __forceinline__ float4 pack(const float (&f)[4]) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct VecWords<bf16_t> {
  static __device__ __forceinline__ void unpack(const uint4& v, float (&f)[8]) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);             // bf16 -> fp32 is exact: the upper 16 bits
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ uint4 pack(const float (&f)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x2_t p;
      p[0] = (bf16_t)f[2 * i];                              // round to nearest even, as from_f32<bf16_t>
      p[1] = (bf16_t)f[2 * i + 1];
      w[i] = __builtin_bit_cast(unsigned, p);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};
template <> struct VecWords<f16_t> {
  static __device__ __forceinline__ void unpack(const uint4& v, float (&f)[8]) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f16x2_t p = __builtin_bit_cast(f16x2_t, w[i]);
      f[2 * i] = (float)p[0];
      f[2 * i + 1] = (float)p[1];
    }
  }
  static __device__ __forceinline__ uint4 pack(const float (&f)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f16x2_t p;
      p[0] = (f16_t)f[2 * i];
      p[1] = (f16_t)f[2 * i + 1];
      w[i] = __builtin_bit_cast(unsigned, p);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// Four fp32 values -> the 4-element group as stored (two 32-bit words of rounded 16-bit pairs, or a float4), and the values
// AS STORED back in fp32 (what the BatchNorm statistics are taken of).  Typed 32-bit words only: an epilogue that fills a
// T[4] and copies it out is reassembled by hipcc byte by byte (~8 extra instructions per pair).
template <typename T> struct Quad;
template <> struct Quad<float> {
  typedef float4 q_t;
  static __device__ __forceinline__ q_t pack(float (&v)[4]) { return make_float4(v[0], v[1], v[2], v[3]); }
  static __device__ __forceinline__ void unpack(const q_t& q, float (&v)[4]) { v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w; }
};
template <> struct Quad<bf16_t> {
  typedef uint2 q_t;
  static __device__ __forceinline__ void unpack(const q_t& q, float (&v)[4]) {
    v[0] = __uint_as_float(q.x << 16), v[1] = __uint_as_float(q.x & 0xffff0000u);
    v[2] = __uint_as_float(q.y << 16), v[3] = __uint_as_float(q.y & 0xffff0000u);
  }
  static __device__ __forceinline__ q_t pack(float (&v)[4]) {                 // round to nearest even, as from_f32<bf16_t>
    bf16x2_t a, b;
    a[0] = (bf16_t)v[0], a[1] = (bf16_t)v[1], b[0] = (bf16_t)v[2], b[1] = (bf16_t)v[3];
    const q_t q = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    unpack(q, v);
    return q;
  }
};
template <> struct Quad<f16_t> {
  typedef uint2 q_t;
  static __device__ __forceinline__ void unpack(const q_t& q, float (&v)[4]) {
    const f16x2_t a = __builtin_bit_cast(f16x2_t, q.x), b = __builtin_bit_cast(f16x2_t, q.y);
    v[0] = (float)a[0], v[1] = (float)a[1], v[2] = (float)b[0], v[3] = (float)b[1];
  }
  static __device__ __forceinline__ q_t pack(float (&v)[4]) {                 // overflow -> inf (GradScaler's signal)
    f16x2_t a, b;
    a[0] = (f16_t)v[0], a[1] = (f16_t)v[1], b[0] = (f16_t)v[2], b[1] = (f16_t)v[3];
    const q_t q = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    unpack(q, v);
    return q;
  }
};

// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: four v_add_f32_dpp (quad xor 1, quad
// xor 2, half-row mirror, row mirror).  The same pairings as an xor butterfly -- fp addition commutes, so the same bits --
// without its four ds_bpermute round trips through the LDS unit (what __shfl_xor compiles to).
__device__ __forceinline__ float fi_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

// wave-wide sum via xor shuffles (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// counter-based RNG for dropout: one 32-bit draw per (seed, element index), stateless, so backward
// regenerates the identical mask.  lowbias32 finaliser (2 multiplies) keyed by both seed halves -- cheap
// enough (~10 VALU ops per element) to stay under the HBM roofline of the fused BN+act+dropout pass.
__device__ __forceinline__ uint32_t fi_rand32(uint64_t seed, uint64_t idx) {
  uint32_t x = (uint32_t)idx * 0x9E3779B1u + (uint32_t)seed;
  x ^= (uint32_t)(idx >> 32) * 0x85EBCA6Bu;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x ^ (uint32_t)(seed >> 32);
}
// Element-wise dropout draws for the 4 consecutive elements of group `group` (= flat element index / 4).  A PAIR of groups (8
// consecutive elements: one 16-byte vector of 16-bit storage) shares ONE full hash of the pair index, the word w0; three more
// multiply-xorshift rounds give w1 = m(w0), w2 = m(w1), w3 = m(w2): eight 16-BIT draws, the even group from (w0, w1), the odd one from
// (w2, w3), returned in the top halves of r[0..3] so that callers keep comparing against the 32-bit threshold (the keep probability is
// quantised to 2^-16: 1.5e-5 relative at p = 0.05 .. 0.5).  m is a BIJECTION, so the eight draws carry 32 bits of entropy: each is a
// deterministic -- mixing -- function of the others, which is what a keep mask needs: each position keeps at the nominal rate and the
// keep bits of a pair (and of neighbouring pairs) show no pairwise correlation at p = 0.05 / 0.3 / 0.5
// (tests/test_ops_gpu.py::test_dropout_rng_statistics_and_replay).  Integer multiplies are quarter rate on CDNA and every conv / BN kernel
// that regenerates the mask is VALU-issue-bound (tools/ws2_trace.py: a loader stage of 256^2 32->32 takes 3 200 cycles without and
// 5 700 with dropout; tools/dma_trace.py: +2 700 on a 7 100-cycle stage): per 8 elements this is 6 multiplies + ~25 other VALU ops where
// one hash per 4 elements (rounds 3-5) cost 8 + ~28 and a full hash per element ~65 instructions per element.  A caller that draws both
// groups of a vector (VG = 8) shares the hash and w1 by common-subexpression elimination -- the calls are inlined.
__device__ __forceinline__ uint32_t fi_rand_mix(uint32_t h) {
  uint32_t g = (h ^ 0x9E3779B9u) * 0x2C1B3C6Du;
  return g ^ (g >> 15);
}
__device__ __forceinline__ void fi_rand32x4(uint64_t seed, uint64_t group, uint32_t (&r)[4]) {
  const uint32_t w0 = fi_rand32(seed, group >> 1);
  const uint32_t w1 = fi_rand_mix(w0);
  uint32_t a = w0, b = w1;
  if (group & 1) {
    a = fi_rand_mix(w1);
    b = fi_rand_mix(a);
  }
  r[0] = a << 16, r[1] = a & 0xFFFF0000u;
  r[2] = b << 16, r[3] = b & 0xFFFF0000u;
}
// keep with probability (1-p): threshold on a 32-bit uniform
__device__ __forceinline__ bool fi_keep(uint64_t seed, uint64_t idx, uint32_t drop_thresh) {
  return fi_rand32(seed, idx) >= drop_thresh;
}

// l0 * a + l1 * b with ONE defined rounding sequence (the second product rounded, then a fused multiply-add of the first): the
// bilinear up-sampling kernels (ops.hip: flat and row forms; upfuse.hip: behind the 1x1 convolution) all interpolate through this,
// along x and then along y, so that they agree bit for bit instead of as hipcc happens to contract each kernel's expression.
// (This IS the sequence hipcc had chosen for the round-1..3 kernels -- v_pk_mul_f32 l1 * b, v_pk_fma_f32 l0, a -- so the fp32
// parity runs keep the bits they had; the mirrored order moved a 4-iteration AdamW loss by 2.3e-3 against the oracle, outside
// tests/test_parity2_gpu.py's 2e-3.)
__device__ __forceinline__ float fi_lerp2(float l0, float a, float l1, float b) { return __builtin_fmaf(l0, a, l1 * b); }

static inline int fi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#define FI_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)
