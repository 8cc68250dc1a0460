// C-ABI entry points for the convolution kernels: argument checks + tile-shape selection.
#include "conv_impl.h"
#include "wgrad_rows.h"

int fi_conv_fwd_f32_k1(int th, int nf, int ck, const ConvArgs& a, hipStream_t st);
int fi_conv_fwd_f32_k3(int th, int nf, int ck, const ConvArgs& a, hipStream_t st);
int fi_conv_fwd_bf16_k1(int th, int nf, int ck, const ConvArgs& a, hipStream_t st);
int fi_conv_fwd_bf16_k3(int th, int nf, int ck, const ConvArgs& a, hipStream_t st);
int fi_conv_wgrad_f32_k1(int th, int nfo, int nfi, const WgradArgs& a, hipStream_t st);
int fi_conv_wgrad_f32_k3(int th, int nfo, int nfi, const WgradArgs& a, hipStream_t st);
int fi_conv_wgrad_bf16_k1(int th, int nfo, int nfi, const WgradArgs& a, hipStream_t st);
int fi_conv_wgrad_bf16_k3(int th, int nfo, int nfi, const WgradArgs& a, hipStream_t st);
int fi_conv_wgrad_rows_bf16(int nci, int nco, const WgRowsArgs& a, int items, hipStream_t st);
int fi_conv_wgrad_rows_f16(int nci, int nco, const WgRowsArgs& a, int items, hipStream_t st);
int fi_conv_wgrad_rows3d_bf16(int nci, int nco, const WgRowsArgs& a, int items, hipStream_t st);
int fi_conv_wgrad_rows3d_f16(int nci, int nco, const WgRowsArgs& a, int items, hipStream_t st);
int fi_conv_wgrad_rows_narrow_bf16(int narrow, const WgRowsArgs& a, int items, hipStream_t st);
int fi_conv_wgrad_rows64_bf16(int tco, int tci, const WgRowsArgs& a, int items, hipStream_t st);
int fi_conv_wgrad_rows64_f16(int tco, int tci, const WgRowsArgs& a, int items, hipStream_t st);
int fi_conv_wgrad_rows_narrow_f16(int narrow, const WgRowsArgs& a, int items, hipStream_t st);
int fi_conv_wgrad_quad_f32_k1(int th, const WgradArgs& a, hipStream_t st);
int fi_conv_wgrad_quad_f32_k3(int th, const WgradArgs& a, hipStream_t st);
int fi_conv_wgrad_quad_bf16_k1(int th, const WgradArgs& a, hipStream_t st);
int fi_conv_wgrad_quad_bf16_k3(int th, const WgradArgs& a, hipStream_t st);
int fi_conv_fwd_f16_k1(int th, int nf, int ck, const ConvArgs& a, hipStream_t st);
int fi_conv_fwd_f16_k3(int th, int nf, int ck, const ConvArgs& a, hipStream_t st);
int fi_conv_wgrad_f16_k1(int th, int nfo, int nfi, const WgradArgs& a, hipStream_t st);
int fi_conv_wgrad_f16_k3(int th, int nfo, int nfi, const WgradArgs& a, hipStream_t st);
int fi_conv_wgrad_quad_f16_k1(int th, const WgradArgs& a, hipStream_t st);
int fi_conv_wgrad_quad_f16_k3(int th, const WgradArgs& a, hipStream_t st);
int fi_conv_fwd_v2_bf16_k3(int nf, int ck, int wgs_per_cu, const ConvArgs& a, hipStream_t st);
int fi_conv_fwd_v2_f16_k3(int nf, int ck, int wgs_per_cu, const ConvArgs& a, hipStream_t st);
int fi_conv_thin_bf16(int nf, int ck, int wgs_per_cu, const ConvArgs& a, hipStream_t st);
int fi_conv_thin_f16(int nf, int ck, int wgs_per_cu, const ConvArgs& a, hipStream_t st);
int fi_conv_fwd_ws_bf16(int nf, int ck, int pw, int wgs_per_cu, const ConvArgs& a, hipStream_t st);
int fi_conv_fwd_ws_f16(int nf, int ck, int pw, int wgs_per_cu, const ConvArgs& a, hipStream_t st);
int fi_conv_fwd_ws2_bf16(int form, int wgs_per_cu, const ConvArgs& a, hipStream_t st);
int fi_conv_fwd_ws2_f16(int form, int wgs_per_cu, const ConvArgs& a, hipStream_t st);
int fi_conv_fwd_dma_bf16(int wgs_per_cu, const ConvArgs& a, hipStream_t st);
int fi_conv_fwd_dma_f16(int wgs_per_cu, const ConvArgs& a, hipStream_t st);
// conv3d_stream.hip: the thin full-resolution 3x3x3 layers streamed along the depth axis (FI_ERR_UNSUPPORTED: shape not covered)
int fi_conv3d_stream(int dtype, int N, int D, int H, int W, int c0, int c1, int co0, int co1, const void* x0, const void* x1, const void* w,
                     const float* bias, void* y0, void* y1, double* stats, long stats_stride, hipStream_t st);
int fi_conv_thin_f32n_bf16(int ck, int wgs_per_cu, const ConvArgs& a, hipStream_t st);
int fi_conv_thin_f32n_f16(int ck, int wgs_per_cu, const ConvArgs& a, hipStream_t st);
int fi_conv_narrow_in_bf16(const ConvArgs& a, hipStream_t st);
int fi_conv_narrow_in_f16(const ConvArgs& a, hipStream_t st);

// Tile height: the largest of {16, 8, 4} that still gives the 256 CUs >= 2 workgroups each;
// small feature maps fall through to TH = 4 (more, smaller workgroups).
#include <cstdlib>
#include <cstring>
// tuning knobs (read once): FI_MIN_BLOCKS = workgroups a launch should reach before the tile height stops
// shrinking; FI_WGRAD_BLOCKS = total workgroups of a wgrad launch (each spatial slice costs |dw| of workspace).
#ifdef FI_TRACE
static long long* g_trace = nullptr;
extern "C" void fi_debug_set_trace(long long* p) { g_trace = p; }
#endif

static long env_long(const char* name, long dflt) {
  const char* v = getenv(name);
  return v ? atol(v) : dflt;
}
static long min_blocks() {
  static long v = env_long("FI_MIN_BLOCKS", 512);
  return v;
}
static long wgrad_blocks() {
  static long v = env_long("FI_WGRAD_BLOCKS", 512);
  return v;
}
// the pixel-split kernel of the thin layers (fewer than 32 channels on a side): its slices are a few KB, what it lacks is
// bytes in flight -- its own workgroup count (FI_WGRAD_BLOCKS_THIN; 0 = FI_WGRAD_BLOCKS)
static long wgrad_blocks_thin() {
  static long v = env_long("FI_WGRAD_BLOCKS_THIN", 0);
  return v > 0 ? v : wgrad_blocks();
}

// which forward kernel: [0] -1 = FI_V2 from the environment (default 2: per-layer choice), 0 = one-tile kernel, 1 = persistent
// kernel wherever it applies, 3 = thin-layer kernel, 4 / 5 / 6 = wave-specialised kernel with 4 / 8 / 2 x 4 producer waves, 7 = the
// 64 x 64-wave-tile form wherever it applies (needs FiConv.w16; forcing any OTHER form also keeps launches off it); [1] slab width in 16-channel fragments, [2] channel chunk, [3] workgroups per CU (0 = default)
static long g_tune[4] = {-1, 0, 0, 0};
static long env_v2() {
  static long v = env_long("FI_V2", 2);      // 2 = the measured per-layer rule
  return v;
}
extern "C" int fi_conv_tuning(int v2, int nf, int ck, int wgs_per_cu) {
  if ((nf && nf != 1 && nf != 2 && nf != 4 && !(v2 == 7 && nf == 8)) || (ck && ck != 16 && ck != 32) || wgs_per_cu < 0 || wgs_per_cu > 16)
    return FI_ERR_SHAPE;
  if (v2 > 7) return FI_ERR_SHAPE;
  g_tune[0] = v2;
  g_tune[1] = nf;
  g_tune[2] = ck;
  g_tune[3] = wgs_per_cu;
  return 0;
}

// The 64 x 64-wave-tile form (conv_fwd_ws2_kernel) and its chunk-major operand: FI_WS2 = 0 switches both off, 1 (default)
// = the measured per-launch rule below, 2 = wherever the kernel applies.
static long env_ws2() {
  static long v = env_long("FI_WS2", 1);
  return v;
}
extern "C" int fi_conv_weight_chunk16(int dtype, int ksize, int cin, int cout) {
  if (env_ws2() == 0) return 0;
  return (dtype == FI_BF16 || dtype == FI_F16) && ksize == 3 && cin >= 32 && cin % 32 == 0 && cout >= 32 && cout % 32 == 0;
}

static int pick_th(int N, int H, int W, long per_tile_mult) {
  const int cands[3] = {16, 8, 4};
  for (int i = 0; i < 3; ++i) {
    const int th = cands[i];
    if (th > 4 && th / 2 >= H) continue;  // tile would be mostly empty
    const long blocks = (long)N * fi_cdiv(H, th) * fi_cdiv(W, 16) * per_tile_mult;
    if (blocks >= min_blocks() || th == 4) return th;
  }
  return 4;
}

static int conv_fwd_impl(const FiConv* d, const FiInXform* t0, const FiInXform* t1, int group_images, int flags,
                         const void* x0, const void* x1, const void* w, const float* bias, void* y0, void* y1,
                         double* stats, long stats_group_stride, void* stream, int depth = 0);

// FI_NARROW (default 7 = all) / fi_narrow_tuning (a bit mask): the forms for the layers with a <= 4-channel side -- the first convolution and the
// logits convolution of the U-Nets, forward, input gradient and filter gradient (conv_narrow.h, conv_thin_kernel<F32N>,
// conv_wgrad_rows_kernel<XN / DN>); 0 = the general tile kernels (what the parity tests compare against)
static long g_narrow = -1;
static long narrow_mask() {                       // bit 0: narrow-input forward, bit 1: fp32 narrow-output forward, bit 2: filter gradients
  static long v = env_long("FI_NARROW", 7);
  return g_narrow >= 0 ? g_narrow : v;
}
static bool narrow_on() { return (narrow_mask() & 4) != 0; }
extern "C" int fi_narrow_tuning(int on) {
  g_narrow = on;
  return 0;
}

extern "C" int fi_conv2d_fwd(const FiConv* d, const void* x0, const void* x1, const void* w, const float* bias,
                             void* y0, void* y1, double* stats, void* stream) {
  return conv_fwd_impl(d, nullptr, nullptr, 0, 0, x0, x1, w, bias, y0, y1, stats, 0, stream);
}

extern "C" int fi_conv2d_fwd_fused(const FiConv* d, const FiInXform* t0, const FiInXform* t1, int group_images, int flags,
                                   const void* x0, const void* x1, const void* w, const float* bias, void* y,
                                   double* stats, long stats_group_stride, void* stream) {
  if (!d) return FI_ERR_NULL;
  if (d->co1 != 0 || d->accumulate0 || d->accumulate1 || d->y_f32) return FI_ERR_UNSUPPORTED;
  if (group_images < 0 || (group_images > 0 && d->N % group_images)) return FI_ERR_SHAPE;
  if (flags & ~FI_FUSED_SHARED_SOURCE0) return FI_ERR_UNSUPPORTED;
  if ((flags & FI_FUSED_SHARED_SOURCE0) && (group_images < 1 || !t0)) return FI_ERR_SHAPE;
  return conv_fwd_impl(d, t0, t1, group_images, flags, x0, x1, w, bias, y, nullptr, stats, stats_group_stride, stream);
}

static int fill_xform(const FiInXform* t, InXform* o, int is_second) {
  memset(o, 0, sizeof(*o));
  o->slope = 1.f;
  if (!t) return 0;
  if ((t->scale == nullptr) != (t->shift == nullptr)) return FI_ERR_NULL;
  if (t->drop_p > 0.f && t->drop_mode != FI_DROP_NONE) {
    if (t->drop_mode != FI_DROP_RNG_ELEM || is_second || t->pool || !t->scale) return FI_ERR_UNSUPPORTED;
    o->drop_mode = FI_DROP_RNG_ELEM;
    const double th = (double)t->drop_p * 4294967296.0;          // as fi_bn_act_fwd (ops.hip make_drop)
    o->thresh = th >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)th;
    o->keep_scale = t->drop_p < 1.f ? 1.0f / (float)(1.0 - (double)t->drop_p) : 0.f;
    o->seed = t->seed;
    o->seed_gstride = t->seed_group_stride;
    o->seed_offset = t->seed_offset;
  }
  if (!(t->slope >= 0.f && t->slope <= 1.f)) return FI_ERR_UNSUPPORTED;   // the loaders evaluate max(v, slope * v)
  o->scale = t->scale;
  o->shift = t->shift;
  o->slope = t->slope;
  return 0;
}

static int conv_fwd_impl(const FiConv* d, const FiInXform* t0, const FiInXform* t1, int group_images, int flags,
                         const void* x0, const void* x1, const void* w, const float* bias, void* y0, void* y1,
                         double* stats, long stats_group_stride, void* stream, int depth) {
  if (!d || !x0 || !w) return FI_ERR_NULL;
  if (!y0 && (!stats || y1)) return FI_ERR_NULL;          // y0 == NULL: statistics-only launch (nothing is stored)
  if (d->dtype != FI_F32 && d->dtype != FI_BF16 && d->dtype != FI_F16) return FI_ERR_DTYPE;
  if (d->ksize != 1 && d->ksize != 3) return FI_ERR_UNSUPPORTED;
  if (d->N < 1 || d->H < 1 || d->W < 1 || d->c0 < 1 || d->c1 < 0 || d->co0 < 1 || d->co1 < 0) return FI_ERR_SHAPE;
  if ((d->c1 > 0 && !x1) || (d->co1 > 0 && !y1)) return FI_ERR_NULL;
  if ((long)d->N * d->H * d->W >= (1L << 31)) return FI_ERR_UNSUPPORTED;   // kernels index pixels with 32 bits
  const int cin = d->c0 + d->c1, cout = d->co0 + d->co1;
  const bool f32 = d->dtype == FI_F32;
  int ck;
  if (f32)
    ck = cin >= 16 ? 16 : (cin > 4 ? 8 : 4);
  else
    ck = cin >= 32 ? 32 : (cin > 8 ? 16 : 8);
  {  // channel counts that are not whole 16-byte vectors: only the narrowest chunk has the element-wise loader
    const int vg = f32 ? 4 : 8;
    if (d->c0 % vg || d->c1 % vg) ck = vg;
  }
  // Output-channel slab per workgroup: as wide as possible (less re-staging of the input tile) as long as the
  // launch still has FI_TARGET_BLOCKS workgroups.  The small feature maps are latency-bound with one workgroup
  // per CU; narrower slabs put several workgroups on each CU, whose staging round trips then overlap.
  static const long target_blocks = env_long("FI_TARGET_BLOCKS", 1024);
  int nf = cout > 32 ? 4 : (cout > 16 ? 2 : 1);
  int nct = fi_cdiv(cout, nf * 16);
  int th = pick_th(d->N, d->H, d->W, nct);
  while (nf > 1 && (long)d->N * fi_cdiv(d->H, th) * fi_cdiv(d->W, 16) * nct < target_blocks) {
    nf /= 2;
    nct = fi_cdiv(cout, nf * 16);
    th = pick_th(d->N, d->H, d->W, nct);
  }
  {  // tuning knobs (tools/kbench.py sweeps): force the tile height / slab width
    static const long force_th = env_long("FI_FORCE_TH", 0), force_nf = env_long("FI_FORCE_NF", 0);
    if (force_nf) {
      nf = (int)force_nf;
      while (nf > 1 && (nf / 2) * 16 >= cout) nf /= 2;
      nct = fi_cdiv(cout, nf * 16);
    }
    if (force_th) th = (int)force_th;
  }
  // Deep layers (many input channels, small maps) spend their time in the sequential stage -> MFMA round trips of
  // the channel-chunk loop (1.5-2 us each, tools/ktrace.py): chunks twice as wide when the slab is narrow enough for
  // the LDS tile to stay under FI_FWD_LDS_CAP, so that several workgroups still share a CU.
  {
    static const long lds_cap = env_long("FI_FWD_LDS_CAP", 40 * 1024);
    const int big = f32 ? 32 : 64, vg = f32 ? 4 : 8, esz = f32 ? 4 : 2, kstep = f32 ? 4 : 32;
    const int halo = d->ksize / 2, kk = d->ksize * d->ksize;
    const long kcp = (long)((kk * big + kstep - 1) / kstep) * kstep;
    const int pad = f32 ? vg : 16;                     // FiLdsStride: 16-bit tiles are padded to 32 bytes mod 64
    const long lds = ((long)(th + 2 * halo) * (16 + 2 * halo) * (big + pad) + (long)nf * 16 * (kcp + pad)) * esz;
    if (nf <= 2 && ck * 2 == big && cin >= 2 * big && d->c0 % big == 0 && d->c1 % big == 0 && lds <= lds_cap) ck = big;
  }
  ConvArgs a;
  if (int rc = fill_xform(t0, &a.t0, 0)) return rc;
  if (int rc = fill_xform(t1, &a.t1, 1)) return rc;
  a.xf = 0;
  if (t0 || t1) {
    const int vg = f32 ? 4 : 8;
    if (ck <= vg) return FI_ERR_UNSUPPORTED;              // whole-vector channel counts only (host mirror checks)
    const bool pool = t0 && t0->pool;
    if (t1 && t1->pool) return FI_ERR_UNSUPPORTED;
    if (pool && (d->c1 != 0 || d->ksize != 3)) return FI_ERR_UNSUPPORTED;
    a.xf = pool ? 2 : 1;
  }
  a.bcast0 = (flags & FI_FUSED_SHARED_SOURCE0) ? 1 : 0;
  a.gimages = group_images;
  a.stats_gstride = stats_group_stride;
  a.x0 = x0;
  a.x1 = x1 ? x1 : x0;
  a.w = w;
  a.bias = bias;
  a.y0 = y0;
  a.y1 = y1 ? y1 : y0;
  a.stats = stats;
  a.N = d->N;
  a.H = d->H;
  a.W = d->W;
  a.c0 = d->c0;
  a.c1 = d->c1;
  a.co0 = d->co0;
  a.co1 = d->co1;
  a.acc0 = d->accumulate0;
  a.acc1 = d->accumulate1;
  a.y_f32 = f32 ? 0 : d->y_f32;
  a.tilesX = fi_cdiv(d->W, 16);
  a.tilesY = fi_cdiv(d->H, th);
  a.nct = nct;
  a.depth = 0;
  a.wrows = 0;
#ifdef FI_TRACE
  a.trace = g_trace;
#endif
  hipStream_t st = (hipStream_t)stream;
  if (narrow_mask() && !f32 && d->ksize == 3 && depth == 0 && a.xf == 0 && !a.bcast0 && d->c1 == 0 && d->co1 == 0 && y0 && !a.acc0) {
    const long px = (long)d->N * d->H * d->W;
    // narrow input side (conv_narrow_in_kernel): the first convolution (1 / 3 -> 16) and the input gradient of the logits
    // convolution (n_class -> 16); 64-column tiles
    if ((narrow_mask() & 1) && d->c0 <= 4 && (cout == 16 || cout == 8) && !a.y_f32 && d->W >= 32 && d->H >= 8 && px * cout * 2 < (1L << 32)) {
      a.tilesX = fi_cdiv(d->W, 64);
      a.tilesY = fi_cdiv(d->H, 16);
      a.nct = 1;
      return d->dtype == FI_F16 ? fi_conv_narrow_in_f16(a, st) : fi_conv_narrow_in_bf16(a, st);
    }
    // narrow output side with fp32 results (conv_thin_kernel<F32N>): the logits convolution (16 / 32 -> n_class <= 4)
    if ((narrow_mask() & 2) && a.y_f32 && cout <= 4 && (d->c0 == 16 || d->c0 == 32) && d->H >= 8 && px * d->c0 * 2 < (1L << 32) && px * cout * 4 < (1L << 32)) {
      a.tilesY = fi_cdiv(d->H, 16);
      a.nct = 1;
      static const long wgs_env = env_long("FI_THIN_F32N_WGS", 8);
      return d->dtype == FI_F16 ? fi_conv_thin_f32n_f16(d->c0, (int)wgs_env, a, st) : fi_conv_thin_f32n_bf16(d->c0, (int)wgs_env, a, st);
    }
  }
  {
    // persistent form (conv_fwd_v2_kernel): 16-bit storage, 3x3, whole-vector channel counts, plain epilogue, 16-row tiles
    const long v2 = g_tune[0] >= 0 ? g_tune[0] : env_v2(), v2_nf = g_tune[1], v2_ck = g_tune[2], v2_wgs = g_tune[3];
    const bool plain = !a.y_f32 && !a.acc0 && !a.acc1 && a.co0 % 4 == 0 && a.co1 % 4 == 0;
    // 64 x 64-wave-tile form (conv_fwd_ws2_kernel; conv_ws2.h): needs the chunk-major second operand (FiConv.w16), 16-channel
    // chunks that do not straddle the two sources, whole 64-channel slabs, a plain epilogue and 32-bit byte offsets
    // the fields the 64 x 64 form overwrites, as the other forms expect them (restored when its launcher declines)
    const void* const w_plain = a.w;
    const int tilesY_plain = a.tilesY, nct_plain = a.nct;
    if (d->w16 && depth == 0 && !f32 && d->ksize == 3 && (a.xf == 0 || a.xf == 1) && plain && env_ws2() != 0 &&
        (g_tune[0] < 0 || g_tune[0] == 7)) {                   // forcing any other form (incl. 2 = the old rule) keeps off it
      const long big = (long)d->N * d->H * d->W * 2;
      const int rows = d->w16_rows > 0 ? d->w16_rows : cout;
      // pixel groups of 64 (16-row tiles) / 32 (32-row tiles) channels must not straddle the two sources
      const bool okb = d->c0 % 32 == 0 && d->c1 % 32 == 0 && cin >= 32 && d->co0 % 8 == 0 && d->co1 % 8 == 0 &&
                       big * d->c0 < (1L << 32) && big * d->c1 < (1L << 32) && big * d->co0 < (1L << 32) &&
                       big * d->co1 < (1L << 32) && (long)rows * 9 * cin * 2 < (1L << 32);
      const bool ok32 = okb && cout % 64 == 0;
      const bool ok16 = ok32 && d->c0 % 64 == 0 && d->c1 % 64 == 0 && cout % 128 == 0;
      // whole filter resident in LDS (forms 3 / 4): Cout = 64 or 32 and 2 pixel groups (78 336 B) + Cin x Cout x 18 B + strip fit.
      // Measured (profiles/r03_u_kbench2_ws2res.txt, same box): the batched fused launches of 256^2 32->32 233 -> 200 us (with
      // dropout 359 -> 293), 256^2 64->32 385 -> 335, 128^2 64->64 146 -> 136 (with dropout 194 -> 172) against the 32-pixel-tile form
      const bool okres = okb && (cout == 64 || cout == 32) && 78336L + (long)cin * cout * 18 + 768 <= 160 * 1024;
      static const long res_min = env_long("FI_WS2_RES_MIN", 1024);
      const long tiles32 = (long)d->N * fi_cdiv(d->H, 32) * fi_cdiv(d->W, 16);
      const bool want_res = okres && (g_tune[0] == 7 ? g_tune[1] == 4 : (env_ws2() == 2 || (a.xf == 1 && tiles32 >= res_min)));
      if (want_res) {
        a.w = d->w16;
        a.wrows = rows;
        a.tilesY = fi_cdiv(d->H, 32);
        a.nct = 1;
        const int form = cout == 64 ? 3 : 4;
        // 32 outputs with element dropout in the loader: runs of half the length measured 10 % ahead (329 -> 291 us, 256^2 32->32)
        const int wgs = g_tune[3] ? (int)g_tune[3] : ((form == 4 && a.xf == 1 && a.t0.drop_mode == FI_DROP_RNG_ELEM) ? 2 : 0);
        const ConvArgs keep = a;
        const int rc = d->dtype == FI_F16 ? fi_conv_fwd_ws2_f16(form, wgs, a, st) : fi_conv_fwd_ws2_bf16(form, wgs, a, st);
        if (rc != FI_ERR_UNSUPPORTED) return rc;
        a = keep;                                // the launcher's own bounds (LDS bytes, WR / SI shapes) said no: the forms below
        a.w = w_plain, a.wrows = 0, a.tilesY = tilesY_plain, a.nct = nct_plain;
        goto ws2_done;
      }
      // measured rule (FI_WS2 = 1; profiles/r03_*_kbench2_ws2*.txt): the batched fused launches that fill the persistent grid,
      // 128-channel slabs with 64-channel pixel groups -- 1.05-1.16x the 32-pixel-tile form there (64^2 128->128 160 -> 145 us,
      // 32^2 256->256 159 -> 137, head 887 -> 841); the 32-row x 64-channel shape and the 12-image launches measured level
      // or behind and stay where they were
      const long tiles16 = (long)d->N * fi_cdiv(d->H, 16) * fi_cdiv(d->W, 16);
      // (64-output slabs that are not resident -- 128^2 128->64 -- take the 32-row shape: 251 -> 239 us since the butterfly epilogue)
      // (FI_WS2_PLAIN=1: also the batched launches whose source is already an activation -- the pooled DownBlock inputs that
      //  fi_bn_act_pool_groups wrote out)
      static const long ws2_plain = env_long("FI_WS2_PLAIN", 0);
      const bool xfok = a.xf == 1 || (ws2_plain && a.xf == 0 && a.gimages > 0);
      const bool wanted = env_ws2() == 2 || g_tune[0] == 7 || (xfok && ok16 && tiles16 * (cout / 128) >= 1024) ||
                          (xfok && ok32 && cout == 64 && tiles32 >= 1024);
      // LDS-DMA GEMM tile (conv_fwd_dma_kernel; conv_dma.h): two 16 x 16-pixel sub-tiles x 128 output channels per workgroup, operands by
      // LDS-DMA, the transform in place in LDS.  fi_conv_tuning(7, 8, ...) forces it wherever it applies.  Measured level with or behind the
      // 16-row form above on every batched layer of unet_lc (round 6, profiles/r06_a_dma_kbench.txt: 64^2 128->128 dropout 158 vs 158 us,
      // 32^2 256->256 164 vs 143, two-source 64^2 256->128 223 vs 211): a stage is bound by the CU's ~18 B/clk L2 -> LDS fill and by the
      // drain of the matrix pipe at every stage barrier, not by the weight slab's re-reads -- so it is NOT a default (FI_DMA = 1 makes it
      // the choice for the batched launches the 16-row form would take, except the one-group head)
      {
        static const long dma_on = env_long("FI_DMA", 0);
        const bool okd = okb && cout % 128 == 0 && cin % 32 == 0;
        const bool forced = g_tune[0] == 7 && g_tune[1] == 8;
        const bool si_shape = cin == 64 && cout / 128 >= 2 && cout / 128 <= 4;
        const bool auto_dma = dma_on && g_tune[0] < 0 && env_ws2() == 1 && xfok && ok16 && !si_shape && tiles16 * (cout / 128) >= 1024;
        if (okd && (forced || auto_dma)) {
          a.w = d->w16;
          a.wrows = rows;
          a.tilesY = fi_cdiv(d->H, 16);
          a.nct = cout / 128;
          const int rc = d->dtype == FI_F16 ? fi_conv_fwd_dma_f16((int)g_tune[3], a, st) : fi_conv_fwd_dma_bf16((int)g_tune[3], a, st);
          if (rc != FI_ERR_UNSUPPORTED) return rc;
          a.w = w_plain, a.wrows = 0, a.tilesY = tilesY_plain, a.nct = nct_plain;
        }
      }
      if (ok32 && wanted && !(g_tune[0] == 7 && g_tune[1] == 8)) {
        int tr = ok16 ? 16 : 32;
        static const long force_tr = env_long("FI_WS2_TR", 0);
        if (force_tr == 16 && ok16) tr = 16;
        if (force_tr == 32) tr = 32;
        if (g_tune[0] == 7 && g_tune[1] == 1 && ok16) tr = 16;             // fi_conv_tuning(7, 1 | 2, ...): 16- / 32-row tiles
        if (g_tune[0] == 7 && g_tune[1] == 2) tr = 32;
        a.w = d->w16;
        a.wrows = rows;
        a.tilesY = fi_cdiv(d->H, tr);
        a.nct = cout / (tr == 16 ? 128 : 64);
        // one 64-channel pixel group and 2..4 slabs of 128 (the auxiliary head, 64 -> 512): the slabs inside the tile (form 5;
        // FI_WS2_SI=0 keeps the slab-major order)
        static const long si_on = env_long("FI_WS2_SI", 1);
        const bool si = tr == 16 && si_on && cin == 64 && a.nct >= 2 && a.nct <= 4;
        const int form = si ? 5 : (tr == 16 ? 1 : 2);
        // measured (84 x 128^2 64->512, statistics only, same box): slab-major 872 us, slabs inside the tile 828, and 780 with runs of
        // half the length (512 workgroups queued on 256 CUs: the run lengths differ by up to 2x between CUs)
        const int wgs = g_tune[3] ? (int)g_tune[3] : (si ? 2 : 0);
        const int rc = d->dtype == FI_F16 ? fi_conv_fwd_ws2_f16(form, wgs, a, st) : fi_conv_fwd_ws2_bf16(form, wgs, a, st);
        if (rc != FI_ERR_UNSUPPORTED) return rc;
        a.w = w_plain, a.wrows = 0, a.tilesY = tilesY_plain, a.nct = nct_plain;      //  fall through to the forms below
      }
    }
  ws2_done:
    // wave-specialised form (conv_fwd_ws_kernel): 16-bit storage, 3x3, whole-vector channel counts, plain epilogue, 32-bit
    // byte offsets into every tensor
    {
      const long big = (long)d->N * d->H * d->W * 2;
      const bool fits = !f32 && d->ksize == 3 && d->c0 % 8 == 0 && d->c1 % 8 == 0 && cin >= 32 && cout >= 32 && plain && d->H >= 8 &&
                        d->co0 % 8 == 0 && d->co1 % 8 == 0 && big * d->c0 * (a.xf == 2 ? 4 : 1) < (1L << 32) && big * d->c1 < (1L << 32) &&
                        big * d->co0 < (1L << 32) && big * d->co1 < (1L << 32) && (long)cout * 9 * cin * 2 < (1L << 32);
      // Chosen by default for the batched fused launches (the K-1 LC forwards of an iteration in one launch: >= 1024
      // (tile, slab) items keep the persistent grid evenly loaded) with the transforming loader, 32+ channels in and out:
      // 1.1-1.33x the one-tile kernel on every such layer of unet_lc (profiles/r02_k_kbench2_ws.txt); pooled sources and the
      // 12-image launches of the gradient path measured behind it and stay where they were.
      if (depth > 0) {
        // one-launch 3x3x3 convolution (fi_conv3d_*): the depth taps are channel groups of the contraction; 16-output layers take
        // a half-filled 32-channel slab (they are memory-bound)
        const int cr = d->c0 + d->c1;
        const bool ok3 = !f32 && d->ksize == 3 && a.xf == 0 && plain && d->c0 % 8 == 0 && d->c1 % 8 == 0 && cr >= 16 && cout >= 16 &&
                         d->co0 % 8 == 0 && d->co1 % 8 == 0 && d->H >= 8 && d->N % depth == 0 &&
                         big * d->c0 < (1L << 32) && big * d->c1 < (1L << 32) && big * d->co0 < (1L << 32) &&
                         big * d->co1 < (1L << 32) && (long)cout * 27 * cr * 2 < (1L << 32);
        if (!ok3) return FI_ERR_UNSUPPORTED;
        static const long e_nf = env_long("FI_WS3D_NF", 0), e_pw = env_long("FI_WS3D_PW", 0), e_ck = env_long("FI_WS3D_CK", 0),
                          e_wgs = env_long("FI_WS3D_WGS", 0);      // measurement knobs (tools/c3g_bench.py)
        // Output slab of a workgroup (16, 32 or 64 channels).  A stage fills a 20.7 KB halo tile and 9.2 KB of filter per 16 outputs
        // through the same ~20 B/clk path, and at the levels below 64^3 there are fewer (tile, slab) items than CUs: the slab
        // that minimises (waves of 256 items) x (stage bytes) -- 64 channels at 64^3 / 32^3, 16 at 16^3 and 8^3 (8^3 256 -> 256
        // 42.5 -> 24.3 us, 16^3 128 -> 128 24.7 -> 16.0; tools/c3g_bench.py, profiles/r06_d_c3g.txt)
        int n4 = 4;
        {
          const long tiles3 = (long)d->N * fi_cdiv(d->H, 16) * fi_cdiv(d->W, 16);
          long best = -1;
          for (int nf = 4; nf >= 1; nf >>= 1) {
            const long items = tiles3 * fi_cdiv(cout, nf * 16), cost = ((items + 255) / 256) * (207 + 92 * nf);
            if (best < 0 || cost < best) best = cost, n4 = nf;
          }
        }
        if (e_nf) n4 = (int)e_nf;
        a.depth = depth;
        a.tilesY = fi_cdiv(d->H, 16);
        a.nct = fi_cdiv(cout, n4 * 16);
        const int pw = e_pw ? (int)e_pw : (n4 == 4 ? 44 : 8);
        const int ck = e_ck ? (int)e_ck : 32;
        return d->dtype == FI_F16 ? fi_conv_fwd_ws_f16(n4, ck, pw, (int)e_wgs, a, st) : fi_conv_fwd_ws_bf16(n4, ck, pw, (int)e_wgs, a, st);
      }
      const long items = (long)d->N * fi_cdiv(d->H, 16) * fi_cdiv(d->W, 16) * fi_cdiv(cout, cout > 32 ? 64 : 32);
      const bool ws_auto = v2 == 2 && a.xf == 1 && items >= 1024;
      // ... and for the plain launches of the gradient path (forward, dgrad, ALA: 12 images) with 64+ channels in and out:
      // 4 + 8 waves, 32-channel slabs below 256 outputs -- 1.05x (128^2) to 1.2-1.56x (64^2, 32^2) the one-tile kernel, ahead
      // of the persistent form on the layers that one had (profiles/r02_o_kbench2_ws_plain.txt)
      // (128+ input channels: from 48 items on -- the 16^2 / 32^2 maps of a 256^2 input, 1.1-1.66x; 64..127: from 768 items on,
      //  at fewer the one-tile kernel is level or ahead)
      const long tiles2 = (long)d->N * fi_cdiv(d->H, 16) * fi_cdiv(d->W, 16);
      const long items_nf2 = tiles2 * fi_cdiv(cout, 32), items_nf4 = tiles2 * fi_cdiv(cout, 64);
      const bool ws_plain = v2 == 2 && a.xf == 0 && cout >= 64 &&
                            ((cin >= 128 && items_nf2 >= 48) || (cin >= 64 && items_nf2 >= 768));
      if (fits && (v2 == 4 || v2 == 5 || v2 == 6 || ws_auto || ws_plain)) {
        // 8 consumer + 2 x 4 producer waves where the tile is wide enough to feed them (64+ outputs, measured 3-10 % ahead
        // of 4 + 8); 32-output slabs and the statistics-only head measured ahead with 4 + 8
        const int pw = v2 == 4 ? 4 : (v2 == 6 ? 44 : (v2 == 5 ? 8 : ((cout >= 64 && y0 && !ws_plain) ? 44 : 8)));
        int n4 = cout > 32 ? 4 : 2;
        if (ws_plain && v2 == 2) n4 = (cout >= 256 && items_nf4 >= 192) ? 4 : 2;
        if (v2_nf == 2 || v2_nf == 4) n4 = (int)v2_nf;
        int c4 = (a.xf != 2 && d->c0 % 16 == 0 && d->c1 % 16 == 0) ? 32 : 16;
        if (v2_ck && a.xf != 2) c4 = (int)v2_ck;
        a.tilesY = fi_cdiv(d->H, 16);
        a.nct = fi_cdiv(cout, n4 * 16);
        return d->dtype == FI_F16 ? fi_conv_fwd_ws_f16(n4, c4, pw, (int)v2_wgs, a, st)
                                  : fi_conv_fwd_ws_bf16(n4, c4, pw, (int)v2_wgs, a, st);
      }
    }
    // thin-layer form (conv_thin_kernel): the whole filter in registers -- Cin <= 32, Cout <= 32, one destination.
    // Chosen by default wherever it applies: tools/kbench2.py --thin measures it ahead of the one-tile kernel on every such
    // layer (profiles/r02_n_kbench2_thin.txt: 16 outputs 1.25-1.7x, 32 outputs 1.07-1.47x, plain and batched launches alike)
    {
      const bool fits = !f32 && d->ksize == 3 && d->c0 % 8 == 0 && d->c1 % 8 == 0 && cin >= 16 && cin <= 32 && cout <= 32 &&
                        cout % 8 == 0 && d->co1 == 0 && plain && d->H >= 8 &&
                        (long)d->N * d->H * d->W * (cin > cout ? cin : cout) * 2 * (a.xf == 2 ? 4 : 1) < (1L << 32);
      const int n3 = cout > 16 ? 2 : 1;
      const int c3 = (cin > 16 && a.xf != 2) ? 32 : 16;
      const bool thin_auto = v2 == 2;
      if (fits && cin <= c3 && (v2 == 3 || thin_auto)) {
        a.tilesY = fi_cdiv(d->H, 16);
        a.nct = 1;
        const int wgs = v2_wgs ? (int)v2_wgs : (n3 == 1 ? 8 : 2);
        return d->dtype == FI_F16 ? fi_conv_thin_f16(n3, c3, wgs, a, st) : fi_conv_thin_bf16(n3, c3, wgs, a, st);
      }
    }
    // default (-1 / FI_V2 unset): where tools/kbench2.py measured it ahead -- the channel-rich plain launches (grad-path forward,
    // dgrad, ALA: 64^2 128->128 32.7 -> 26.6 us, 256->128 58.6 -> 43.8 us at 12 images); the batched fused launches stay
    // with the one-tile kernel (profiles/r02_c_kbench2.txt)
    const bool v2_auto = v2 == 2 && a.xf == 0 && cin >= 128 && cout >= 128;
    if ((v2 == 1 || v2_auto) && !f32 && d->ksize == 3 && d->c0 % 8 == 0 && d->c1 % 8 == 0 && cin >= 16 && plain && d->H >= 8) {
      int n2 = v2_auto ? 2 : (cout > 32 ? 4 : (cout > 16 ? 2 : 1));
      if (v2_nf) n2 = (int)v2_nf;
      while (n2 > 1 && (n2 / 2) * 16 >= cout) n2 /= 2;
      int c2 = (cin >= 32 && a.xf != 2 && d->c0 % 16 == 0 && d->c1 % 16 == 0) ? 32 : 16;
      if (v2_ck && a.xf != 2) c2 = (int)v2_ck;
      a.tilesY = fi_cdiv(d->H, 16);
      a.nct = fi_cdiv(cout, n2 * 16);
      const int wgs = v2_wgs ? (int)v2_wgs : (n2 == 1 || v2_auto ? 4 : (n2 == 2 ? 3 : 2));
      return d->dtype == FI_F16 ? fi_conv_fwd_v2_f16_k3(n2, c2, wgs, a, st) : fi_conv_fwd_v2_bf16_k3(n2, c2, wgs, a, st);
    }
  }
  if (f32) return d->ksize == 3 ? fi_conv_fwd_f32_k3(th, nf, ck, a, st) : fi_conv_fwd_f32_k1(th, nf, ck, a, st);
  if (d->dtype == FI_F16) return d->ksize == 3 ? fi_conv_fwd_f16_k3(th, nf, ck, a, st) : fi_conv_fwd_f16_k1(th, nf, ck, a, st);
  return d->ksize == 3 ? fi_conv_fwd_bf16_k3(th, nf, ck, a, st) : fi_conv_fwd_bf16_k1(th, nf, ck, a, st);
}

// Work decomposition of wgrad: channel tiles (16*nfo couts x 16*nfi cins) x `sb` spatial workgroups, each
// walking ntiles/sb pixel tiles with register accumulators (~4 workgroups per CU in total).
struct WgradPlan {
  int quad;             // 1: 32x32 channel tile, one 16x16 quadrant per wave (Cin, Cout >= 32)
  int nfo, nfi, nco, nci, th, tilesX, tilesY, sb;
  size_t part_stride;
  int rows;             // 1: the row-streaming kernel (wgrad_rows.h) when the caller gives a workspace: sb_rows items, one slice each
  int ws, strips, rpw, chunks, sb_rows;
  int narrow;           // rows == 1: 1 = <= 4 input channels, 2 = <= 4 gradient channels (conv_wgrad_rows_kernel<NARROW>), the other side 16
  int tco, tci, nct, nit;   // rows == 3 (conv_wgrad_rows64_kernel): channel tile in 16-channel blocks, tiles per side
};
// FI_WGRAD_ROWS (default 1) / fi_wgrad_tuning bit 0: the row-streaming kernel on the thin 3x3 layers it covers;
// FI_WGRAD_ROWS64 (default 1) / bit 1: its 64 x 64-channel-tile form on the channel-rich 3x3 layers
static long g_wgrad_rows = -1;
static bool wgrad_rows_on() {
  static long v = env_long("FI_WGRAD_ROWS", 1);
  return g_wgrad_rows >= 0 ? (g_wgrad_rows & 1) != 0 : v != 0;
}
static bool wgrad_rows64_on() {
  static long v = env_long("FI_WGRAD_ROWS64", 1);
  return g_wgrad_rows >= 0 ? (g_wgrad_rows & 2) != 0 : v != 0;
}
extern "C" int fi_wgrad_tuning(int rows) {
  g_wgrad_rows = rows;
  return 0;
}
static int plan_wgrad(const FiConv* d, WgradPlan* p, int depth = 0) {
  if (!d) return FI_ERR_NULL;
  if (d->dtype != FI_F32 && d->dtype != FI_BF16 && d->dtype != FI_F16) return FI_ERR_DTYPE;
  if (d->ksize != 1 && d->ksize != 3) return FI_ERR_UNSUPPORTED;
  if (d->N < 1 || d->H < 1 || d->W < 1 || d->c0 < 1 || d->c1 < 0 || d->co0 < 1) return FI_ERR_SHAPE;
  const int cin = (depth > 0 ? 3 : 1) * (d->c0 + d->c1), cout = d->co0;       // depth > 0: the depth taps are channel groups
  p->quad = (cin >= 32 && cout >= 32) ? 1 : 0;
  if (p->quad) {
    p->nfo = p->nfi = 2;
  } else {
    p->nfo = 1;
    p->nfi = cin > 16 ? 2 : 1;
  }
  p->nco = fi_cdiv(cout, p->nfo * 16);
  p->nci = fi_cdiv(cin, p->nfi * 16);
  p->th = pick_th(d->N, d->H, d->W, (long)p->nco * p->nci);
  if (p->quad && d->dtype == FI_F32 && p->th > 8) p->th = 8;   // fp32 quad tiles: 16 rows would exceed 64 KB LDS
  p->tilesX = fi_cdiv(d->W, 16);
  p->tilesY = fi_cdiv(d->H, p->th);
  const long ntiles = (long)d->N * p->tilesX * p->tilesY;
  // ~2 workgroups per CU in total; every spatial workgroup costs one |dw| slice of workspace traffic
  long sb = (p->quad ? wgrad_blocks() : wgrad_blocks_thin()) / ((long)p->nco * p->nci);
  if (sb < 1) sb = 1;
  if (sb > ntiles) sb = ntiles;
  p->sb = (int)sb;
  p->part_stride = (((size_t)cout * d->ksize * d->ksize * cin + cout) + 3) & ~(size_t)3;   // 16-B rows for the reducer
  // thin 3x3 layers on large maps: whole rows streamed through an LDS ring instead of 16 x 16-pixel tiles (wgrad_rows.h)
  p->rows = 0;
  p->narrow = 0;
  if (narrow_on() && depth == 0 && d->c1 == 0) {             // first convolution (1 / 3 -> 16), logits convolution (16 -> n_class)
    if (d->c0 <= 4 && cout == 16) p->narrow = 1;
    if (d->c0 == 16 && cout <= 4) p->narrow = 2;
  }
  static const long pair32 = env_long("FI_WGRAD_ROWS_PAIR", 1);     // 32 -> 32: the pair-per-wave form of conv_wgrad_rows_kernel (0: the quadrant tiles)
  const bool thin16 = (cout == 16 || cout == 32) && (cin == 16 || cin == 32) && (cin == 16 || cout == 16 || pair32) && d->c0 % 16 == 0 && d->c1 % 16 == 0;
  if (wgrad_rows_on() && depth == 0 && d->dtype != FI_F32 && d->ksize == 3 && (thin16 || p->narrow) && d->W % 32 == 0 &&
      (d->W <= 256 || d->W % 256 == 0) && d->H >= 8 && (long)d->N * d->H * d->W >= (1L << 17)) {
    p->rows = 1;
    p->ws = d->W <= 256 ? d->W : 256;
    if (cin == 32 && cout == 32 && p->ws > 128) p->ws = 128;   // (pair form: 49 KB of LDS, three workgroups per CU)
    p->strips = d->W / p->ws;
    // ~3 workgroups per CU (16 input channels: 49 KB of LDS each), ~1.5 with 32 (66 KB; measured 67 us at 384 items against
    // 78 at 768 on 12 x 512^2 32 -> 16); a run is at least FI_WGRAD_ROWS_MINR rows (two dy halo rows per run)
    static const long items_env = env_long("FI_WGRAD_ROWS_ITEMS", 0), minr = env_long("FI_WGRAD_ROWS_MINR", 16);
    const long items_target = items_env > 0 ? items_env : ((cin == 16 || p->narrow || (cin == 32 && cout == 32)) ? 768 : 384);
    long rpw = ((long)d->N * p->strips * d->H + items_target - 1) / items_target;
    if (rpw < minr) rpw = minr;
    if (rpw > d->H) rpw = d->H;
    p->rpw = (int)rpw;
    p->chunks = fi_cdiv(d->H, p->rpw);
    p->sb_rows = d->N * p->strips * p->chunks;
  }
  // channel-rich 3x3 layers on 32 ... 256-wide maps (32-wide: one K step a row, still 7-12 % ahead of the quadrant tiles): the same streaming with 64 x 64 (64 x 32 / 32 x 64) channel tiles.  Workgroups
  // = items x channel tiles; every ITEM costs one |dw| slice of workspace traffic, so the run length follows the tile count
  // (at most FI_WGRAD_ROWS64_WGS workgroups in all, a run of at least 8 rows)
  if (wgrad_rows64_on() && depth == 0 && p->quad && !p->rows && d->dtype != FI_F32 && d->ksize == 3 && cout % 32 == 0 && cin % 32 == 0 &&
      (cout % 64 == 0 || cin % 64 == 0) && d->c0 % 8 == 0 && d->c1 % 8 == 0 && d->W % 32 == 0 && d->W >= 32 &&
      (d->W <= 128 || d->W % 128 == 0) && d->H >= 8 && (long)d->N * d->H * d->W >= (d->W < 64 ? (1L << 13) : (1L << 15))) {
    static const long wgs_env = env_long("FI_WGRAD_ROWS64_WGS", 512), tile_env = env_long("FI_WGRAD_ROWS64_TILE", 0);
    // tile (gradient x input side, 16-channel blocks): 32 x 64 wherever the input side allows it, else 64 x 32 -- 216 / 228 registers,
    // two workgroups per CU.  The 64 x 64 tile (348 registers: ONE wave per SIMD, nothing hides its LDS latency) measured level on
    // one-tile layers and behind on the others (profiles/r04_wgbench_rows64.txt: 128^2 128 -> 64 68 us against 55, 64^2 128 -> 128
    // 49 against 40); FI_WGRAD_ROWS64_TILE=44 keeps it for measurements
    p->tci = cin % 64 == 0 ? 4 : 2;
    p->tco = p->tci == 4 ? 2 : 4;
    if (tile_env == 44 && cout % 64 == 0 && cin % 64 == 0) p->tco = p->tci = 4;
    if (tile_env == 42 && cout % 64 == 0) p->tco = 4, p->tci = 2;
    p->rows = 3;
    p->nct = cout / (p->tco * 16);
    p->nit = cin / (p->tci * 16);
    p->ws = d->W <= 128 ? d->W : 128;
    p->strips = d->W / p->ws;
    // one full wave of workgroups (two per CU) and no second, partial one: the run length is the shortest that keeps
    // items x tiles <= FI_WGRAD_ROWS64_WGS (profiles/r04_wgbench_rows64_wgs.txt: 456-512 workgroups 546 us per body-phase iteration,
    // 312-384: 594, 624-672: 585)
    long chunks_max = wgs_env / ((long)p->nct * p->nit * d->N * p->strips);
    if (chunks_max < 1) chunks_max = 1;
    long rpw = (d->H + chunks_max - 1) / chunks_max;
    if (rpw < 8) rpw = 8;
    if (rpw > d->H) rpw = d->H;
    p->rpw = (int)rpw;
    p->chunks = fi_cdiv(d->H, p->rpw);
    p->sb_rows = d->N * p->strips * p->chunks;
  }
  // ... and the 3x3x3 layers on slices >= 32 wide (conv_wgrad_rows3d_kernel): Cout = 16 with c0 + c1 = 16 / 32 / 48 on slices up to 128
  // wide (one tile of 1 x 1 ... 3 blocks), or Cout a multiple of 32 with c0 + c1 = 16 or a multiple of 32 on slices up to 64 wide
  // (tiles of 2 gradient x 1 / 2 input blocks; bit 1 of the switch)
  const int cr = d->c0 + d->c1;
  const bool thin3 = cout == 16 && (cr == 16 || cr == 32 || cr == 48) && d->W <= 128 && (long)d->N * d->H * d->W >= (1L << 17);
  const bool wide3 = wgrad_rows64_on() && cout % 32 == 0 && (cr == 16 || cr % 32 == 0) && d->W <= 64 && (long)d->N * d->H * d->W >= (1L << 16);
  if (wgrad_rows_on() && depth > 0 && d->dtype != FI_F32 && d->ksize == 3 && (thin3 || wide3) &&
      d->c0 % 16 == 0 && d->c1 % 16 == 0 && d->W % 32 == 0 && d->H >= 8 && d->N % depth == 0) {
    p->rows = 2;
    p->ws = d->W;
    p->strips = 1;
    p->tco = thin3 ? 1 : 2;                                   // blocks of a workgroup's tile
    p->tci = thin3 ? cr / 16 : (cr == 16 ? 1 : 2);
    p->nct = cout / (p->tco * 16);
    p->nit = cr / (p->tci * 16);
    static const long items3 = env_long("FI_WGRAD_ROWS3D_ITEMS", 512);          // ~2 workgroups per CU (56 ... 73 KB of LDS)
    long rpw = ((long)d->N * d->H * p->nct * p->nit + items3 - 1) / items3;
    if (rpw < 16) rpw = 16;
    if (rpw > d->H) rpw = d->H;
    p->chunks = fi_cdiv(d->H, (int)rpw);
    p->rpw = fi_cdiv(d->H, p->chunks);                        // even runs
    p->sb_rows = d->N * p->chunks;
  }
  return 0;
}

extern "C" long fi_conv2d_wgrad_workspace(const FiConv* d) {
  WgradPlan p;
  const int rc = plan_wgrad(d, &p);
  if (rc) return rc;
  return (long)(p.part_stride * (p.rows && p.sb_rows > p.sb ? p.sb_rows : p.sb) * sizeof(float));
}

static int wgrad_impl(const FiConv* d, const void* x0, const void* x1, const void* dy, float* dw, float* dbias,
                      void* workspace, long workspace_bytes, int reduce_now, int* slices_out, long* stride_out,
                      void* stream, int depth = 0);

extern "C" int fi_conv2d_wgrad(const FiConv* d, const void* x0, const void* x1, const void* dy, float* dw,
                               float* dbias, void* workspace, long workspace_bytes, void* stream) {
  return wgrad_impl(d, x0, x1, dy, dw, dbias, workspace, workspace_bytes, 1, nullptr, nullptr, stream);
}

extern "C" int fi_conv2d_wgrad_partial(const FiConv* d, const void* x0, const void* x1, const void* dy, int want_bias,
                                       void* workspace, long workspace_bytes, int* slices, long* stride,
                                       void* stream) {
  if (!workspace || !slices || !stride) return FI_ERR_NULL;
  // dw / dbias are only used as "is requested" flags when the reduction is deferred
  return wgrad_impl(d, x0, x1, dy, (float*)workspace, want_bias ? (float*)workspace : nullptr, workspace,
                    workspace_bytes, 0, slices, stride, stream);
}

// table rows (int64 x 7): { partial ptr, slice stride (floats), slices, dw ptr, n_dw, dbias ptr or 0, cout }
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const long long* __restrict__ table, int ntensors) {
  // One launch folds every layer's partial slices (278 MB per U-Net step) into the gradient buffer.  Rows of
  // FI_WGRAD_ROW int64: {part, stride, slices, dw, n_dw, dbias|0, cout, first_block, log2(lanes), cin3}.  A block owns
  // 4*lanes consecutive elements of one tensor; its 256/lanes thread groups take the slices round-robin with 16-B loads
  // (rows are 16-B aligned: stride % 4 == 0), and the groups are then folded through LDS in a fixed order.
  __shared__ float sm[256 * 4];
  int row = 0;
  for (int t = 1; t < ntensors; ++t)
    if ((long long)blockIdx.x >= table[(size_t)t * FI_WGRAD_ROW + 7]) row = t;
  const long long* r = table + (size_t)row * FI_WGRAD_ROW;
  const float* part = reinterpret_cast<const float*>(r[0]);
  const size_t stride = (size_t)r[1];
  const int slices = (int)r[2];
  float* dw = reinterpret_cast<float*>(r[3]);
  const size_t n_dw = (size_t)r[4];
  float* dbias = reinterpret_cast<float*>(r[5]);
  const size_t n = n_dw + (dbias ? (size_t)r[6] : 0);
  const int ll = (int)r[8], lanes = 1 << ll, groups = 256 >> ll;
  // cin3 != 0: the slices are a one-launch 3x3x3 gradient [cout][9][3][|cin3|] and dw is the PARAMETER layout [cout][|cin3|][3][3][3].
  // > 0: permuted here (4-byte read-modify-writes 108 B apart: 202 us per unet_3D iteration against 77 for the sums alone);
  // < 0: the sums are left in slice 0, in the slices' layout, for fi_wgrad_permute3d_multi's transposed, coalesced add
  const long long cin3s = r[9];
  const size_t cin3 = (size_t)(cin3s < 0 ? -cin3s : cin3s);
  const int e = threadIdx.x & (lanes - 1), g = threadIdx.x >> ll;
  const size_t base = ((size_t)blockIdx.x - (size_t)r[7]) * (size_t)(lanes * 4);
  const size_t i = base + (size_t)e * 4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (i < stride) {
    const float4* p = reinterpret_cast<const float4*>(part + i);
    const size_t s4 = stride / 4;
    int k = g;
    for (; k + groups < slices; k += 2 * groups) {
      const float4 a = p[(size_t)k * s4], b = p[(size_t)(k + groups) * s4];
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
    }
    if (k < slices) {
      const float4 a = p[(size_t)k * s4];
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
  }
  float* mine = sm + (size_t)g * (lanes * 4) + e * 4;
  mine[0] = s0.x + s1.x; mine[1] = s0.y + s1.y; mine[2] = s0.z + s1.z; mine[3] = s0.w + s1.w;
  __syncthreads();
  for (int q = threadIdx.x; q < lanes * 4; q += 256) {
    const size_t j = base + q;
    if (j >= n) break;
    float s = sm[q];
    for (int h = 1; h < groups; ++h) s += sm[h * (lanes * 4) + q];
    if (j >= n_dw) {
      dbias[j - n_dw] += s;
    } else if (cin3 == 0) {
      dw[j] += s;
    } else if (cin3s < 0) {
      const_cast<float*>(part)[j] = s;          // this workgroup alone read these elements of slice 0, before the barrier
    } else {
      const size_t ci = j % cin3, q1 = j / cin3, kd = q1 % 3, q2 = q1 / 3, t = q2 % 9, co = q2 / 9;
      dw[(co * cin3 + ci) * 27 + kd * 9 + t] += s;
    }
  }
}

extern "C" int fi_wgrad_reduce_multi(const long long* table, int ntensors, int nblocks, void* stream) {
  if (!table) return FI_ERR_NULL;
  if (ntensors <= 0 || nblocks <= 0) return 0;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, table, ntensors);
  FI_CHECK_LAUNCH();
  return 0;
}

// Second half of the deferred 3x3x3 reduce (table rows with cin3 < 0): sums [cout][9][3][cin] in slice 0 -> dw [cout][cin][3][3][3].
// A workgroup owns one output channel x <= 64 input channels: 27 runs of <= 256 B in, one run of <= 6912 B out, transposed in LDS.
__global__ __launch_bounds__(256) void wgrad_permute3d_multi_kernel(const long long* __restrict__ table, int ntensors) {
  __shared__ float sm[27][65];
  int row = -1;
  for (int t = 0; t < ntensors; ++t) {
    const long long* q = table + (size_t)t * FI_WGRAD_ROW;
    if (q[9] < 0 && (long long)blockIdx.x >= q[10]) row = t;
  }
  if (row < 0) return;
  const long long* r = table + (size_t)row * FI_WGRAD_ROW;
  const float* sum = reinterpret_cast<const float*>(r[0]);
  float* dw = reinterpret_cast<float*>(r[3]);
  const int cin = (int)-r[9], cout = (int)r[6], nchunk = (cin + 63) / 64;
  const int item = (int)((long long)blockIdx.x - r[10]), co = item / nchunk, ci0 = (item % nchunk) * 64;
  if (co >= cout) return;
  const int cw = cin - ci0 < 64 ? cin - ci0 : 64;
  for (int q = threadIdx.x; q < 27 * 64; q += 256) {
    const int tk = q >> 6, c = q & 63;                       // tk = t * 3 + kd (t = ky * 3 + kx): the slices' tap order
    if (c < cw) sm[tk][c] = sum[((size_t)co * 27 + tk) * cin + ci0 + c];
  }
  __syncthreads();
  float* out = dw + ((size_t)co * cin + ci0) * 27;
  for (int q = threadIdx.x; q < 27 * cw; q += 256) {
    const int c = q / 27, p = q - c * 27, kd = p / 9, t = p - kd * 9;
    out[q] += sm[t * 3 + kd][c];
  }
}

extern "C" int fi_wgrad_permute3d_multi(const long long* table, int ntensors, int nblocks, void* stream) {
  if (!table) return FI_ERR_NULL;
  if (ntensors <= 0 || nblocks <= 0) return 0;
  hipLaunchKernelGGL(wgrad_permute3d_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, table, ntensors);
  FI_CHECK_LAUNCH();
  return 0;
}

static int wgrad_impl(const FiConv* d, const void* x0, const void* x1, const void* dy, float* dw, float* dbias,
                      void* workspace, long workspace_bytes, int reduce_now, int* slices_out, long* stride_out,
                      void* stream, int depth) {
  if (!d || !x0 || !dy || !dw) return FI_ERR_NULL;
  WgradPlan p;
  const int rc = plan_wgrad(d, &p, depth);
  if (rc) return rc;
  if (d->c1 > 0 && !x1) return FI_ERR_NULL;
  const int cin = (depth > 0 ? 3 : 1) * (d->c0 + d->c1), cout = d->co0;
  if (!workspace) p.rows = 0;                               // (atomic accumulation: the tile kernels)
  const int sb_tile = p.sb;                                 // slices of the tile kernels' plan: what a declined rows launch falls back to
  if (p.rows) p.sb = p.sb_rows;
  if (workspace && workspace_bytes < (long)(p.part_stride * p.sb * sizeof(float))) return FI_ERR_SHAPE;
  WgradArgs a;
  a.x0 = x0;
  a.x1 = x1 ? x1 : x0;
  a.dy = dy;
  a.dw = dw;
  a.dbias = dbias;
  a.part = (float*)workspace;
  a.part_stride = p.part_stride;
  a.N = d->N;
  a.H = d->H;
  a.W = d->W;
  a.c0 = d->c0;
  a.c1 = d->c1;
  a.cout = cout;
  a.tilesX = p.tilesX;
  a.tilesY = p.tilesY;
  a.nco = p.nco;
  a.nci = p.nci;
  a.spatialBlocks = p.sb;
  a.depth = depth;
#ifdef FI_TRACE
  a.trace = g_trace;
#endif
  hipStream_t st = (hipStream_t)stream;
  int r = FI_ERR_UNSUPPORTED;
  if (p.rows) {
    WgRowsArgs ra;
    ra.x0 = x0, ra.x1 = x1 ? x1 : x0, ra.dy = dy;
    ra.part = (float*)workspace, ra.part_stride = p.part_stride;
    ra.N = d->N, ra.H = d->H, ra.W = d->W, ra.c0 = d->c0, ra.c1 = d->c1;
    ra.ws = p.ws, ra.strips = p.strips, ra.rpw = p.rpw, ra.chunks = p.chunks;
    ra.want_bias = dbias != nullptr;
    ra.depth = depth;
    ra.cd = cout;
    ra.cout = cout, ra.nct = p.nct, ra.nit = p.nit;
#ifdef FI_TRACE
    ra.trace = g_trace;
#endif
    if (p.rows == 3)
      r = d->dtype == FI_F16 ? fi_conv_wgrad_rows64_f16(p.tco, p.tci, ra, p.sb, st) : fi_conv_wgrad_rows64_bf16(p.tco, p.tci, ra, p.sb, st);
    else if (p.rows == 1 && p.narrow)
      r = d->dtype == FI_F16 ? fi_conv_wgrad_rows_narrow_f16(p.narrow, ra, p.sb, st) : fi_conv_wgrad_rows_narrow_bf16(p.narrow, ra, p.sb, st);
    else if (p.rows == 2)
      r = d->dtype == FI_F16 ? fi_conv_wgrad_rows3d_f16(p.tci, p.tco, ra, p.sb, st) : fi_conv_wgrad_rows3d_bf16(p.tci, p.tco, ra, p.sb, st);
    else
      r = d->dtype == FI_F16 ? fi_conv_wgrad_rows_f16(cin / 16, cout / 16, ra, p.sb, st)
                             : fi_conv_wgrad_rows_bf16(cin / 16, cout / 16, ra, p.sb, st);
    if (r == FI_ERR_UNSUPPORTED) {
      // a row-streaming launcher declined (LDS beyond what the device grants: ADVICE r4): nothing was launched -- the tile
      // kernels' plan takes the layer (its slice count fits the same workspace: fi_conv2d_wgrad_workspace sizes for both)
      p.rows = 0;
      p.sb = sb_tile;
      a.spatialBlocks = sb_tile;
      if (workspace && workspace_bytes < (long)(p.part_stride * p.sb * sizeof(float))) return FI_ERR_SHAPE;
    }
  }
  if (p.rows) {
    // (launched above)
  } else if (p.quad) {
    if (d->dtype == FI_F32)
      r = d->ksize == 3 ? fi_conv_wgrad_quad_f32_k3(p.th, a, st) : fi_conv_wgrad_quad_f32_k1(p.th, a, st);
    else if (d->dtype == FI_F16)
      r = d->ksize == 3 ? fi_conv_wgrad_quad_f16_k3(p.th, a, st) : fi_conv_wgrad_quad_f16_k1(p.th, a, st);
    else
      r = d->ksize == 3 ? fi_conv_wgrad_quad_bf16_k3(p.th, a, st) : fi_conv_wgrad_quad_bf16_k1(p.th, a, st);
  } else if (d->dtype == FI_F32) {
    r = d->ksize == 3 ? fi_conv_wgrad_f32_k3(p.th, p.nfo, p.nfi, a, st) : fi_conv_wgrad_f32_k1(p.th, p.nfo, p.nfi, a, st);
  } else if (d->dtype == FI_F16) {
    r = d->ksize == 3 ? fi_conv_wgrad_f16_k3(p.th, p.nfo, p.nfi, a, st) : fi_conv_wgrad_f16_k1(p.th, p.nfo, p.nfi, a, st);
  } else {
    r = d->ksize == 3 ? fi_conv_wgrad_bf16_k3(p.th, p.nfo, p.nfi, a, st)
                      : fi_conv_wgrad_bf16_k1(p.th, p.nfo, p.nfi, a, st);
  }
  if (slices_out) *slices_out = p.sb;
  if (stride_out) *stride_out = (long)p.part_stride;
  if (r || !workspace || !reduce_now) return r;
  const size_t n_dw = (size_t)cout * d->ksize * d->ksize * cin;
  const size_t n = n_dw + (dbias ? cout : 0);
  int grid = (int)((n + 31) / 32);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float*)workspace, p.part_stride, p.sb,
                     dw, n_dw, dbias, cout);
  FI_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// 3D convolution (3x3x3 pad 1, or 1x1x1) over dense NDHWC volumes: a volume is D consecutive NHWC slices, so depth tap kd
// is ONE 2D implicit-GEMM launch over the slices it reaches, accumulating into the output volume.  The centre tap goes
// last -- it covers every slice, so its epilogue sees the finished sums (bias, per-sample statistics).
// d describes a slice batch: N = samples, H, W, ksize (in-plane = depth extent), channel split; D = depth.
// ---------------------------------------------------------------------------------------------
static inline size_t fi_esz(int dtype) { return dtype == FI_F32 ? 4 : 2; }
struct FiTap { int t, lo, hi, off; };
static int fi_taps(int ksize, int D, FiTap* taps) {
  if (ksize == 1) {
    taps[0] = FiTap{0, 0, D, 0};
    return 1;
  }
  int n = 0;
  const int order[3] = {0, 2, 1};
  for (int i = 0; i < 3; ++i) {
    const int kd = order[i], lo = kd == 0 ? 1 : 0, hi = D + 1 - kd < D ? D + 1 - kd : D;
    if (hi > lo) taps[n++] = FiTap{kd, lo, hi, kd - 1};
  }
  return n;
}

extern "C" int fi_conv3d_fwd(const FiConv* d, int D, const void* x0, const void* x1, const void* const* w_taps,
                             const float* bias, void* y, double* stats, long stats_stride, void* stream) {
  if (!d || !x0 || !w_taps || !y) return FI_ERR_NULL;
  if (D < 1 || d->co1 != 0 || (d->c1 > 0 && !x1)) return FI_ERR_SHAPE;
  FiTap taps[3];
  const int nt = fi_taps(d->ksize, D, taps);
  const size_t es = fi_esz(d->dtype), ys = d->y_f32 ? 4 : es, plane = (size_t)d->H * d->W;
  for (int n = 0; n < d->N; ++n)
    for (int i = 0; i < nt; ++i) {
      const FiTap& t = taps[i];
      FiConv s = *d;
      s.N = t.hi - t.lo;
      s.accumulate0 = 1;
      const size_t in0 = ((size_t)n * D + t.lo + t.off) * plane;
      const char* p0 = (const char*)x0 + in0 * d->c0 * es;
      const char* p1 = d->c1 ? (const char*)x1 + in0 * d->c1 * es : nullptr;
      char* py = (char*)y + ((size_t)n * D + t.lo) * plane * d->co0 * ys;
      const bool centre = t.off == 0;
      const int rc = fi_conv2d_fwd(&s, p0, p1, w_taps[t.t], centre ? bias : nullptr, py, nullptr,
                                   (stats && centre) ? stats + (size_t)n * stats_stride : nullptr, stream);
      if (rc) return rc;
    }
  return 0;
}

// One-launch form (conv_fwd_ws_kernel with depth taps): every slice of every volume is an image of ONE implicit GEMM whose
// contraction runs over 9 in-plane taps x 3 depth taps x channels; w_all = the filter as [Cout][k*k][3][c0 + c1] (the three
// per-tap operands of fi_conv3d_fwd interleaved), y is WRITTEN (no zeroing, no read-modify-write passes), the statistics are
// per volume.  FI_ERR_UNSUPPORTED when the shape is not covered (fp32, 1x1x1, channel counts that are not multiples of 8,
// fewer than 16 channels): the caller falls back to fi_conv3d_fwd.
extern "C" int fi_conv3d_fwd_fused(const FiConv* d, int D, const void* x0, const void* x1, const void* w_all, const float* bias,
                                   void* y, double* stats, long stats_stride, void* stream) {
  if (!d || !x0 || !w_all || !y) return FI_ERR_NULL;
  if (D < 1 || d->co1 != 0 || (d->c1 > 0 && !x1)) return FI_ERR_SHAPE;
  if (d->ksize != 3 || d->accumulate0 || d->accumulate1 || d->y_f32) return FI_ERR_UNSUPPORTED;
  if ((long)d->N * D > 0x7fffffffL) return FI_ERR_UNSUPPORTED;
  {  // the thin 128^3 layers (16 [+ 32] -> 16): one staging of every input slice per tile instead of three (conv3d_stream.hip)
    const int rc = fi_conv3d_stream(d->dtype, d->N, D, d->H, d->W, d->c0, d->c1, d->co0, 0, x0, x1, w_all, bias, y, nullptr, stats,
                                    stats_stride, (hipStream_t)stream);
    if (rc != FI_ERR_UNSUPPORTED) return rc;
  }
  FiConv s = *d;
  s.N = d->N * D;
  return conv_fwd_impl(&s, nullptr, nullptr, D, 0, x0, x1, w_all, bias, y, nullptr, stats, stats_stride, stream, D);
}

// dgrad of the same: d->c0 = channels of dy, d->co0 / co1 = channels of the (possibly concatenated) input; wt_all = the
// flipped / transposed filter as [c_in][k*k][3][Cout] with the depth taps reversed; d0 / d1 are WRITTEN.
extern "C" int fi_conv3d_dgrad_fused(const FiConv* d, int D, const void* dy, const void* wt_all, void* d0, void* d1,
                                     void* stream) {
  if (!d || !dy || !wt_all || !d0) return FI_ERR_NULL;
  if (D < 1 || d->c1 != 0 || (d->co1 > 0 && !d1)) return FI_ERR_SHAPE;
  if (d->ksize != 3 || d->accumulate0 || d->accumulate1 || d->y_f32) return FI_ERR_UNSUPPORTED;
  if ((long)d->N * D > 0x7fffffffL) return FI_ERR_UNSUPPORTED;
  {  // 16 -> 16 and 16 -> (16 + 32): the same streaming kernel on the flipped, transposed filter
    const int rc = fi_conv3d_stream(d->dtype, d->N, D, d->H, d->W, d->c0, 0, d->co0, d->co1, dy, nullptr, wt_all, nullptr, d0, d1, nullptr, 0,
                                    (hipStream_t)stream);
    if (rc != FI_ERR_UNSUPPORTED) return rc;
  }
  FiConv s = *d;
  s.N = d->N * D;
  return conv_fwd_impl(&s, nullptr, nullptr, 0, 0, dy, nullptr, wt_all, nullptr, d0, d1, nullptr, 0, stream, D);
}

// dgrad: d->c0 = channels of dy, d->co0 / co1 = channels of the (possibly concatenated) input; d0 / d1 accumulate
extern "C" int fi_conv3d_dgrad(const FiConv* d, int D, const void* dy, const void* const* wt_taps, void* d0, void* d1,
                               void* stream) {
  if (!d || !dy || !wt_taps || !d0) return FI_ERR_NULL;
  if (D < 1 || d->c1 != 0 || (d->co1 > 0 && !d1)) return FI_ERR_SHAPE;
  FiTap taps[3];
  const int nt = fi_taps(d->ksize, D, taps);
  const size_t es = fi_esz(d->dtype), plane = (size_t)d->H * d->W;
  for (int n = 0; n < d->N; ++n)
    for (int i = 0; i < nt; ++i) {
      const FiTap& t = taps[i];
      FiConv s = *d;
      s.N = t.hi - t.lo;
      s.accumulate0 = s.accumulate1 = 1;
      const char* pdy = (const char*)dy + ((size_t)n * D + t.lo) * plane * d->c0 * es;
      const size_t out0 = ((size_t)n * D + t.lo + t.off) * plane;
      char* p0 = (char*)d0 + out0 * d->co0 * es;
      char* p1 = d->co1 ? (char*)d1 + out0 * d->co1 * es : nullptr;
      const int rc = fi_conv2d_fwd(&s, pdy, nullptr, wt_taps[t.t], nullptr, p0, p1, nullptr, stream);
      if (rc) return rc;
    }
  return 0;
}

extern "C" long fi_conv3d_wgrad_workspace(const FiConv* d, int D) {
  if (!d || D < 1) return FI_ERR_SHAPE;
  FiConv s = *d;
  s.N = D;
  return fi_conv2d_wgrad_workspace(&s);
}

// The 3x3x3 filter gradient as ONE launch over all slices of all volumes (conv_wgrad*_kernel with WgradArgs.depth): the depth
// taps are channel groups of the input side, dw_all fp32 [cout][9][3][c0 + c1] = the layout of fi_conv3d_fwd_fused's operand,
// ADDED to (caller zeroes), dbias fp32 [cout] or NULL.  The per-tap form below is 3 N launches + 3 N slice reductions per
// convolution (40 % of a unet_3D iteration at 2 x 128^3).  Whole-vector channel counts only: FI_ERR_UNSUPPORTED otherwise.
static int conv3d_wgrad_fused_ok(const FiConv* d, int D) {
  if (!d) return FI_ERR_NULL;
  const int vg = d->dtype == FI_F32 ? 4 : 8;
  if (d->ksize != 3 || D < 1 || d->c0 % vg || d->c1 % vg || d->co0 % vg) return FI_ERR_UNSUPPORTED;
  if ((long)d->N * D > 0x7fffffffL) return FI_ERR_SHAPE;
  return 0;
}
extern "C" long fi_conv3d_wgrad_fused_workspace(const FiConv* d, int D) {
  if (int rc = conv3d_wgrad_fused_ok(d, D)) return rc;
  FiConv s = *d;
  s.N = d->N * D;
  WgradPlan p;
  if (int rc = plan_wgrad(&s, &p, D)) return rc;
  return (long)(p.part_stride * (p.rows && p.sb_rows > p.sb ? p.sb_rows : p.sb) * sizeof(float));
}
extern "C" int fi_conv3d_wgrad_fused(const FiConv* d, int D, const void* x0, const void* x1, const void* dy, float* dw_all,
                                     float* dbias, void* workspace, long workspace_bytes, void* stream) {
  if (!d || !x0 || !dy || !dw_all || !workspace) return FI_ERR_NULL;
  if (int rc = conv3d_wgrad_fused_ok(d, D)) return rc;
  if (d->c1 > 0 && !x1) return FI_ERR_NULL;
  FiConv s = *d;
  s.N = d->N * D;
  return wgrad_impl(&s, x0, x1, dy, dw_all, dbias, workspace, workspace_bytes, 1, nullptr, nullptr, stream, D);
}

// Stage 1 of the same only (fi_conv2d_wgrad_partial's 3D sibling): the partial slices stay in `workspace`, *slices / *stride (floats)
// describe them -- [slices][cout*9*3*cin (+ cout bias sums when want_bias)] -- and fi_wgrad_reduce_multi folds them later, for all
// layers of a backward pass in one launch, straight into the parameter's own [cout][cin][3][3][3] layout (table word 9 = cin).
extern "C" int fi_conv3d_wgrad_fused_partial(const FiConv* d, int D, const void* x0, const void* x1, const void* dy, int want_bias,
                                             void* workspace, long workspace_bytes, int* slices, long* stride, void* stream) {
  if (!d || !x0 || !dy || !workspace || !slices || !stride) return FI_ERR_NULL;
  if (int rc = conv3d_wgrad_fused_ok(d, D)) return rc;
  if (d->c1 > 0 && !x1) return FI_ERR_NULL;
  FiConv s = *d;
  s.N = d->N * D;
  return wgrad_impl(&s, x0, x1, dy, (float*)workspace, want_bias ? (float*)workspace : nullptr, workspace, workspace_bytes, 0, slices,
                    stride, stream, D);
}

// dw_taps: fp32 [kd][cout][k][k][cin] (one 2D filter gradient per depth tap), dbias fp32 [cout] or NULL; both ADDED to.
extern "C" int fi_conv3d_wgrad(const FiConv* d, int D, const void* x0, const void* x1, const void* dy, float* dw_taps,
                               float* dbias, void* workspace, long workspace_bytes, void* stream) {
  if (!d || !x0 || !dy || !dw_taps) return FI_ERR_NULL;
  if (D < 1 || (d->c1 > 0 && !x1)) return FI_ERR_SHAPE;
  FiTap taps[3];
  const int nt = fi_taps(d->ksize, D, taps);
  const size_t es = fi_esz(d->dtype), plane = (size_t)d->H * d->W;
  const size_t n_dw = (size_t)d->co0 * d->ksize * d->ksize * (d->c0 + d->c1);
  for (int n = 0; n < d->N; ++n)
    for (int i = 0; i < nt; ++i) {
      const FiTap& t = taps[i];
      FiConv s = *d;
      s.N = t.hi - t.lo;
      const size_t in0 = ((size_t)n * D + t.lo + t.off) * plane;
      const char* p0 = (const char*)x0 + in0 * d->c0 * es;
      const char* p1 = d->c1 ? (const char*)x1 + in0 * d->c1 * es : nullptr;
      const char* pdy = (const char*)dy + ((size_t)n * D + t.lo) * plane * d->co0 * es;
      const int rc = fi_conv2d_wgrad(&s, p0, p1, pdy, dw_taps + (size_t)t.t * n_dw, t.off == 0 ? dbias : nullptr, workspace,
                                     workspace_bytes, stream);
      if (rc) return rc;
    }
  return 0;
}

// ---------------------------------------------------------------- ConvTranspose{2,3}d(kernel 2, stride 2) -----------
// No overlapping taps: ONE 1x1 implicit GEMM to / from P*Cout packed channels ([tap][co]) + the depth-to-space shuffle.
// `packed` is caller-owned scratch of N*D*H*W*P*Cout elements of `dtype`.
static inline FiConv fi_ct_desc(int dtype, long slices, int H, int W, int cin, int cout) {
  return FiConv{dtype, (int)slices, H, W, 1, cin, 0, cout, 0, 0, 0, 0};
}
static inline int fi_ct_check(int N, int D, int H, int W, int cin, int cout, int three_d) {
  if (N < 1 || D < 1 || H < 1 || W < 1 || cin < 1 || cout < 1 || (!three_d && D != 1)) return FI_ERR_SHAPE;
  if ((long)N * D > 0x7fffffffL) return FI_ERR_SHAPE;
  return 0;
}

extern "C" int fi_convtranspose2x_fwd(int dtype, int N, int D, int H, int W, int cin, int cout, int three_d, const void* x,
                                      const void* w_packed, const float* bias_taps, void* packed, void* y, void* stream) {
  if (!x || !w_packed || !packed || !y) return FI_ERR_NULL;
  if (int rc = fi_ct_check(N, D, H, W, cin, cout, three_d)) return rc;
  const int P = three_d ? 8 : 4;
  const FiConv d = fi_ct_desc(dtype, (long)N * D, H, W, cin, P * cout);
  if (int rc = fi_conv2d_fwd(&d, x, nullptr, w_packed, bias_taps, packed, nullptr, nullptr, stream)) return rc;
  return fi_depth_to_space2x(dtype, packed, y, N, D, H, W, cout, three_d, 0, stream);
}

extern "C" int fi_convtranspose2x_dgrad(int dtype, int N, int D, int H, int W, int cin, int cout, int three_d, const void* dy,
                                        const void* wt_packed, void* packed, void* dx, void* stream) {
  if (!dy || !packed) return FI_ERR_NULL;
  if (int rc = fi_ct_check(N, D, H, W, cin, cout, three_d)) return rc;
  const int P = three_d ? 8 : 4;
  if (int rc = fi_depth_to_space2x(dtype, dy, packed, N, D, H, W, cout, three_d, 1, stream)) return rc;
  if (!dx) return 0;                                   // only the packed gradient was wanted (input needs no gradient)
  if (!wt_packed) return FI_ERR_NULL;
  const FiConv d = fi_ct_desc(dtype, (long)N * D, H, W, P * cout, cin);
  return fi_conv2d_fwd(&d, packed, nullptr, wt_packed, nullptr, dx, nullptr, nullptr, stream);
}

extern "C" long fi_convtranspose2x_wgrad_workspace(int dtype, int N, int D, int H, int W, int cin, int cout, int three_d) {
  if (int rc = fi_ct_check(N, D, H, W, cin, cout, three_d)) return rc;
  const FiConv d = fi_ct_desc(dtype, (long)N * D, H, W, cin, (three_d ? 8 : 4) * cout);
  return fi_conv2d_wgrad_workspace(&d);
}

// dy_packed: the packed gradient fi_convtranspose2x_dgrad left in `packed`; dw fp32 [P*Cout][Cin], dbias_taps fp32 [P*Cout]
extern "C" int fi_convtranspose2x_wgrad(int dtype, int N, int D, int H, int W, int cin, int cout, int three_d, const void* x,
                                        const void* dy_packed, float* dw, float* dbias_taps, void* workspace,
                                        long workspace_bytes, void* stream) {
  if (!x || !dy_packed || !dw) return FI_ERR_NULL;
  if (int rc = fi_ct_check(N, D, H, W, cin, cout, three_d)) return rc;
  const FiConv d = fi_ct_desc(dtype, (long)N * D, H, W, cin, (three_d ? 8 : 4) * cout);
  return fi_conv2d_wgrad(&d, x, nullptr, dy_packed, dw, dbias_taps, workspace, workspace_bytes, stream);
}
