// Elementwise glue of the tree-energy losses (/root/reference/code/flower_common.py:636-643 tv_loss, :646-689 TreeEnergyLoss,
// :692-753 MScaleAddTreeEnergyLoss, :756-818 MScaleRecurveTreeEnergyLoss): what stood between the tree kernels of tree.hip as
// ATen launches through round 4 -- softmax over the class axis, F.interpolate(bilinear, align_corners=False) of the guidance maps,
// F.interpolate(nearest) of the unlabeled-pixel mask and its count, the masked L1 between the soft-max and the filtered map, and
// tv_loss's 3x3 erosion / dilation -- as fused launches: ONE for everything a loss needs before its trees (fi_tree_prep_fwd),
// ONE for the masked L1 of up to three maps with the division by the pixel count (fi_tree_masked_l1_fwd), one each for their
// gradients.  All tensors NCHW fp32 (the loss runs in fp32 like the reference's tree_filter_cuda extension), HBM-bound, a few
// hundred KB to a few MB per launch: one thread per pixel, coalesced along W.
#include "common.h"

namespace {

// torch's area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false): max(0, scale * (dst + 0.5) - 0.5), the
// multiply-subtract as ONE fused operation -- what ATen's CPU build contracts it to (checked bit for bit on ragged scales, 37 x 29 ->
// 101 x 131: every element with the fused form, 93 % with two roundings)
__device__ __forceinline__ float src_index(float scale, int dst) {
  const float s = __builtin_fmaf(scale, (float)dst + 0.5f, -0.5f);
  return s < 0.f ? 0.f : s;
}
// bilinear taps along one axis (upsample_bilinear2d: i0 = (int)src, i1 = i0 + (i0 < n - 1), l1 = src - i0, l0 = 1 - l1)
__device__ __forceinline__ void taps(float scale, int dst, int n, int& i0, int& i1, float& l0, float& l1) {
  const float s = src_index(scale, dst);
  i0 = (int)s;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}
// torch's nearest_neighbor_compute_source_index (the legacy "nearest" mode): min(floor(dst * scale), n - 1)
__device__ __forceinline__ int nearest_index(float scale, int dst, int n) {
  const int i = (int)floorf((float)dst * scale);
  return i < n - 1 ? i : n - 1;
}

struct PrepMaps {
  const float* src[FI_TREE_MAPS];
  float* dst[FI_TREE_MAPS];
  long sn[FI_TREE_MAPS], sc[FI_TREE_MAPS], sy[FI_TREE_MAPS], sx[FI_TREE_MAPS];     // element strides of the source (any layout)
  int C[FI_TREE_MAPS], h[FI_TREE_MAPS], w[FI_TREE_MAPS];
  int n;
};
struct Strides4 {
  long n, c, y, x;
};

// one thread per output pixel (n, y, x): soft-max over the C logits, every guidance map's channels interpolated to H x W, the
// nearest-resized mask and (block-reduced, one fp64 atomic per workgroup) the number of unlabeled pixels
__global__ __launch_bounds__(256) void tree_prep_kernel(const float* __restrict__ preds, Strides4 ps, float* __restrict__ prob, int N,
                                                        int C, int H, int W, PrepMaps m, const unsigned char* __restrict__ roi_src, int rh,
                                                        int rw, float* __restrict__ rois, double* count) {
  const long HW = (long)H * W, total = (long)N * HW;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float mine = 0.f;
  if (i < total) {
    const int n = (int)(i / HW);
    const int r = (int)(i - (long)n * HW), y = r / W, x = r - y * W;
    if (preds) {
      const float* p = preds + (long)n * ps.n + (long)y * ps.y + (long)x * ps.x;     // (the logits usually arrive as an NCHW view of NHWC)
      float mx = p[0];
      for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[(long)c * ps.c]);
      float sum = 0.f;
      for (int c = 0; c < C; ++c) sum += expf(p[(long)c * ps.c] - mx);
      const float inv = 1.f / sum;
      float* q = prob + (long)n * C * HW + r;
      for (int c = 0; c < C; ++c) q[(long)c * HW] = expf(p[(long)c * ps.c] - mx) * inv;
    }
    for (int k = 0; k < m.n; ++k) {
      const int h = m.h[k], w = m.w[k], Ck = m.C[k];
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      taps((float)h / (float)H, y, h, y0, y1, ly0, ly1);
      taps((float)w / (float)W, x, w, x0, x1, lx0, lx1);
      const float* s = m.src[k] + (long)n * m.sn[k];
      const long sy = m.sy[k], sx = m.sx[k];
      float* d = m.dst[k] + (long)n * Ck * HW + r;
      for (int c = 0; c < Ck; ++c) {
        const float* sc = s + (long)c * m.sc[k];
        // upsample_bilinear2d's expression h0lambda * (w0lambda * a + w1lambda * b) + h1lambda * (w0lambda * c + w1lambda * d) in ATen-CPU's
        // ROUNDING ORDER (round 6): its generic kernel evaluates `out = t0 * w0; out += t1 * w1` per axis, which its build
        // contracts to fma(t0, w0, round(t1 * w1)) = fi_lerp2 -- along x for both rows, then along y.  Checked against torch-CPU in this
        // image: every element of 24^2 -> 96^2 and 64^2 -> 256^2, 96 % of a ragged 30 x 41 -> 120 x 131; on SMALL outputs (64^2 and below)
        // torch's kernel mixes contractions by position on ~25 % of the elements and no single formula reproduces it -- the golden g17 maps
        // are of that size, so its tree-tie bars stay an envelope (tests/test_parity2_gpu.py), but a tighter one than with hipcc's own
        // contraction of the line: this order lands on the trees torch's GPU kernel lands on (tools/tree_g17_diag.py)
        const float r0 = fi_lerp2(lx0, sc[y0 * sy + x0 * sx], lx1, sc[y0 * sy + x1 * sx]);
        const float r1 = fi_lerp2(lx0, sc[y1 * sy + x0 * sx], lx1, sc[y1 * sy + x1 * sx]);
        d[(long)c * HW] = fi_lerp2(ly0, r0, ly1, r1);
      }
    }
    if (roi_src) {
      const int sy = nearest_index((float)rh / (float)H, y, rh), sx = nearest_index((float)rw / (float)W, x, rw);
      mine = roi_src[((long)n * rh + sy) * rw + sx] ? 1.f : 0.f;
      rois[i] = mine;
    }
  }
  if (count) {
    __shared__ float red[4];
    const float s = wave_sum(mine);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float t = red[0] + red[1] + red[2] + red[3];
      if (t != 0.f) atomicAdd(count, (double)t);
    }
  }
}

// gradient of the prep launch: dpreds = prob * (dprob - sum_c dprob * prob) per pixel; one thread per SOURCE pixel of each
// guidance map gathers d src = sum over the destination pixels whose two taps along each axis reach it (deterministic: no
// atomics) -- the destination range of a source index is found by inverting the tap rule and re-checked tap by tap
struct PrepGrads {
  const float* gdst[FI_TREE_MAPS];
  float* gsrc[FI_TREE_MAPS];
  int C[FI_TREE_MAPS], h[FI_TREE_MAPS], w[FI_TREE_MAPS];
  long first[FI_TREE_MAPS + 1];      // thread ranges: [first[k], first[k+1]) = source pixels (n, y, x) of map k
  int n;
};

__device__ __forceinline__ void dst_range(float scale, int i, int nsrc, int ndst, int& lo, int& hi) {
  // destination indices whose taps can touch source index i: src(j) in [i - 1, i + 1) apart from the clamped ends
  const float inv = 1.f / scale;
  lo = (int)floorf(((float)i - 1.f + 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.f + 0.5f) * inv - 0.5f) + 1;
  if (i == 0) lo = 0;
  if (i >= nsrc - 1) hi = ndst - 1;
  if (lo < 0) lo = 0;
  if (hi > ndst - 1) hi = ndst - 1;
}

__global__ __launch_bounds__(256) void tree_prep_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ dprob,
                                                            float* __restrict__ dpreds, int N, int C, int H, int W, PrepGrads g) {
  const long HW = (long)H * W, total = dpreds ? (long)N * HW : 0;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) {
    const int n = (int)(i / HW);
    const long r = i - (long)n * HW;
    const float* p = prob + (long)n * C * HW + r;
    const float* d = dprob + (long)n * C * HW + r;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += d[(long)c * HW] * p[(long)c * HW];
    float* o = dpreds + (long)n * C * HW + r;
    for (int c = 0; c < C; ++c) o[(long)c * HW] = p[(long)c * HW] * (d[(long)c * HW] - dot);
    return;
  }
  const long t = i - total;
  for (int k = 0; k < g.n; ++k) {
    if (t < g.first[k] || t >= g.first[k + 1]) continue;
    const int h = g.h[k], w = g.w[k], Ck = g.C[k];
    const long q = t - g.first[k];
    const int n = (int)(q / ((long)h * w));
    const int r = (int)(q - (long)n * h * w), sy = r / w, sx = r - sy * w;
    const float scy = (float)h / (float)H, scx = (float)w / (float)W;
    int ylo, yhi, xlo, xhi;
    dst_range(scy, sy, h, H, ylo, yhi);
    dst_range(scx, sx, w, W, xlo, xhi);
    for (int c = 0; c < Ck; ++c) {
      const float* gd = g.gdst[k] + ((long)n * Ck + c) * HW;
      float acc = 0.f;
      for (int y = ylo; y <= yhi; ++y) {
        int y0, y1;
        float ly0, ly1;
        taps(scy, y, h, y0, y1, ly0, ly1);
        const float wy = (y0 == sy ? ly0 : 0.f) + (y1 == sy ? ly1 : 0.f);
        if (wy == 0.f) continue;
        for (int x = xlo; x <= xhi; ++x) {
          int x0, x1;
          float lx0, lx1;
          taps(scx, x, w, x0, x1, lx0, lx1);
          const float wx = (x0 == sx ? lx0 : 0.f) + (x1 == sx ? lx1 : 0.f);
          if (wx != 0.f) acc += wy * wx * gd[(long)y * W + x];
        }
      }
      g.gsrc[k][((long)n * Ck + c) * h * w + r] = acc;
    }
    return;
  }
}

// sum_k sum rois * |prob - AS_k| (terms kept apart: the reference adds the K sums in order), then by the last workgroup:
// loss = weight * (t_0 + t_1 + ...) / max(count, 1)  -- `if N > 0: tree_loss /= N` without a host round trip (N = 0 => the sum is 0)
struct L1Maps {
  const float* as[FI_TREE_TERMS];
  int n;
};
__global__ __launch_bounds__(256) void tree_masked_l1_kernel(const float* __restrict__ prob, L1Maps m, const float* __restrict__ rois,
                                                             int C, long HW, long total, const double* count, float weight,
                                                             double* acc, unsigned* ticket, float* loss) {
  float s[FI_TREE_TERMS];
#pragma unroll
  for (int k = 0; k < FI_TREE_TERMS; ++k) s[k] = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / ((long)C * HW), r = i % HW;
    const float roi = rois[n * HW + r];
    if (roi == 0.f) continue;
    const float p = prob[i];
#pragma unroll
    for (int k = 0; k < FI_TREE_TERMS; ++k)
      if (k < m.n) s[k] += roi * fabsf(p - m.as[k][i]);
  }
  __shared__ float red[4][FI_TREE_TERMS];
#pragma unroll
  for (int k = 0; k < FI_TREE_TERMS; ++k) {
    const float v = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < m.n; ++k) atomicAdd(&acc[k], (double)red[0][k] + (double)red[1][k] + (double)red[2][k] + (double)red[3][k]);
    __threadfence();
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {               // the last workgroup: every partial sum has landed
      __threadfence();
      double tot = 0.0;
      for (int k = 0; k < m.n; ++k) tot += __hip_atomic_load(&acc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double nn = count[0] > 1.0 ? count[0] : 1.0;
      loss[0] = (float)((double)weight * (tot / nn));
    }
  }
}

// d loss / d prob = g * weight / max(N, 1) * rois * sum_k sign(prob - AS_k);  d loss / d AS_k = -(its own term)
__global__ __launch_bounds__(256) void tree_masked_l1_bwd_kernel(const float* __restrict__ prob, L1Maps m, const float* __restrict__ rois,
                                                                 int C, long HW, long total, const double* count, float weight,
                                                                 const float* __restrict__ gout, float* __restrict__ dprob,
                                                                 float* das0, float* das1, float* das2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const double nn = count[0] > 1.0 ? count[0] : 1.0;
  const float f = gout[0] * (float)((double)weight / nn);
  const long n = i / ((long)C * HW), r = i % HW;
  const float roi = rois[n * HW + r] * f;
  const float p = prob[i];
  float dp = 0.f;
  float* das[FI_TREE_TERMS] = {das0, das1, das2};
#pragma unroll
  for (int k = 0; k < FI_TREE_TERMS; ++k)
    if (k < m.n) {
      const float d = p - m.as[k][i];
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);      // torch.abs' gradient: sign(x), 0 at 0
      dp += roi * sg;
      if (das[k]) das[k][i] = -roi * sg;
    }
  if (dprob) dprob[i] = dp;
}

// ---- tv_loss (flower_common.py:636-643): eroded = -maxpool3x3(-p); contour = relu(maxpool3x3(eroded) - eroded); mean |contour|.
// max_pool2d's tie rule on both passes: the window is scanned row by row and a LATER element replaces the maximum only if it is
// strictly greater, so the FIRST extremum wins; padding never wins (-inf).  idx = window position 0..8 of the winner (for the
// erosion: of the first MINIMUM of p).
__global__ __launch_bounds__(256) void tv_erode_kernel(const float* __restrict__ p, float* __restrict__ er, unsigned char* __restrict__ idx,
                                                       long planes, int H, int W) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= planes * H * W) return;
  const long pl = i / ((long)H * W);
  const int r = (int)(i - pl * H * W), y = r / W, x = r - y * W;
  const float* q = p + pl * H * W;
  float best = INFINITY;
  int bi = 4;
  bool have = false;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const float v = q[yy * W + xx];
      if (!have || v < best || v != v) {                           // (-v > -best) || isnan(-v): torch's update rule on the negation
        best = v;
        bi = (dy + 1) * 3 + dx + 1;
        have = true;
      }
    }
  er[i] = best;
  idx[i] = (unsigned char)bi;
}
__global__ __launch_bounds__(256) void tv_dilate_kernel(const float* __restrict__ er, unsigned char* __restrict__ idx,
                                                        unsigned char* __restrict__ pos, long planes, int H, int W, double* acc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float c = 0.f;
  if (i < planes * H * W) {
    const long pl = i / ((long)H * W);
    const int r = (int)(i - pl * H * W), y = r / W, x = r - y * W;
    const float* q = er + pl * H * W;
    float best = -INFINITY;
    int bi = 4;
    bool have = false;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const float v = q[yy * W + xx];
        if (!have || v > best || v != v) {
          best = v;
          bi = (dy + 1) * 3 + dx + 1;
          have = true;
        }
      }
    const float d = best - q[r];
    c = d > 0.f ? d : 0.f;                                         // relu, then |.| of a non-negative number
    idx[i] = (unsigned char)bi;
    pos[i] = d > 0.f ? 1 : 0;                                      // threshold_backward: gradient where the input is > 0
  }
  __shared__ float red[4];
  const float s = wave_sum(c);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = red[0] + red[1] + red[2] + red[3];
    if (t != 0.f) atomicAdd(acc, (double)t);
  }
}
// d eroded[q] = -dc[q] + sum over the windows w that contain q and whose dilation winner is q of dc[w]   (dc = g / numel where pos)
__global__ __launch_bounds__(256) void tv_bwd_eroded_kernel(const unsigned char* __restrict__ idx_d, const unsigned char* __restrict__ pos,
                                                            const float* __restrict__ gout, float scale, float* __restrict__ der,
                                                            long planes, int H, int W) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= planes * H * W) return;
  const long pl = i / ((long)H * W);
  const int r = (int)(i - pl * H * W), y = r / W, x = r - y * W;
  const float g = gout[0] * scale;
  const unsigned char* id = idx_d + pl * H * W;
  const unsigned char* ps = pos + pl * H * W;
  float acc = ps[r] ? -g : 0.f;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int wy = y - dy, wx = x - dx;                          // window centre for which (y, x) is position (dy, dx)
      if (wy < 0 || wy >= H || wx < 0 || wx >= W) continue;
      if (ps[wy * W + wx] && id[wy * W + wx] == (dy + 1) * 3 + dx + 1) acc += g;
    }
  der[i] = acc;
}
__global__ __launch_bounds__(256) void tv_bwd_input_kernel(const unsigned char* __restrict__ idx_e, const float* __restrict__ der,
                                                           float* __restrict__ dp, long planes, int H, int W) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= planes * H * W) return;
  const long pl = i / ((long)H * W);
  const int r = (int)(i - pl * H * W), y = r / W, x = r - y * W;
  const unsigned char* id = idx_e + pl * H * W;
  const float* de = der + pl * H * W;
  float acc = 0.f;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int wy = y - dy, wx = x - dx;
      if (wy < 0 || wy >= H || wx < 0 || wx >= W) continue;
      if (id[wy * W + wx] == (dy + 1) * 3 + dx + 1) acc += de[wy * W + wx];
    }
  dp[i] = acc;
}

inline unsigned grid_of(long n) { return (unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1); }

}  // namespace

extern "C" int fi_tree_prep_fwd(const float* preds, const long* preds_strides, float* prob, int N, int C, int H, int W,
                                const FiTreeMap* maps, int nmaps, const unsigned char* roi_src, int roi_h, int roi_w, float* rois,
                                double* count, void* stream) {
  if ((preds && (!prob || !preds_strides)) || (nmaps > 0 && !maps) || (roi_src && (!rois || !count))) return FI_ERR_NULL;
  if (N < 1 || H < 1 || W < 1 || nmaps < 0 || nmaps > FI_TREE_MAPS || (preds && C < 1)) return FI_ERR_SHAPE;
  PrepMaps m;
  m.n = nmaps;
  for (int k = 0; k < nmaps; ++k) {
    if (!maps[k].src || !maps[k].dst) return FI_ERR_NULL;
    if (maps[k].C < 1 || maps[k].h < 1 || maps[k].w < 1) return FI_ERR_SHAPE;
    m.src[k] = maps[k].src, m.dst[k] = maps[k].dst, m.C[k] = maps[k].C, m.h[k] = maps[k].h, m.w[k] = maps[k].w;
    m.sn[k] = maps[k].stride[0], m.sc[k] = maps[k].stride[1], m.sy[k] = maps[k].stride[2], m.sx[k] = maps[k].stride[3];
  }
  Strides4 ps = {0, 0, 0, 0};
  if (preds) ps = Strides4{preds_strides[0], preds_strides[1], preds_strides[2], preds_strides[3]};
  hipLaunchKernelGGL(tree_prep_kernel, dim3(grid_of((long)N * H * W)), dim3(256), 0, (hipStream_t)stream, preds, ps, prob, N, C, H, W, m,
                     roi_src, roi_h, roi_w, rois, roi_src ? count : nullptr);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_tree_prep_bwd(const float* prob, const float* dprob, float* dpreds, int N, int C, int H, int W,
                                const FiTreeMap* maps, int nmaps, void* stream) {
  if ((dpreds && (!prob || !dprob)) || (nmaps > 0 && !maps)) return FI_ERR_NULL;
  if (N < 1 || H < 1 || W < 1 || nmaps < 0 || nmaps > FI_TREE_MAPS) return FI_ERR_SHAPE;
  PrepGrads g;
  g.n = nmaps;
  long t = 0;
  for (int k = 0; k < nmaps; ++k) {
    if (!maps[k].src || !maps[k].dst) return FI_ERR_NULL;          // src = gradient w.r.t. the H x W map (in), dst = w.r.t. the source (out)
    g.gdst[k] = maps[k].src, g.gsrc[k] = maps[k].dst, g.C[k] = maps[k].C, g.h[k] = maps[k].h, g.w[k] = maps[k].w;
    g.first[k] = t;
    t += (long)N * maps[k].h * maps[k].w;
  }
  g.first[nmaps] = t;
  const long total = (dpreds ? (long)N * H * W : 0) + t;
  if (total == 0) return 0;
  hipLaunchKernelGGL(tree_prep_bwd_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, prob, dprob, dpreds, N, C, H, W, g);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_tree_masked_l1_fwd(const float* prob, const float* const* as, int nterms, const float* rois, int N, int C, int H,
                                     int W, const double* count, float weight, double* acc_zeroed, float* loss, void* stream) {
  if (!prob || !as || !rois || !count || !acc_zeroed || !loss) return FI_ERR_NULL;
  if (nterms < 1 || nterms > FI_TREE_TERMS || N < 1 || C < 1) return FI_ERR_SHAPE;
  L1Maps m;
  m.n = nterms;
  for (int k = 0; k < nterms; ++k) {
    if (!as[k]) return FI_ERR_NULL;
    m.as[k] = as[k];
  }
  const long HW = (long)H * W, total = (long)N * C * HW;
  unsigned grid = grid_of(total);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(tree_masked_l1_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, prob, m, rois, C, HW, total, count, weight,
                     acc_zeroed, reinterpret_cast<unsigned*>(acc_zeroed + FI_TREE_TERMS), loss);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_tree_masked_l1_bwd(const float* prob, const float* const* as, int nterms, const float* rois, int N, int C, int H,
                                     int W, const double* count, float weight, const float* grad_out, float* dprob,
                                     float* const* das, void* stream) {
  if (!prob || !as || !rois || !count || !grad_out) return FI_ERR_NULL;
  if (nterms < 1 || nterms > FI_TREE_TERMS || N < 1 || C < 1) return FI_ERR_SHAPE;
  L1Maps m;
  m.n = nterms;
  float* d[FI_TREE_TERMS] = {nullptr, nullptr, nullptr};
  for (int k = 0; k < nterms; ++k) {
    if (!as[k]) return FI_ERR_NULL;
    m.as[k] = as[k];
    d[k] = das ? das[k] : nullptr;
  }
  const long HW = (long)H * W, total = (long)N * C * HW;
  hipLaunchKernelGGL(tree_masked_l1_bwd_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, prob, m, rois, C, HW, total,
                     count, weight, grad_out, dprob, d[0], d[1], d[2]);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_tv_loss_fwd(const float* p, long planes, int H, int W, float* eroded, unsigned char* idx_e, unsigned char* idx_d,
                              unsigned char* positive, double* acc_zeroed, void* stream) {
  if (!p || !eroded || !idx_e || !idx_d || !positive || !acc_zeroed) return FI_ERR_NULL;
  if (planes < 1 || H < 1 || W < 1) return FI_ERR_SHAPE;
  const long n = planes * H * W;
  hipLaunchKernelGGL(tv_erode_kernel, dim3(grid_of(n)), dim3(256), 0, (hipStream_t)stream, p, eroded, idx_e, planes, H, W);
  hipLaunchKernelGGL(tv_dilate_kernel, dim3(grid_of(n)), dim3(256), 0, (hipStream_t)stream, eroded, idx_d, positive, planes, H, W,
                     acc_zeroed);
  FI_CHECK_LAUNCH();
  return 0;
}

extern "C" int fi_tv_loss_bwd(const unsigned char* idx_e, const unsigned char* idx_d, const unsigned char* positive,
                              const float* grad_out, long planes, int H, int W, float* scratch, float* dp, void* stream) {
  if (!idx_e || !idx_d || !positive || !grad_out || !scratch || !dp) return FI_ERR_NULL;
  if (planes < 1 || H < 1 || W < 1) return FI_ERR_SHAPE;
  const long n = planes * H * W;
  hipLaunchKernelGGL(tv_bwd_eroded_kernel, dim3(grid_of(n)), dim3(256), 0, (hipStream_t)stream, idx_d, positive, grad_out,
                     (float)(1.0 / (double)n), scratch, planes, H, W);
  hipLaunchKernelGGL(tv_bwd_input_kernel, dim3(grid_of(n)), dim3(256), 0, (hipStream_t)stream, idx_e, scratch, dp, planes, H, W);
  FI_CHECK_LAUNCH();
  return 0;
}
