/* fedicra_hip.h -- C ABI of libfedicra_hip.so (MI355X / gfx950 only).
 *
 * The reference (llmir/FedICRA) has NO native boundary on this path: every op below is an
 * ATen call made from Python.  This header is therefore the boundary
 * the *build* defines; each entry point cites the reference line whose arithmetic it replaces.
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  Every pointer is DEVICE memory owned by the
 *     caller (incl. workspaces); nothing is allocated, freed or synchronised inside.
 *   - all work is enqueued on the hipStream_t passed as `stream` (void*, 0 = default stream);
 *     calls are stream-ordered, re-entrant, and hipGraph-capturable.
 *   - activations are dense NHWC ("channels_last"): element (n,h,w,c) at ((n*H+h)*W+w)*C+c.
 *     conv weights are [Cout][kh][kw][Cin] (= torch channels_last of [Cout,Cin,kh,kw]).
 *   - dtype: FI_F32 = exact-fp32 path (v_mfma_f32_16x16x4_f32), FI_BF16 = bf16 storage with
 *     fp32 accumulate (v_mfma_f32_16x16x32_bf16).  Statistics, losses, optimizer: fp32/fp64.
 *   - return value: 0 = ok; FI_ERR_* (negative) = bad argument; positive = hipError_t.
 *     Nothing throws across the ABI.
 */
#ifndef FEDICRA_HIP_H
#define FEDICRA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FI_F32 0
#define FI_BF16 1
#define FI_F16 2 /* IEEE half storage, fp32 accumulation: the reference's autocast dtype (flower_pCE_2D.py:104) */

#define FI_OK 0
#define FI_ERR_DTYPE (-1)
#define FI_ERR_SHAPE (-2)
#define FI_ERR_UNSUPPORTED (-3)
#define FI_ERR_NULL (-4)

/* dropout specification shared by fi_bn_act_fwd / _bwd_* */
#define FI_DROP_NONE 0
#define FI_DROP_MASK_ELEM 1 /* explicit uint8 keep-mask, one per element [N,H,W,C] (parity mode)      */
#define FI_DROP_RNG_ELEM 2  /* counter RNG keyed (seed, element index) -- nn.Dropout                   */
#define FI_DROP_MASK_CHAN 3 /* explicit uint8 keep-mask per (n,c) [N,C] -- nn.Dropout2d (parity mode) */
#define FI_DROP_RNG_CHAN 4  /* counter RNG keyed (seed, n*C+c) -- nn.Dropout2d                        */

/* Bumped whenever the signature of an exported function or the layout of a struct below changes (additions of new entry
 * points included): the host mirror (fedicra_amd/_lib.py) refuses a library whose version differs from the header it was
 * written against, so a stale or foreign libfedicra_hip.so fails at load, not by passing a pointer in the wrong slot.
 *   1: rounds 1-2 (fi_ala_update gained `skip` without a bump -- the reason for this note)
 *   2: round 3 (FiConv.w16 / w16_rows + fi_pack_weights modes 2 / 3 and the 8-column pack table, fi_conv_weight_chunk16,
 *      fi_pcs_gate_*, fi_lc_loss_*, fi_conv3d_wgrad_fused*, this check)
 *   3: round 4 (new entry points: fi_conv2d_stats_xcorr*, fi_bn_act_pool_groups, fi_conv1x1_up2x_fwd, fi_wgrad_tuning / fi_narrow_tuning /
 *      fi_upfuse_tuning, fi_pack_weights3d_multi; fi_wgrad_tuning's argument became a bit mask)
 *   5: round 6 (fi_conv3d_tuning; fi_conv_tuning(7, 8): the LDS-DMA forward form) */
#define FI_ABI_VERSION 6
int fi_abi_version(void);

/* ---------------------------------------------------------------- convolution ------------
 * Implicit-GEMM Conv2d, stride 1, kernel 1x1 or 3x3 with "same" zero padding.
 * Replaces nn.Conv2d in ConvBlock / UpBlock.conv1x1 / out_conv / dsn_head
 * (/root/reference/code/networks/unet.py:19-27, 57, 225-226, 261-267).
 * The input may be the channel-concatenation of two tensors (x0: c0 channels, x1: c1 channels;
 * x1 == NULL, c1 == 0 for a plain conv): this folds torch.cat([skip, up], dim=1) (unet.py:69)
 * into the gather.  The output may likewise be split over two tensors (dgrad w.r.t. a concat).
 */
typedef struct FiConv {
  int dtype;          /* FI_F32 / FI_BF16: storage type of x, w, y                                  */
  int N, H, W;        /* batch, height, width (output == input size)                                 */
  int ksize;          /* 1 or 3                                                                      */
  int c0, c1;         /* input channels from x0 / x1                                                 */
  int co0, co1;       /* output channels to y0 / y1 (co1 == 0 for a single destination)              */
  int accumulate0;    /* y0 += result instead of y0 = result                                         */
  int accumulate1;    /* same for y1                                                                 */
  int y_f32;          /* store the output as fp32 even when dtype == FI_BF16 (logits)                */
  /* Optional second form of the SAME filter, chunk-major (fi_pack_weights mode 2 / 3; NULL = none): where
   * fi_conv_weight_chunk16() says so for the filter's shape, fi_conv2d_fwd / _fwd_fused may take the launch through the
   * 64 x 64-wave-tile kernel (conv_fwd_ws2_kernel), which stages 16-channel slabs of it as contiguous blocks.  Ignored by
   * every other entry point.  w16_rows = output rows per chunk of that operand (0: co0 + co1): a launch may cover a
   * sub-range of the packed filter's output channels (`w16` then points at its first row inside chunk 0). */
  const void* w16;
  int w16_rows;
} FiConv;

/* y = conv(cat(x0,x1), w) + bias.  w: [co0+co1][k*k][c0+c1] in `dtype` (see fi_pack_weights).
 * bias: fp32 [co0+co1] or NULL.  stats: fp64 [FI_STATS_SLOTS][co0+co1][2], or NULL; when given,
 * per-channel (sum, sum of squares) of the stored output are ATOMICALLY ADDED to one of the slots
 * (caller zeroes the whole buffer; fi_bn_finalize sums the slots) -- the batch statistics
 * BatchNorm2d (unet.py:21) needs, produced in the conv epilogue.  y0 == NULL (with stats given, co1 == 0) is a
 * statistics-only launch: nothing is stored -- a train-mode forward whose output nobody reads still has to move the
 * BatchNorm running statistics (the no-grad forwards of FedICRA's LC loss through the auxiliary heads,
 * flower_pCE_2D.py:128-139). */
#define FI_STATS_SLOTS 8
int fi_conv2d_fwd(const FiConv* d, const void* x0, const void* x1, const void* w, const float* bias,
                  void* y0, void* y1, double* stats, void* stream);

/* Fused ("probe") forward: the same convolution, but every source may hold the RAW output y of its producing
 * convolution, and the loader evaluates  z = dropout(act(scale[c]*y + shift[c]))  -- the BatchNorm apply, LeakyReLU / ReLU
 * and element-wise Dropout of ConvBlock (unet.py:21-24) -- while it stages the tile, optionally followed by the 2x2
 * max-pool of a DownBlock (unet.py:40; `pool`: source 0 is then [N][2H][2W][c0], c1 must be 0).  z is rounded to the
 * storage dtype exactly as fi_bn_act_fwd rounds it, so a chain of fused launches reproduces the unfused forward bit for
 * bit while the activations z are never written to or read from HBM.  Used for the no-grad forwards of FedICRA's LC
 * loss (flower_pCE_2D.py:128-139): K-1 forwards per iteration whose only outputs are the heat-map and the BatchNorm
 * running statistics.  Those K-1 forwards run as ONE batch of `group_images`-image groups: coefficient rows, dropout
 * seeds (seed + g * seed_group_stride, element index inside the group) and the statistics accumulators
 * (stats + g * stats_group_stride) belong to group g = n / group_images, i.e. each group sees exactly the batch
 * statistics and masks its own forward would have seen.  scale == NULL: source used as it is (t == NULL likewise).
 * Channel counts must be whole 16-byte vectors; single destination, storage dtype, no accumulation. */
typedef struct FiInXform {
  const float* scale;          /* fp32 [groups][C] (fi_bn_finalize_groups), or NULL                                  */
  const float* shift;          /* fp32 [groups][C]                                                                   */
  float slope;                 /* LeakyReLU negative slope; 0 = ReLU                                                 */
  int pool;                    /* 1: 2x2 max-pool after the transform (source 0 only)                                */
  int drop_mode;               /* FI_DROP_NONE or FI_DROP_RNG_ELEM (source 0 only, not with pool)                    */
  float drop_p;
  uint64_t seed;
  uint64_t seed_group_stride;
  const int32_t* seed_offset;  /* as FiBnAct.seed_offset                                                             */
} FiInXform;
#define FI_FUSED_SHARED_SOURCE0 1 /* source 0 holds ONE group (group_images images) that every group reads: the first
                                    ConvBlock output of the batch, identical for all K-1 forwards up to its dropout   */
int fi_conv2d_fwd_fused(const FiConv* d, const FiInXform* t0, const FiInXform* t1, int group_images, int flags,
                        const void* x0, const void* x1, const void* w, const float* bias, void* y, double* stats,
                        long stats_group_stride, void* stream);

/* The statistics of a 3x3 convolution WITHOUT the convolution (csrc/xcorr.hip): what a STATISTICS-ONLY fi_conv2d_fwd_fused
 * launch (y == NULL) adds to `stats` -- per group and output channel the sum and the sum of squares of y = conv(z) + bias over
 * the group's pixels -- from the 13 autocorrelation matrices of the 64-channel input (A_d = sum_q z(q) z(q+d)^T, |d| <= 2,
 * on the matrix pipe with K = pixels), the quadratic form with the filter, and the convolution evaluated on the one-pixel
 * frame around each image (taken off again).  53 K instead of 295 K multiply-adds per pixel for the auxiliary head
 * Conv2d(64, 512, 3) of the LC forwards (/root/reference/code/networks/unet.py:261-267, flower_pCE_2D.py:128-139), whose
 * output nobody reads.  Covered: 16-bit storage, ksize 3, c0 == 64, c1 == 0, W in {64, 128}, 4 <= H <= 144, co0 % 8 == 0,
 * at most 8 groups; t0 as for fi_conv2d_fwd_fused without pool / dropout (NULL: x0 is the activation itself); w = the
 * forward operand [co0][9][64] (fi_pack_weights mode 0).  FI_ERR_UNSUPPORTED otherwise: the caller makes the direct launch.
 * stats: fp64 [groups][FI_STATS_SLOTS][co0][2], zeroed by the caller; slot 0 of every group is WRITTEN.  The values are those
 * of the exact fp32 outputs; the direct launch takes them of the outputs as rounded to the storage type (~1e-6 relative
 * apart).  workspace: caller-owned, fi_conv2d_stats_xcorr_workspace bytes (< 0: FI_ERR_*).
 * fi_conv2d_stats_xcorr_layout (tests): byte offsets of { partials, A fp64 [groups][13*4096 + 64], B, edge strips, ring
 * weights, ring statistics, Q, workgroups per group that hold rows } inside the workspace. */
long fi_conv2d_stats_xcorr_workspace(const FiConv* d, int group_images);
int fi_conv2d_stats_xcorr_layout(const FiConv* d, int group_images, long* offsets);
int fi_conv2d_stats_xcorr(const FiConv* d, const FiInXform* t0, int group_images, const void* x0, const void* w,
                          const float* bias, double* stats, long stats_group_stride, void* workspace, long workspace_bytes,
                          void* stream);

/* Measurement hook (tools/kbench.py, tests): which forward kernel fi_conv2d_fwd[_fused] launches.  v2 = -1: the library's
 * per-layer choice (default), 0: the one-tile kernel everywhere, 1: the persistent kernel wherever it applies (16-bit
 * storage, 3x3, whole-vector channel counts, plain epilogue), 3: the thin-layer kernel (filter in registers) wherever it
 * applies (additionally Cin, Cout <= 32, Cout % 8 == 0, one destination), 4 / 5 / 6: the wave-specialised kernel (producer /
 * consumer waves, double-buffered LDS stages; additionally Cin, Cout >= 32, destinations of whole 8-channel groups) with
 * 4 + 4, 4 + 8 or 8 + 2x4 consumer + producer waves, 7: the 64 x 64-wave-tile kernel wherever it applies (needs FiConv.w16;
 * nf = 1: 16-row x 128-channel tiles, 2: 32 x 64, 4: the whole filter resident in LDS -- Cout = 32 or 64 and
 * Cin * Cout * 18 bytes <= ~84 KB, 8: the LDS-DMA form -- two 16 x 16-pixel sub-tiles x 128 channels per workgroup, every
 * operand byte by buffer_load ... lds, the input transform in place in LDS; Cout % 128 == 0, sources of whole 32-channel
 * groups); nf / ck / wgs_per_cu = 0 keep the defaults, else force the slab width
 * (1, 2, 4 fragments of 16 channels; 2 or 4 for the wave-specialised kernel), channel chunk (16, 32) and workgroups per CU.
 * Process-wide, not thread-safe.  The kernels compute the same products in fp32; only the ORDER in which the channel chunks
 * are accumulated follows the chunk width, so two configurations agree to fp32 round-off (a bf16 output may differ in its
 * last bit for a few elements in 10^5), and bit for bit when their chunk widths agree. */
int fi_conv_tuning(int v2, int nf, int ck, int wgs_per_cu);

/* dw[co][k*k][ci] += sum_pixels dy * x  (fp32);  dbias[co] += sum_pixels dy (fp32, may be NULL).
 * d->co0 = Cout, co1 ignored; dy is [N,H,W,Cout].  With a caller-owned `workspace` of at least
 * fi_conv2d_wgrad_workspace(d) bytes the reduction is two-stage and DETERMINISTIC (per-workgroup partial
 * sums -> plain stores -> fixed-order sum); with workspace == NULL partial sums are added with fp32
 * atomics (order-dependent rounding).  Either way the result is ADDED to dw / dbias (caller zeroes). */
long fi_conv2d_wgrad_workspace(const FiConv* d);
int fi_conv2d_wgrad(const FiConv* d, const void* x0, const void* x1, const void* dy, float* dw,
                    float* dbias, void* workspace, long workspace_bytes, void* stream);

/* Conv3d (3x3x3 pad 1 or 1x1x1, stride 1; /root/reference/code/networks/utils.py:99-123, networks/vnet.py:15) over dense
 * NDHWC volumes [N][D][H][W][C]: a volume is D consecutive NHWC slices, so depth tap kd is ONE 2D implicit-GEMM launch over
 * the slices it reaches, accumulating into the output (the centre tap last: its epilogue adds the bias and yields the
 * statistics).  `d` describes a slice batch (N = samples, H, W, ksize, channel split); w_taps[kd] are the packed 2D
 * operands of the depth taps (fi_pack_weights of weight[:, :, kd]).
 *   fwd  : y (caller ZEROES it) += conv; stats: per-SAMPLE accumulators, sample n at stats + n*stats_stride (InstanceNorm),
 *          or stats_stride = 0 for batch statistics, or NULL.
 *   dgrad: d->c0 = channels of dy, co0 / co1 = channels of the (concatenated) input; d0 / d1 (caller zeroes) += .
 *   wgrad: dw_taps fp32 [kd][cout][k][k][cin] and dbias ADDED to; workspace >= fi_conv3d_wgrad_workspace(d, D) bytes. */
int fi_conv3d_fwd(const FiConv* d, int D, const void* x0, const void* x1, const void* const* w_taps, const float* bias,
                  void* y, double* stats, long stats_stride, void* stream);
int fi_conv3d_dgrad(const FiConv* d, int D, const void* dy, const void* const* wt_taps, void* d0, void* d1, void* stream);
/* One-launch forms (16-bit storage, 3x3x3, channel counts multiples of 8 and >= 16): the slices of all volumes are the images of
 * ONE implicit GEMM, the depth taps three channel groups of its contraction.  w_all = [Cout][k*k][3][c0 + c1] (the per-tap
 * operands of fi_conv3d_fwd interleaved); wt_all = [c_in][k*k][3][Cout] with the depth taps reversed.  Outputs are WRITTEN (no
 * zeroing by the caller), statistics per volume as above.  d->accumulate* must be 0.  FI_ERR_UNSUPPORTED: shape not covered --
 * fall back to the per-tap forms.  Replaces the same reference calls as fi_conv3d_fwd / fi_conv3d_dgrad
 * (networks/utils.py:224-245 UnetConv3: nn.Conv3d(3,3,3) + InstanceNorm3d + ReLU). */
int fi_conv3d_fwd_fused(const FiConv* d, int D, const void* x0, const void* x1, const void* w_all, const float* bias, void* y,
                        double* stats, long stats_stride, void* stream);
int fi_conv3d_dgrad_fused(const FiConv* d, int D, const void* dy, const void* wt_all, void* d0, void* d1, void* stream);
/* Measurement / test hook: 1 (default; FI_CONV3D_STREAM) = fi_conv3d_fwd_fused / fi_conv3d_dgrad_fused run the thin full-resolution
 * layers of unet_3D (/root/reference/code/networks/unet_3D.py:40-41,60-61: 16 -> 16, (16 + 32) -> 16; their input gradients 16 -> 16,
 * 16 -> (16 + 32); 16-bit storage) on the depth-streaming kernel (csrc/conv3d_stream.hip: every input slice staged once per 16 x 16
 * tile through a three-slice LDS ring); 0 = the general one-launch form for them too; -1 = the environment default. */
int fi_conv3d_tuning(int stream_on);
long fi_conv3d_wgrad_workspace(const FiConv* d, int D);
int fi_conv3d_wgrad(const FiConv* d, int D, const void* x0, const void* x1, const void* dy, float* dw_taps, float* dbias,
                    void* workspace, long workspace_bytes, void* stream);
/* The same filter gradient in ONE launch (whole-vector channel counts; FI_ERR_UNSUPPORTED otherwise): dw_all fp32
 * [cout][9][3][c0 + c1] -- depth taps as channel groups, the layout of fi_conv3d_fwd_fused's operand -- and dbias are ADDED to;
 * workspace >= fi_conv3d_wgrad_fused_workspace(d, D) bytes (deterministic two-stage reduction). */
long fi_conv3d_wgrad_fused_workspace(const FiConv* d, int D);
int fi_conv3d_wgrad_fused(const FiConv* d, int D, const void* x0, const void* x1, const void* dy, float* dw_all, float* dbias,
                          void* workspace, long workspace_bytes, void* stream);
/* Stage 1 of fi_conv3d_wgrad_fused only (the 3D sibling of fi_conv2d_wgrad_partial): partial slices [slices][cout*9*3*cin (+ cout
 * bias sums when want_bias)] stay in `workspace` (fi_conv3d_wgrad_fused_workspace bytes); fold them with fi_wgrad_reduce_multi, table
 * word 9 = cin, into the parameter gradient [cout][cin][3][3][3] -- one launch for every layer of a backward pass instead of a reduce
 * launch and a permuted add per layer (unet_3D: 17 + 17 launches, 0.28 ms of a 6.3 ms iteration). */
int fi_conv3d_wgrad_fused_partial(const FiConv* d, int D, const void* x0, const void* x1, const void* dy, int want_bias,
                                  void* workspace, long workspace_bytes, int* slices, long* stride, void* stream);

/* Deferred form: only stage 1 (partial sums into `workspace`); *slices / *stride (floats) describe the layout
 * [slices][cout*k*k*cin (+ cout bias sums when want_bias)].  Many layers' stage 2 are then done by ONE launch of
 * fi_wgrad_reduce_multi over a device table (int64[n][FI_WGRAD_ROW]):
 * { partial ptr, stride (multiple of 4), slices, dw ptr, cout*k*k*cin, dbias ptr or 0, cout, first_block, log2(lanes), cin3,
 *   first_block of fi_wgrad_permute3d_multi }
 * (cin3 = 0: dw has the slices' own layout; cin3 != 0: the slices come from fi_conv3d_wgrad_fused_partial -- [cout][9][3][|cin3|] --
 * and dw is the parameter's [cout][|cin3|][3][3][3].  cin3 > 0: fi_wgrad_reduce_multi permutes as it adds -- 4-byte read-modify-writes
 * 108 B apart, 202 us per unet_3D iteration; cin3 < 0: it adds the bias gradient but leaves the weight sums in slice 0 of the partials,
 * and ONE launch of fi_wgrad_permute3d_multi over the same table then adds them into dw through an LDS transpose -- a workgroup per
 * (output channel, <= 64 input channels), so tensor t owns its workgroups [word 10, word 10 + cout * ceil(|cin3| / 64)) and nblocks is
 * their total over the rows with cin3 < 0 -- 77 + 12 us)
 * where a workgroup folds 4*lanes consecutive elements (lanes in {16, 64, 256}: few lanes when there are many slices),
 * tensor t owns workgroups [first_block_t, first_block_t + ceil(stride_t / (4*lanes_t))) and nblocks is their total.
 * dw/dbias += fixed-order slice sums. */
#define FI_WGRAD_ROW 11
int fi_conv2d_wgrad_partial(const FiConv* d, const void* x0, const void* x1, const void* dy, int want_bias,
                            void* workspace, long workspace_bytes, int* slices, long* stride, void* stream);
int fi_wgrad_reduce_multi(const long long* table, int ntensors, int nblocks, void* stream);
int fi_wgrad_permute3d_multi(const long long* table, int ntensors, int nblocks, void* stream);
/* Measurement / test hook (a bit mask): rows = 0 keeps every filter gradient on the tile kernels, bit 0 lets the thin 3x3 layers on
 * large maps (16 / 32 channels a side, 16-bit storage, W % 32 == 0, a workspace given) take the row-streaming kernel
 * (csrc/wgrad_rows.h), bit 1 the channel-rich 3x3 layers (32 ... 256 channels a side, one side a multiple of 64, 64 <= W, W <= 128
 * or W % 128 == 0) its 64 x 64-channel-tile form; -1 = the FI_WGRAD_ROWS / FI_WGRAD_ROWS64 environment defaults (1 / 1). */
int fi_wgrad_tuning(int rows);
/* Measurement / test hook (a bit mask; 7 = all forms): on = 0 keeps the layers with a <= 4-channel side (the U-Nets' first convolution in_chns -> 16,
 * /root/reference/code/networks/unet.py:82,163, and their logits convolution 16 -> n_class, :228) on the general tile kernels;
 * bit 0 = fi_conv2d_fwd with C <= 4 inputs -> 8 / 16 outputs (also the logits convolution's input gradient), bit 1 = fi_conv2d_fwd with
 * 16 / 32 inputs -> <= 4 fp32 outputs, bit 2 = fi_conv2d_wgrad* with either side <= 4 channels against 16 (row-streaming,
 * workspace given); 3x3, 16-bit storage, one source.  -1 = the FI_NARROW environment default (7). */
int fi_narrow_tuning(int on);

/* weight repack from the fp32 master [Cout][k*k][Cin]:
 *   mode 0: dst[co][t][ci]          = (dtype) src[co][t][ci]          (forward operand)
 *   mode 1: dst[ci][k*k-1-t][co]    = (dtype) src[co][t][ci]          (dgrad operand: conv of dy
 *           with the flipped, transposed filter -- run through fi_conv2d_fwd with Cin<->Cout)
 *   mode 2: dst[ci/16][co][t][ci%16]        = mode 0's values, CHUNK-MAJOR (cin % 16 == 0): the slab a 16-channel
 *           stage of conv_fwd_ws2_kernel needs -- all output rows x 9 taps x 16 channels -- is one contiguous block
 *   mode 3: dst[co/16][ci][k*k-1-t][co%16]  = mode 1's values, chunk-major over ITS contraction channels (cout % 16 == 0) */
int fi_pack_weights(const float* src, void* dst, int cout, int kk, int cin, int mode, int dtype, void* stream);

/* 1 when a filter of this shape gets the chunk-major second operand (FiConv.w16): 16-bit storage, 3x3, contraction channels
 * (cin; for the dgrad operand pass the conv's cout as cin and vice versa) and output channels both multiples of 32 (the 32-output resident-filter form; 64 / 128-output slabs otherwise).
 * FI_WS2=0 in the environment switches the form off (A/B runs of one build). */
int fi_conv_weight_chunk16(int dtype, int ksize, int cin, int cout);

/* The same repack for MANY weights in one launch.  table (device, int64[ntensors][8]) rows:
 * { src fp32 master ptr, dst forward operand ptr or 0, dst dgrad operand ptr or 0, cout, k*k, cin,
 *   dst chunk-major forward operand (mode 2) or 0, dst chunk-major dgrad operand (mode 3) or 0 }. */
#define FI_PACK_ROW 8
int fi_pack_weights_multi(const long long* table, int ntensors, int dtype, void* stream);

/* ---------------------------------------------------------------- BatchNorm + activation --
 * nn.BatchNorm2d (eps 1e-5, momentum 0.1) -> LeakyReLU(0.01)/ReLU -> Dropout, unet.py:21-24, 263-265. */

/* training: from stats (fi_conv2d_fwd) and count = N*H*W compute batch mean / biased var,
 * scale = gamma*invstd, shift = beta - mean*scale; update running_mean/var (unbiased var) and
 * ++num_batches_tracked (all in place, fp32 / int64).  eval (training == 0): scale/shift from the
 * running statistics; mean/invstd outputs hold the running values; stats may be NULL. */
int fi_bn_finalize(const double* stats, double count, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int training,
                   float* scale, float* shift, float* mean, float* invstd, int C, void* stream);

/* The same for G statistics groups of one launch of fi_conv2d_fwd_fused (group g at stats + g*stats_group_stride,
 * `count` elements each): coef = fp32 [2][G][C] (scale rows, then shift rows), and the running statistics are moved G
 * times IN GROUP ORDER -- what G consecutive train-mode forwards would do -- with num_batches_tracked += G.
 * Either half alone (round 5): running_mean == running_var == NULL -> coefficients only (no state moves, the counter is left);
 * coef == NULL -> running statistics and counter only.  The two calls together leave exactly what the whole call leaves: a
 * caller whose running-statistics update has to wait for another stream gets the coefficients at once and makes the update
 * later (the batched LC forwards beside the client's own forward: flower_pCE_2D.py:106,128-139 fixes their order). */
int fi_bn_finalize_groups(const double* stats, long stats_group_stride, int groups, double count, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                          float momentum, float eps, float* coef, int C, void* stream);

/* The running-statistics half (coef == NULL above) for `n` BatchNorm layers in ONE launch -- the K-1 batched LC forwards of an
 * iteration leave the updates of all their layers to the training stream (flower_pCE_2D.py:106,128-139 fixes the order: the own
 * forward's update of a layer first), where they would otherwise be n dependent launches.  Same arithmetic, same order per layer. */
#define FI_BN_RUN_MAX 32
typedef struct FiBnRunItem {
  const double* stats;         /* [groups][SLOTS][C][2] (stats_group_stride doubles apart; 0 = one shared accumulator set)  */
  long stats_group_stride;
  double count;                /* elements behind one group's statistics (N_group * H * W)                                 */
  float* running_mean;         /* fp32 [C], in place                                                                       */
  float* running_var;
  int64_t* num_batches_tracked;/* += groups, or NULL                                                                       */
  float momentum;
  int groups, C;
} FiBnRunItem;
int fi_bn_running_groups_multi(const FiBnRunItem* items, int n, void* stream);

typedef struct FiBnAct {
  int dtype;
  long pixels;        /* N*H*W */
  int C;
  int hw;             /* H*W (to find n for the per-(n,c) dropout forms) */
  float slope;        /* LeakyReLU negative slope; 0 = ReLU; 1 = identity */
  int drop_mode;      /* FI_DROP_* */
  float drop_p;
  uint64_t seed;      /* RNG forms */
  const uint8_t* mask;/* MASK forms */
  const int32_t* seed_offset; /* RNG forms, may be NULL: device int32 added to the seed stream (e.g. the
                                 training-iteration counter) so a replayed hipGraph draws fresh masks */
} FiBnAct;

/* fi_bn_finalize + fi_bn_act_fwd in ONE launch: every workgroup folds the statistic slots itself; coef
 * (fp32 [4][C] = scale, shift, mean, invstd) is published for the backward kernels. */
int fi_bn_fused_fwd(const FiBnAct* d, const void* y, void* z, const double* stats, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float momentum, float eps, int training, float* coef, void* stream);

/* z = dropout(act(y*scale[c] + shift[c])) */
int fi_bn_act_fwd(const FiBnAct* d, const void* y, const float* scale, const float* shift, void* z, void* stream);
/* sums[slot][c][0] += sum g, sums[slot][c][1] += sum g*xhat   (g = dz through dropout and activation,
 * xhat = (y-mean)*invstd); fp64 [FI_STATS_SLOTS][C][2], atomically added to one slot per workgroup, caller
 * zeroes; fi_bn_act_bwd_apply folds the slots.  C <= 512 and C/vec must divide 256. */
int fi_bn_act_bwd_reduce(const FiBnAct* d, const void* dz, const void* y, const float* scale, const float* shift,
                         const float* mean, const float* invstd, double* sums, void* stream);
/* dy = scale*(g - sum_g/M - xhat*sum_gx/M) (training) or scale*g (eval);
 * dgamma[c] (+)= sum_gx, dbeta[c] (+)= sum_g  (written by block 0; accumulate_param selects += ). */
int fi_bn_act_bwd_apply(const FiBnAct* d, const void* dz, const void* y, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const double* sums, int training, void* dy,
                        float* dgamma, float* dbeta, int accumulate_param, void* stream);

/* ---------------------------------------------------------------- pooling / resampling ---- */
/* nn.MaxPool2d(2) (unet.py:40): y[N,H/2,W/2,C]; H, W even. */
int fi_maxpool2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream);
/* z [N][Ho][Wo][C] = maxpool2x2(act(scale_g * y + shift_g)) of a RAW convolution output y [N][2Ho][2Wo][C], one coefficient row
 * pair [group][C] per statistics group g = n / group_images (0: one group): exactly what the pooling loader of
 * fi_conv2d_fwd_fused evaluates (each value rounded to the storage type like fi_bn_act_fwd, then the scan-order strict
 * maximum of fi_maxpool2_fwd), written out once -- for the deep DownBlocks of the batched LC forwards, whose input tile
 * would otherwise be transformed and pooled once per output slab (ConvBlock + MaxPool2d, /root/reference/code/networks/
 * unet.py:14-46).  pool == 0: the same without the pooling (y [N][Ho][Wo][C]): act(BN(y)) of every group in one launch,
 * fi_bn_act_fwd's arithmetic -- the frozen encoder's feature maps for all batches of an ALA epoch (code/flower_common.py:
 * 566-602).  0 <= slope <= 1. */
int fi_bn_act_pool_groups(int dtype, const void* y, const float* scale, const float* shift, float slope, void* z, int N, int Ho,
                          int Wo, int C, int group_images, int pool, void* stream);
/* dx = route dy to the first maximum of each window (scan order, strict >), zeros elsewhere. */
int fi_maxpool2_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int H, int W, int C, int accumulate,
                    void* stream);
/* dx = add + (dy routed as above): the encoder features x0..x3 are pooled AND returned as skip connections
 * (unet.py:91-99), so their gradient is the sum of two; this writes the sum in the pooling pass (add != dx allowed). */
int fi_maxpool2_bwd_add(int dtype, const void* x, const void* dy, const void* add, void* dx, int N, int H, int W, int C,
                        void* stream);
/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (unet.py:58-59): [N,h,w,C]->[N,2h,2w,C] */
int fi_upsample2x_fwd(int dtype, const void* x, void* y, int N, int h, int w, int C, void* stream);
int fi_upsample2x_bwd(int dtype, const void* dy, void* dx, int N, int h, int w, int C, int accumulate, void* stream);
/* UpBlock's first half as ONE launch (csrc/upfuse.hip; /root/reference/code/networks/unet.py:57-59,65-67: self.conv1x1 then
 * self.up): y [N,2h,2w,cout] = fi_upsample2x_fwd(conv1x1(z) + bias), z = x [N,h,w,cin] as it is (t0 == NULL or t0->scale == NULL) or
 * act(BN(x)) per statistics group as in fi_conv2d_fwd_fused (no pool / dropout).  The low-resolution convolution output is
 * rounded to the storage type as the separate launch stores it, but lives in LDS only; the interpolation is
 * fi_upsample2x_fwd's, operand for operand.  wmat = the forward operand [cout][cin] (fi_pack_weights mode 0), bias may be NULL.
 * 16-bit storage and (cin, cout) in {(32,16), (64,32), (128,64), (256,128), (32,32), (64,64)}; FI_ERR_UNSUPPORTED otherwise
 * (the caller makes the two launches).  Backward: fi_upsample2x_bwd, then the 1x1 convolution's own. */
int fi_conv1x1_up2x_fwd(int dtype, int N, int h, int w, int cin, int cout, const FiInXform* t0, int group_images, const void* x,
                        const void* wmat, const float* bias, void* y, void* stream);
/* Measurement / test hook: input rows convolved per workgroup of fi_conv1x1_up2x_fwd (2 * rows output rows; 0 = the per-shape default: 3, or 6 from 128 input channels on). */
int fi_upfuse_tuning(int rows);

/* 3D surface (unet_3D, /root/reference/code/networks/unet_3D.py:20-94): volumes are dense NDHWC, i.e. D consecutive NHWC
 * slices -- Conv3d and InstanceNorm3d are run slice-wise through fi_conv2d_* / fi_bn_* by the host mirror.
 * nn.MaxPool3d(2) (unet_3D.py:35): y[N,D/2,H/2,W/2,C]; backward routes dy to the first maximum in (d,h,w) scan order. */
int fi_maxpool3d_fwd(int dtype, const void* x, void* y, int N, int D, int H, int W, int C, void* stream);
int fi_maxpool3d_bwd(int dtype, const void* x, const void* dy, void* dx, int N, int D, int H, int W, int C, void* stream);
/* dx = add + the routed gradient (add: the gradient the same tensor receives as a skip connection, same layout as x): the sum
 * autograd would make with an elementwise launch of its own. */
int fi_maxpool3d_bwd_add(int dtype, const void* x, const void* dy, const void* add, void* dx, int N, int D, int H, int W, int C,
                         void* stream);
/* nn.Upsample(scale_factor=(2,2,2), mode='trilinear') with align_corners=False (networks/utils.py:264):
 * [N,d,h,w,C] -> [N,2d,2h,2w,C];  backward = exact adjoint (gather over the 4x4x4 candidate outputs). */
int fi_upsample3d2x_fwd(int dtype, const void* x, void* y, int N, int d, int h, int w, int C, void* stream);
int fi_upsample3d2x_bwd(int dtype, const void* dy, void* dx, int N, int d, int h, int w, int C, void* stream);
/* Both one-launch operands of every 3x3x3 convolution of a model in ONE launch (the 3D path's fi_pack_weights_multi; Conv3d weights of
 * /root/reference/code/networks/unet_3D.py:20-94 / networks/utils.py:99-123): table = device rows of 6 int64 { fp32 parameter
 * [Cout][Cin][3][3][3], forward operand [Cout][9][3][Cin] (fi_conv3d_fwd_fused), dgrad operand [Cin][9][3][Cout] with all three filter
 * axes reversed (fi_conv3d_dgrad_fused), Cout, Cin, first block }, a block = 256 (co, ci) pairs of its tensor; nblocks = their sum.
 * 16-bit storage types; values rounded as fi_pack_weights rounds. */
int fi_pack_weights3d_multi(const long long* table, int ntensors, int nblocks, int dtype, void* stream);

/* ---------------------------------------------------------------- losses ------------------
 * CrossEntropyLoss(ignore_index) (/root/reference/code/flower_pCE_2D.py:57,124): logits fp32 NHWC
 * [M][C], labels uint8 [M].  acc is fp64 [FI_CE_SLOTS][2] (caller zeroes): workgroups add {sum of -log p[label],
 * #non-ignored} to slot (workgroup % FI_CE_SLOTS); fi_ce_finalize / fi_ce_bwd fold the slots. */
#define FI_CE_SLOTS 16
int fi_ce_fwd(const float* logits, const uint8_t* labels, long M, int C, int ignore_index, double* acc, void* stream);
/* loss[0] = sum_s acc[s][0] / sum_s acc[s][1] (fp32; NaN when nothing is labeled, as torch) */
int fi_ce_finalize(const double* acc, float* loss, void* stream);
/* dlogits = gscale * (softmax - onehot) / count for labeled pixels, 0 otherwise; written in `dtype`.
 * gscale: device fp32 scalar (upstream gradient) or NULL for 1. */
int fi_ce_bwd(const float* logits, const uint8_t* labels, long M, int C, int ignore_index, const double* acc,
              const float* gscale, void* dlogits, int dtype, void* stream);

/* pDLoss / DiceLoss (/root/reference/code/utils/losses.py:156-232).  probs: fp32 NHWC [B][HW][C] (softmax already
 * applied), labels uint8 [B][HW].  ignore_index >= 0: pDLoss incl. the reference's [B,B,H,W] mask broadcast
 * (every image's ignore mask multiplies every image's products); ignore_index < 0: plain DiceLoss.
 * acc (fp64 [C][3], caller zeroes) += {sum s*t*M, sum s*s*M, sum t*t*M};  loss = sum_c w_c*(1 - (2I+1e-5)/(Z+Y+1e-5))/C
 * (weight fp32 [C] or NULL = ones); dprobs = gscale * dloss/dprobs (gscale device fp32 scalar or NULL = 1). */
int fi_pdice_fwd(const float* probs, const uint8_t* labels, int B, long HW, int C, int ignore_index, double* acc,
                 void* stream);
int fi_pdice_finalize(const double* acc, const float* weight, int C, float* loss, void* stream);
int fi_pdice_bwd(const float* probs, const uint8_t* labels, int B, long HW, int C, int ignore_index, const double* acc,
                 const float* weight, const float* gscale, float* dprobs, void* stream);

/* Dice bookkeeping of val_2D.py:9-22,66-74: pred = argmax_c logits (first max wins, as torch.argmax);
 * for class i in 1..C-1: region = (lbl == 1) if i == 1 else (lbl >= 1); counts[(i-1)*3 + {0,1,2}] +=
 * {|P & G|, |P|, |G|}  (int64, atomically added, caller zeroes). */
int fi_dice_counts(const float* logits, const uint8_t* gt, long M, int C, long long* counts, void* stream);

/* Gated CRF loss, Potts model, no masks (/root/reference/code/utils/gate_crf_loss.py:20-124 as called at
 * flower_pCE_2D_GateCRFMsacleTreeEnergyLoss_Ours.py:143-150), fused: no unfolded tensors.
 *   y    fp32 [N][H][W][C] class probabilities (C <= 8);  feat fp32 [N][H][W][F] the `sample` modality (F <= 4)
 *   nk <= 4 kernels: weights[k], sigma_xy[k], sigma_sample[k] -- HOST arrays, read before the launch (a sigma <= 0
 *   leaves that modality out of kernel k)
 *   K(i,d) = sum_k w_k exp(-0.5 (|d_xy|^2/sigma_xy^2 + |feat(i+d)-feat(i)|^2/sigma_s^2)), K(i,0) = 0, |d| <= radius (<= 8);
 *   a neighbour outside the image counts with feat = 0, mesh position (0,0) and y = 0 (the reference's zero-padded unfold).
 *   prod[i][c] = sum_d K(i,d) y[i+d][c];  acc (fp64 [FI_CRF_SLOTS][2], caller zeroes) += { sum K, sum_c y[i][c] prod[i][c] }.
 * loss = (sum_s acc[s][0] - sum_s acc[s][1]) / (N*H*W);  d loss / d y = -2 prod / (N*H*W)  (K is symmetric). */
#define FI_CRF_SLOTS 16
int fi_gatedcrf_fwd(const float* y, const float* feat, int N, int H, int W, int C, int F, int radius, int nk,
                    const float* weights, const float* sigma_xy, const float* sigma_sample, float* prod, double* acc,
                    void* stream);

/* ---------------------------------------------------------------- tree filter (tree-energy loss) --------
 * The reference's `tree_filter_cuda` extension (/root/reference/code/utils/TreeEnergyLoss/kernels/lib_tree_filter/src:
 * mst_forward = host Boruvka behind a D2H/H2D round trip, bfs_forward, refine_forward / _backward_feature / _backward_weight)
 * as device-resident kernels.  Per-image planes [B][C][V] fp32, V = H*W row-major; "sorted" = breadth-first position.
 *
 * fi_tree_grid_weights: weight[b][e] = |fm[b][:,u] - fm[b][:,v]|^2 + 1 over the 4-neighbour grid, vertical pairs
 *   ((h,w),(h+1,w)) first, then horizontal ((h,w),(h,w+1))  (modules/tree_filter.py:14-34);  E = 2HW - H - W.
 * fi_tree_mst: minimum spanning tree under the total order (weight, edge index) -- the tree the reference's Boruvka
 *   (mst/boruvka.cpp:20-112: a tie goes to the first edge in list order) selects; edge_out int32 [B][V-1][2] in edge-index
 *   order (the reference emits the same SET in Boruvka order).  workspace: B * fi_tree_mst_workspace(H, W) bytes.
 * fi_tree_bfs: breadth-first order from vertex 0 (bfs/bfs.cu:19-98): sorted_index [B][V] (vertex at each position),
 *   sorted_parent [B][V] (position of the parent; 0 for the root), sorted_child [B][V][4] (positions, 0 = none);
 *   levels int32 [B][V+2]: levels[b][0] = number of levels L, levels[b][1+l] = first position of level l (.. [1+L] = V);
 *   adjacency_workspace int32 [B][V][4] (only touched when the image is too large for the LDS-resident traversal,
 *   ~512^2).  Deterministic (frontier order; neighbours up, down, left, right).  Test hooks (environment, read per
 *   call): FI_TREE_BFS_GLOBAL=1 forces the global-memory traversal; FI_TREE_CAP / FI_TREE_CHUNK shrink the LDS level
 *   cache / streamed chunk of the three recursions so that small images exercise their global-memory paths.
 * fi_tree_edge_weights: w[b][i] = exp(-|e[:,sorted_index[i]] - e[:,parent]|^2 * inv_sigma)   (tree_filter.py:92-110);
 *   fi_tree_edge_weights_bwd: grad_embed (original order) from grad_w (sorted).
 * fi_tree_aggr_up  : out[i] = x[sorted_index[i]] + sum_child out[child]*w[child]   (x == NULL: 1)   refine.cu:70-134
 * fi_tree_prop_down: out[sorted_index[i]] = x[i]*(1 - w[i]^2) + out[parent vertex]*w[i], w[root] := 0 refine.cu:19-68
 * fi_tree_grad_rec : edge-weight gradient recursion, in_grad propagated in place                    refine.cu:136-199
 *   (in_data / in_grad sorted, out_data original order; Cd data channels, Cg gradient channels, grad [B][max][V] sorted;
 *   Cd == Cg or Cd == 1 -- in_grad is propagated in place by one workgroup per gradient channel, else FI_ERR_UNSUPPORTED). */
long fi_tree_mst_workspace(int H, int W);
int fi_tree_grid_weights(const float* fm, int B, int C, int H, int W, float* weight, void* stream);
int fi_tree_mst(const float* weight, int B, int H, int W, int* edge_out, void* workspace, long workspace_bytes,
                void* stream);
int fi_tree_bfs(const int* edges, int B, int H, int W, int* sorted_index, int* sorted_parent, int* sorted_child,
                int* levels, int* adjacency_workspace, void* stream);
int fi_tree_edge_weights(const float* embed, const int* sorted_index, const int* sorted_parent, int B, int Ce, int V,
                         float inv_sigma, float* w, void* stream);
int fi_tree_edge_weights_bwd(const float* embed, const int* sorted_index, const int* sorted_parent, const int* sorted_child,
                             const float* w, const float* grad_w, int B, int Ce, int V, float inv_sigma, float* grad_embed,
                             void* stream);
int fi_tree_aggr_up(const float* x, const float* w, const int* sorted_index, const int* sorted_child, const int* levels,
                    int B, int C, int V, float* out, void* stream);
int fi_tree_prop_down(const float* x_sorted, const float* w, const int* sorted_index, const int* sorted_parent,
                      const int* levels, int B, int C, int V, float* out, void* stream);
int fi_tree_grad_rec(const float* in_data, float* in_grad, const float* out_data, const float* w, const int* sorted_index,
                     const int* sorted_parent, const int* levels, int B, int Cd, int Cg, int V, float* grad, void* stream);

/* InstanceNorm3d(affine=False) + ReLU of a whole batch in ONE launch per pass (round 5; /root/reference/code/networks/utils.py:
 * 106-110): `nbatch` samples, each with its own statistics (stats_stride doubles apart), coefficient rows (coef fp32 [nbatch][4][C],
 * coef_stride = 4*C floats between samples -- also the stride of the scale / shift / mean / invstd ROW pointers of the backward) and
 * slice of the activation (tensor_stride elements apart).  The per-tensor entry points' kernels and arithmetic; no dropout, no
 * affine gradients; gamma / beta = the caller's constant (1, 0) rows, the running-statistics pointers are scratch (momentum 0). */
int fi_bn_fused_fwd_batched(const FiBnAct* d, int nbatch, long tensor_stride, long stats_stride, long coef_stride, const void* y,
                            void* z, const double* stats, const float* gamma, const float* beta, float* running_scratch_mean,
                            float* running_scratch_var, float eps, float* coef, void* stream);
int fi_bn_act_bwd_reduce_batched(const FiBnAct* d, int nbatch, long tensor_stride, long stats_stride, long coef_stride,
                                 const void* dz, const void* y, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, double* sums, void* stream);
int fi_bn_act_bwd_apply_batched(const FiBnAct* d, int nbatch, long tensor_stride, long stats_stride, long coef_stride,
                                const void* dz, const void* y, const float* scale, const float* shift, const float* mean,
                                const float* invstd, const double* sums, int training, void* dy, void* stream);

/* The first convolution of the 3D U-Net, Conv3d(1 -> 16, 3x3x3, pad 1) (/root/reference/code/networks/unet_3D.py:38,
 * networks/utils.py:99-123), as ONE pass that streams the output once (the implicit-GEMM forms pad the 27-long contraction and make
 * three read-modify-write passes): on the matrix pipe with the taps (forward) / the voxels (filter gradient) as the contraction,
 * the fp32 filter split into three exact 16-bit parts (round 6; round 5's vector-ALU stencil gave the same results 2 x slower).  16-bit storage (FI_BF16 / FI_F16), x [N][D][H][W] (one channel),
 * w fp32 [16][27] with the taps in (kd, kh, kw) order, y / dy [N][D][H][W][16].
 * fwd: y = conv + bias, rounded to the storage type; stats (or NULL): fp64 [N][FI_STATS_SLOTS][16][2] (stats_stride doubles per
 *   sample), per-sample sum / sum of squares of the values AS STORED, added to one slot per workgroup (InstanceNorm3d's input).
 * wgrad: dw fp32 [16][27] and dbias fp32 [16] are ADDED to (either may be NULL); workspace: fi_conv3d_first_wgrad_workspace bytes,
 *   one partial slice per workgroup, summed in slice order by a second launch (deterministic). */
long fi_conv3d_first_wgrad_workspace(int N, int D, int H, int W);
int fi_conv3d_first_fwd(int dtype, int N, int D, int H, int W, const void* x, const float* w, const float* bias, void* y,
                        double* stats, long stats_stride, void* stream);
int fi_conv3d_first_wgrad(int dtype, int N, int D, int H, int W, const void* x, const void* dy, float* dw, float* dbias,
                          void* workspace, long workspace_bytes, void* stream);

/* The last convolution of the 3D U-Net, Conv3d(16 -> cout <= 4, 1x1x1) to fp32 logits (/root/reference/code/networks/unet_3D.py:57):
 * one streaming pass per direction (round 5).  x / dx [voxels][16] 16-bit (FI_BF16 / FI_F16), w fp32 [cout][16], y / dy fp32
 * [voxels][cout] -- the loss's fp32 gradient is consumed as it is.  wgrad: dw fp32 [cout][16] and dbias fp32 [cout] are ADDED to
 * (either may be NULL); workspace: fi_conv3d_point_wgrad_workspace() bytes (one partial row per workgroup, fixed-order sum). */
long fi_conv3d_point_wgrad_workspace(void);
int fi_conv3d_point_fwd(int dtype, long voxels, int cout, const void* x, const float* w, const float* bias, float* y, void* stream);
int fi_conv3d_point_dgrad(int dtype, long voxels, int cout, const float* dy, const float* w, void* dx, void* stream);
int fi_conv3d_point_wgrad(int dtype, long voxels, int cout, const void* x, const float* dy, float* dw, float* dbias,
                          void* workspace, long workspace_bytes, void* stream);

/* Elementwise glue of the tree-energy losses (round 5; through round 4 these were ATen launches), NCHW fp32:
 * fi_tree_prep_fwd: everything a loss computes before its trees, ONE launch over the N*H*W output pixels --
 *   prob = softmax(preds, dim=1)                                   (flower_common.py:665,717,781; preds == NULL: skipped);
 *   maps[k].dst [N][C_k][H][W] = F.interpolate(maps[k].src [N][C_k][h_k][w_k], size=(H, W), mode='bilinear',
 *     align_corners=False) for k < nmaps <= FI_TREE_MAPS          (:659,672,711-714,775-778: the low-level image and the head maps);
 *   rois [N][1][H][W] = F.interpolate(unlabeled_ROIs.unsqueeze(1).float(), size=(H, W), mode='nearest') of a uint8 / bool mask
 *     [N][roi_h][roi_w] and count[0] += rois.sum()                (:660-662; count: fp64, zeroed by the caller; roi_src == NULL: skipped).
 * fi_tree_prep_bwd: dpreds = softmax' (prob, dprob) and, per map, the gradient w.r.t. its source (maps[k].src = gradient w.r.t.
 *   the resized map [N][C_k][H][W], maps[k].dst = gradient w.r.t. the source, WRITTEN); gathers, no atomics: deterministic.
 * fi_tree_masked_l1_fwd: loss[0] = weight * (sum_k sum(rois * |prob - as[k]|)) / max(count, 1), nterms <= FI_TREE_TERMS maps
 *   (:682-686 one term; :745-751 three terms added; `if N > 0: loss /= N` without a host sync: N == 0 => the sums are 0).
 *   acc_zeroed: FI_TREE_TERMS + 1 doubles, zeroed by the caller (per-term sums, arrival ticket of the finishing workgroup).
 * fi_tree_masked_l1_bwd: dprob (WRITTEN, may be NULL) and das[k] (WRITTEN where non-NULL) for the upstream gradient grad_out[0].
 * fi_tv_loss_fwd / _bwd: tv_loss (:636-643): eroded = -max_pool2d(-p, 3, 1, 1), contour = relu(max_pool2d(eroded, 3, 1, 1) -
 *   eroded), acc[0] += sum |contour| (the caller divides by the element count); max_pool2d's first-extremum-wins tie rule is
 *   kept in the saved window indices (uint8 per element), so the backward routes gradients as torch's does. */
#define FI_TREE_MAPS 4
#define FI_TREE_TERMS 3
typedef struct FiTreeMap {
  const float* src;            /* forward: [N][C][h][w] through `stride` (elements; any layout -- head maps arrive as NCHW views */
  float* dst;                  /*   of NHWC tensors); dst dense NCHW [N][C][H][W].  backward: see fi_tree_prep_bwd (dense both) */
  long stride[4];              /* n, c, y, x element strides of src (forward only)                                           */
  int C, h, w;
} FiTreeMap;
int fi_tree_prep_fwd(const float* preds, const long* preds_strides /* n, c, y, x */, float* prob, int N, int C, int H, int W,
                     const FiTreeMap* maps, int nmaps, const unsigned char* roi_src, int roi_h, int roi_w, float* rois,
                     double* count, void* stream);
int fi_tree_prep_bwd(const float* prob, const float* dprob, float* dpreds, int N, int C, int H, int W, const FiTreeMap* maps,
                     int nmaps, void* stream);
int fi_tree_masked_l1_fwd(const float* prob, const float* const* as, int nterms, const float* rois, int N, int C, int H, int W,
                          const double* count, float weight, double* acc_zeroed, float* loss, void* stream);
int fi_tree_masked_l1_bwd(const float* prob, const float* const* as, int nterms, const float* rois, int N, int C, int H, int W,
                          const double* count, float weight, const float* grad_out, float* dprob, float* const* das, void* stream);
int fi_tv_loss_fwd(const float* p, long planes, int H, int W, float* eroded, unsigned char* idx_e, unsigned char* idx_d,
                   unsigned char* positive, double* acc_zeroed, void* stream);
int fi_tv_loss_bwd(const unsigned char* idx_e, const unsigned char* idx_d, const unsigned char* positive, const float* grad_out,
                   long planes, int H, int W, float* scratch, float* dp, void* stream);

/* Surface distances for medpy.metric.binary.hd95 (/root/reference/code/val_2D.py:14): border(m) = m AND NOT
 * erode(m) with the 4-neighbourhood (connectivity 1, outside = background).  fi_seg_borders appends the flat pixel
 * indices of the border of the prediction (argmax of logits [H*W][C]; class k=1: ==1, k>=2: >=1, val_2D.py:66-74) to
 * pred_list and of the ground truth to gt_list (int32 [H*W] each, order unspecified) and adds their numbers to
 * counts[0], counts[1] (int32, caller zeroes).  fi_surface_distances: out[i] (fp64, i < counts[from_index]) = Euclidean
 * distance from from_list[i] to the nearest pixel of to_list (exhaustive, exact); max_from bounds the launch. */
int fi_seg_borders(const float* logits, const uint8_t* gt, int H, int W, int C, int k, int* pred_list, int* gt_list,
                   int* counts, void* stream);
int fi_surface_distances(const int* from_list, const int* to_list, const int* counts, int from_index, int to_index,
                         int W, int max_from, double* out, void* stream);

/* Training-time augmentation of a batch drawn from an HBM-resident data set
 * (/root/reference/code/dataloaders/dataset.py:190-256: RandomGenerator = random_rot_flip then random_rotate, each
 * optional per sample), as ONE gather: out[b] = rotate(flip(rot90(src[ip[b][0]], k), axis), angle).
 *   src_img fp32 [n][C][H][W], src_lab uint8 [n][H][W] -> out_img fp32 [B][C][H][W], out_lab uint8 [B][H][W]
 *   ip int32 [B][4] = { source sample, k of np.rot90 (0..3; -1: no rot/flip), spatial flip axis (0 rows, 1 columns),
 *                       rotate (0/1) }
 *   dp fp64  [B][6] = { m00, m01, m10, m11, off0, off1 } of scipy.ndimage.rotate(order=0, reshape=False): the input
 *                     coordinate of output pixel (i,j) is ((0 + i*m0) + j*m1) + off in fp64 WITHOUT contraction,
 *                     outside [0, len-1] -> cval (img_cval / lab_cval), else floor(x + 0.5) -- bit-exact with scipy.
 * Odd k needs H == W (rot90 would change the shape): the caller checks, the entry cannot see ip. */
int fi_augment2d(const float* src_img, const uint8_t* src_lab, const int* ip, const double* dp, float* out_img,
                 uint8_t* out_lab, int B, int C, int H, int W, float img_cval, int lab_cval, void* stream);

/* ---------------------------------------------------------------- optimizer ----------------
 * torch.optim.AdamW(betas, eps, weight_decay, amsgrad=False) as created at flower_pCE_2D.py:55.
 * All scalars live on the device so that a captured hipGraph can be replayed:
 *   lr_state (fp64[1])  current learning rate
 *   step     (int32[1]) this parameter group's AdamW step count t (the reference re-creates the
 *                       optimizer every round, and FedICRA's freeze schedule gives out_conv and
 *                       the rest different t -- SURVEY.md 8-a6/a15)
 *   hyper    (fp32[4])  { lr, 1 - lr*wd, lr / (1 - beta1^t), sqrt(1 - beta2^t) }
 * fi_adamw_hyper: ++step[0]; hyper <- f(lr_state[0], step[0]).
 * fi_lr_poly_advance: ++iter[0]; lr_state[0] = base_lr*(1 - iter/max_iter)^0.9 (flower_pCE_2D.py:154-157). */
int fi_adamw_hyper(int* step, float* hyper, const double* lr_state, float beta1, float beta2, float wd, void* stream);
int fi_lr_poly_advance(int* iter, double* lr_state, double base_lr, double max_iter, void* stream);
/* p, m, v updated in place over [0,n); g read.  shadow (bf16, may be NULL) receives a bf16 copy of p.
 * A negative hyper[0] makes the call a no-op (see fi_amp_guard). */
int fi_adamw_step(float* p, const float* g, float* m, float* v, long n, const float* hyper, float beta1, float beta2,
                  float eps, void* shadow_bf16, void* stream);

/* torch.optim.SGD(lr, momentum, weight_decay) of the single-site trainer (/root/reference/code/Unet_pCE.py:88-89):
 * g' = g + wd*p; buf = momentum*buf + g' (buf zero-initialised); p -= lr_state[0]*buf.  skip_hyper (may be NULL): a
 * negative skip_hyper[0] makes the call a no-op (fi_amp_guard). */
int fi_sgd_step(float* p, const float* g, float* momentum_buf, long n, const double* lr_state, float momentum,
                float weight_decay, const float* skip_hyper, void* stream);

/* Dynamic loss scaling = torch.cuda.amp.GradScaler (`--amp 1`: /root/reference/code/flower_pCE_2D.py:47-48,143-146,
 * flower_common.py:466-468,576-584), with every scalar on the device (hipGraph-capturable):
 *   scale (fp32[1]), growth_tracker (int32[1]), found_inf (fp32[1], 0 or 1).
 * fi_amp_unscale: grads[i] *= 1/scale; found_inf = 1 when any result is inf/NaN        (scaler.unscale_)
 * fi_amp_guard  : after fi_adamw_hyper -- when found_inf, undo the step-count increment and set hyper[0] = -1 so that
 *                 fi_adamw_step does nothing                                            (scaler.step skips optimizer.step)
 * fi_amp_update : found_inf ? (scale *= backoff, tracker = 0) : (++tracker == interval ? scale *= growth, tracker = 0);
 *                 then found_inf = 0                                                    (scaler.update) */
int fi_amp_unscale(float* grads, long n, const float* scale, float* found_inf, void* stream);
int fi_amp_guard(int* step, float* hyper, const float* found_inf, void* stream);
int fi_amp_update(float* scale, int* growth_tracker, float* found_inf, float growth_factor, float backoff_factor,
                  int growth_interval, void* stream);

/* ---------------------------------------------------------------- aggregation helpers ------
 * flwr `aggregate`: w = reduce(add, [w_k * n_k]) / total.
 * fi_scale: y = divide ? x / a : x * a   (pre-scale by n_k before, divide by total after the RCCL sum)
 * fi_axpy : acc = acc + a * x with the product rounded to fp32 BEFORE the add (no FMA contraction),
 *           i.e. numpy's  acc + (x * n_k)  bit for bit -- used by the single-process aggregator. */
int fi_scale(const float* x, float* y, long n, float a, int divide, void* stream);
int fi_axpy(float* acc, const float* x, long n, float a, void* stream);
/* The FedOpt server optimizers the reference can select (flower_common.py:432-448 -> flwr 1.0.0 FedAdagrad / FedAdam /
 * FedYogi; third-party and absent: restated from the published source, parity unpinned) on the flat fp32 state:
 *   delta = agg - cur;  m = beta1*m + (1-beta1)*delta;
 *   v = v + delta^2 (mode 0) | beta2*v + (1-beta2)*delta^2 (1) | v - (1-beta2)*delta^2*sign(v - delta^2) (2);
 *   cur = cur + eta*m / (sqrt(v) + tau)          -- every product and sum rounded to fp32, numpy's evaluation order. */
int fi_fedopt_step(int mode, float* cur, const float* agg, float* m, float* v, long n, float eta, float beta1,
                   float one_minus_beta1, float beta2, float one_minus_beta2, float tau, void* stream);
/* FedICRA ALA element-wise step (/root/reference/code/flower_common.py:590-602):
 *   w    = clamp(w - eta * grad * (local - global), 0, 1)
 *   temp = global + (local - global) * w
 * skip (device fp32[1], may be NULL): when non-zero nothing is written -- the GradScaler's found-inf flag of an `--amp 1`
 * batch whose scaled gradients overflowed (:576-584; the reference would feed the inf/NaN gradients into w).           */
int fi_ala_update(float* w, float* temp, const float* grad, const float* local, const float* global, long n,
                  float eta, const float* skip, void* stream);

/* ---------------------------------------------------------------- PCS helpers --------------
 * PersonalizedChannelSelection (unet.py:103-144): global avg / max pool over H*W per (n,c);
 * amax (int32 [N,C], may be NULL) = pixel index of the first maximum (AdaptiveMaxPool2d backward). */
int fi_global_avgmax(int dtype, const void* x, float* avg, float* mx, int* amax, int N, int HW, int C, void* stream);
/* Split form of the same pooling for large maps (networks/unet.py:96-100 pools the full-resolution feature map): pixel
 * ranges reduced by separate workgroups into a caller workspace and folded in range order -- deterministic, ties keep the
 * first pixel.  fi_global_avgmax_ranges -> S (0: shape not worth splitting / unsupported: use fi_global_avgmax);
 * workspace >= N * S * 3 * C floats. */
int fi_global_avgmax_ranges(int dtype, int N, int HW, int C);
int fi_global_avgmax_split(int dtype, const void* x, float* avg, float* mx, int* amax, int N, int HW, int C,
                           float* workspace, long workspace_bytes, void* stream);
/* y = x * (1 + h[n][c])   (x*h + x) */
int fi_channel_gate_fwd(int dtype, const void* x, const float* h, void* y, int N, int HW, int C, void* stream);
/* dx = dy*(1+h) + davg[n][c]/HW + (pixel == amax[n][c] ? dmx[n][c] : 0);
 * dh[n][c] = sum_hw dy*x.  amax/davg/dmx may be NULL. */
int fi_channel_gate_bwd(int dtype, const void* x, const void* dy, const float* h, const int* amax, const float* davg,
                        const float* dmx, void* dx, float* dh, int N, int HW, int C, void* stream);

/* Diagnostic: one wavefront executes ds_read_b64_tr_b16 on a 1024-element int16 LDS image of `in`; lane l
 * addresses element offs[l] (multiple of 4); out[l*4+j] = j-th returned element.  Pins the transpose-read
 * semantics the bf16 wgrad kernel relies on (tests/test_ops_gpu.py::test_tr16_semantics). */
int fi_probe_tr16(const short* in, const int* offs, short* out, void* stream);

/* misc elementwise */
int fi_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream);
/* NCHW fp32 (reference layout) <-> NHWC `dtype` */
int fi_nchw_to_nhwc(const float* src, void* dst, int dtype, int N, int C, int H, int W, void* stream);
int fi_nhwc_to_nchw(const void* src, int dtype, float* dst, int N, int C, int H, int W, void* stream);

/* PersonalizedChannelSelection's gate (/root/reference/code/networks/unet.py:103-144) as one launch per direction:
 *   e = fc1(onehot(who)),  h = sigmoid(fc2([avg ; e]) + fc2([max ; e]))       fc1: K -> C -> C, fc2: 2C -> C/16 -> C, ReLU between,
 * no biases.  avg / mx / h: fp32 [B][C] (fi_global_avgmax); who: int32 [B] embedding index per image; w1a [C][K], w1b [C][C],
 * w2a [C/16][2C], w2b [C][C/16]: the four 1x1-conv weights as stored; hidden: fp32 [B][2][C/16], kept for backward.  C % 16 == 0,
 * C <= 512.  The PCS weights are frozen in the reference (never registered: unet.py:172-177): backward yields d/d avg, d/d max. */
int fi_pcs_gate_fwd(const float* avg, const float* mx, const int* who, const float* w1a, const float* w1b, const float* w2a,
                    const float* w2b, float* h, float* hidden, int B, int C, int K, void* stream);
int fi_pcs_gate_bwd(const float* dh, const float* h, const float* hidden, const float* w2a, const float* w2b, float* davg,
                    float* dmx, int B, int C, void* stream);

/* FedICRA's LC loss and the total it enters (/root/reference/code/flower_pCE_2D.py:128-139):
 *   out[1] = loss_lc = -(1/G) sum_g mean((h - others[g])^2),   out[0] = loss_ce[0] + alpha * loss_lc
 * h: fp32 [n] (the client's own heat-map), others: fp32 [G][n] (the heat-maps under the other clients' embeddings, no gradient),
 * loss_ce: device scalar.  dcoef fp32 [n] = d loss_lc / d h, kept for fi_lc_loss_bwd: dh = g[0] * alpha * dcoef. */
int fi_lc_loss_fwd(const float* h, const float* others, const float* loss_ce, float alpha, int G, int n, float* out,
                   float* dcoef, void* stream);
int fi_lc_loss_bwd(const float* dcoef, const float* g, float alpha, float* dh, int n, void* stream);

/* ---------------------------------------------------------------- defined-but-unused module surface -----------
 * ConvTranspose{2,3}d(kernel 2, stride 2) -- UpBlock(bilinear=False) (/root/reference/code/networks/unet.py:60-62, never
 * selected by the reference's decoders) and VNet's UpsamplingDeconvBlock (networks/vnet.py:94-118) -- has no overlapping
 * taps: it is ONE 1x1 convolution to P*Cout channels ordered [tap][co] (P = 4: (a, b); 8: (c, a, b)), run by
 * fi_conv2d_fwd / _wgrad, followed by this depth-to-space shuffle:
 *   packed [N][D][H][W][P][C]  <->  spatial [N][(2)D][2H][2W][C]   (three_d = 0: D must be 1; inverse: spatial -> packed). */
int fi_depth_to_space2x(int dtype, const void* src, void* dst, int N, int D, int H, int W, int C, int three_d, int inverse,
                        void* stream);
/* The whole operator in one call each (replaces nn.ConvTranspose2d / ConvTranspose3d(k 2, s 2) forward and the two halves
 * of its backward).  x [N][D][H][W][Cin], y / dy [N][(2)D][2H][2W][Cout]; `packed` = caller-owned scratch of
 * N*D*H*W*P*Cout elements of `dtype`.  w_packed / wt_packed: fi_pack_weights (mode 0 / 1) of W'[(tap, co)][ci] =
 * weight[ci][co][tap]; bias_taps fp32 [P*Cout] (the bias repeated per tap) or NULL.
 *   dgrad: un-shuffles dy into `packed` (kept for the wgrad call) and, when dx != NULL, runs the 1x1 dgrad GEMM;
 *   wgrad: dw fp32 [P*Cout][Cin] and dbias_taps fp32 [P*Cout] are ADDED to (caller zeroes; sum dbias over the taps);
 *          workspace as for fi_conv2d_wgrad, size from fi_convtranspose2x_wgrad_workspace (< 0: FI_ERR_*). */
int fi_convtranspose2x_fwd(int dtype, int N, int D, int H, int W, int cin, int cout, int three_d, const void* x,
                           const void* w_packed, const float* bias_taps, void* packed, void* y, void* stream);
int fi_convtranspose2x_dgrad(int dtype, int N, int D, int H, int W, int cin, int cout, int three_d, const void* dy,
                             const void* wt_packed, void* packed, void* dx, void* stream);
long fi_convtranspose2x_wgrad_workspace(int dtype, int N, int D, int H, int W, int cin, int cout, int three_d);
int fi_convtranspose2x_wgrad(int dtype, int N, int D, int H, int W, int cin, int cout, int three_d, const void* x,
                             const void* dy_packed, float* dw, float* dbias_taps, void* workspace, long workspace_bytes,
                             void* stream);
/* GroupNorm(G, C) (+ optional ReLU) over dense channel-last samples [N][pixels][C] (networks/vnet.py:5-31 with
 * normalization='groupnorm'; InstanceNorm is G = C without affine): statistics in fp64 per (sample, group), mean / invstd
 * [N][G] saved for the backward.  bwd: dx, and dgamma / dbeta ATOMICALLY ADDED (either may be NULL); z is only read when
 * relu != 0 (the ReLU mask).  C / G <= 256. */
/* stats[slot][c][2] += per-channel (sum, sum of squares) of x [pixels][C] (C <= 256): the accumulator layout of
 * fi_bn_finalize / fi_bn_fused_fwd, for a BatchNorm that does not sit behind a 2D convolution epilogue (BatchNorm3d of
 * VNet, networks/vnet.py:16-17).  Caller zeroes stats. */
int fi_channel_stats(int dtype, const void* x, double* stats, long pixels, int C, void* stream);
int fi_groupnorm_fwd(int dtype, const void* x, void* z, const float* gamma, const float* beta, float* mean, float* invstd,
                     int N, long pixels, int C, int G, float eps, int relu, void* stream);
int fi_groupnorm_bwd(int dtype, const void* dz, const void* x, const void* z, const float* gamma, const float* mean,
                     const float* invstd, void* dx, float* dgamma, float* dbeta, int N, long pixels, int C, int G, int relu,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FEDICRA_HIP_H */
