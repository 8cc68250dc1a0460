"""ctypes loader for libfedicra_hip.so (the C ABI declared in include/fedicra_hip.h).

The product path has NO CPU fallback: if the shared library is missing this module raises at
import time of any op (``lib()``), and every wrapper checks the return code.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfedicra_hip.so")
LIB_PATH = os.environ.get("FEDICRA_HIP_LIB", LIB_PATH)        # another build of the same C ABI (kernel A/B runs)

FI_F32, FI_BF16, FI_F16 = 0, 1, 2
STATS_SLOTS = 8                # FI_STATS_SLOTS in include/fedicra_hip.h
DROP_NONE, DROP_MASK_ELEM, DROP_RNG_ELEM, DROP_MASK_CHAN, DROP_RNG_CHAN = 0, 1, 2, 3, 4

EXPORTS = [
    "fi_abi_version", "fi_conv_weight_chunk16", "fi_conv3d_wgrad_fused", "fi_conv3d_wgrad_fused_workspace", "fi_pcs_gate_fwd", "fi_pcs_gate_bwd", "fi_lc_loss_fwd", "fi_lc_loss_bwd", "fi_conv3d_fwd_fused", "fi_conv3d_dgrad_fused", "fi_conv3d_tuning", "fi_global_avgmax_ranges", "fi_global_avgmax_split", "fi_conv2d_fwd", "fi_conv2d_fwd_fused", "fi_bn_finalize_groups", "fi_conv_tuning", "fi_conv2d_wgrad", "fi_conv2d_wgrad_workspace", "fi_conv2d_wgrad_partial",
    "fi_wgrad_reduce_multi", "fi_wgrad_permute3d_multi", "fi_pack_weights",
    "fi_pack_weights_multi", "fi_bn_fused_fwd", "fi_bn_finalize", "fi_bn_act_fwd",
    "fi_bn_act_bwd_reduce", "fi_bn_act_bwd_apply", "fi_maxpool2_fwd", "fi_maxpool2_bwd", "fi_maxpool2_bwd_add", "fi_upsample2x_fwd",
    "fi_upsample2x_bwd", "fi_maxpool3d_fwd", "fi_maxpool3d_bwd", "fi_maxpool3d_bwd_add", "fi_conv3d_wgrad_fused_partial", "fi_upsample3d2x_fwd", "fi_upsample3d2x_bwd", "fi_ce_fwd", "fi_ce_finalize", "fi_ce_bwd", "fi_pdice_fwd", "fi_pdice_finalize",
    "fi_pdice_bwd", "fi_dice_counts", "fi_gatedcrf_fwd", "fi_tree_mst_workspace", "fi_tree_grid_weights", "fi_tree_mst", "fi_tree_bfs", "fi_tree_edge_weights", "fi_tree_edge_weights_bwd", "fi_tree_aggr_up", "fi_tree_prop_down", "fi_tree_grad_rec", "fi_seg_borders", "fi_surface_distances", "fi_augment2d", "fi_fedopt_step", "fi_depth_to_space2x", "fi_groupnorm_fwd", "fi_groupnorm_bwd", "fi_channel_stats", "fi_conv3d_fwd", "fi_conv3d_dgrad", "fi_conv3d_wgrad_workspace", "fi_conv3d_wgrad", "fi_convtranspose2x_fwd", "fi_convtranspose2x_dgrad", "fi_convtranspose2x_wgrad_workspace",
    "fi_convtranspose2x_wgrad", "fi_adamw_hyper",
    "fi_lr_poly_advance", "fi_adamw_step", "fi_sgd_step", "fi_amp_unscale", "fi_amp_guard", "fi_amp_update", "fi_scale", "fi_axpy", "fi_ala_update", "fi_global_avgmax", "fi_channel_gate_fwd",
    "fi_channel_gate_bwd", "fi_cast", "fi_nchw_to_nhwc", "fi_nhwc_to_nchw", "fi_probe_tr16",
    "fi_conv2d_stats_xcorr", "fi_conv2d_stats_xcorr_workspace", "fi_conv2d_stats_xcorr_layout", "fi_wgrad_tuning", "fi_bn_act_pool_groups", "fi_narrow_tuning", "fi_conv1x1_up2x_fwd", "fi_upfuse_tuning", "fi_pack_weights3d_multi",
    "fi_bn_running_groups_multi", "fi_tree_prep_fwd", "fi_tree_prep_bwd", "fi_tree_masked_l1_fwd", "fi_tree_masked_l1_bwd",
    "fi_tv_loss_fwd", "fi_tv_loss_bwd", "fi_conv3d_first_fwd", "fi_conv3d_first_wgrad", "fi_conv3d_first_wgrad_workspace",
    "fi_conv3d_point_fwd", "fi_conv3d_point_dgrad", "fi_conv3d_point_wgrad", "fi_conv3d_point_wgrad_workspace",
    "fi_bn_fused_fwd_batched", "fi_bn_act_bwd_reduce_batched", "fi_bn_act_bwd_apply_batched",
]


class FiConv(C.Structure):
    _fields_ = [("dtype", C.c_int), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("ksize", C.c_int),
                ("c0", C.c_int), ("c1", C.c_int), ("co0", C.c_int), ("co1", C.c_int), ("accumulate0", C.c_int),
                ("accumulate1", C.c_int), ("y_f32", C.c_int), ("w16", C.c_void_p), ("w16_rows", C.c_int)]


class FiInXform(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("slope", C.c_float), ("pool", C.c_int),
                ("drop_mode", C.c_int), ("drop_p", C.c_float), ("seed", C.c_uint64), ("seed_group_stride", C.c_uint64),
                ("seed_offset", C.c_void_p)]


class FiBnAct(C.Structure):
    _fields_ = [("dtype", C.c_int), ("pixels", C.c_long), ("C", C.c_int), ("hw", C.c_int), ("slope", C.c_float),
                ("drop_mode", C.c_int), ("drop_p", C.c_float), ("seed", C.c_uint64), ("mask", C.c_void_p),
                ("seed_offset", C.c_void_p)]


FI_ERR_UNSUPPORTED = -3                                    # include/fedicra_hip.h


class FiError(RuntimeError):
    pass


_lib = None


ABI_VERSION = 6             # include/fedicra_hip.h FI_ABI_VERSION


def source_hash():
    """sha256[:16] over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h, sorted by name): names the BUILD a
    measurement belongs to on a box that has no .git -- bench.py refuses a committed PMC traffic file whose recorded hash
    differs from the tree it runs in (traffic taken three kernel commits before the benchmarked build)."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")) +
                   glob.glob(os.path.join(os.path.dirname(here), "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FiError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C fedicra_amd/csrc).  fedicra_amd has no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.fi_abi_version.restype = C.c_int
        got = _lib.fi_abi_version()
        if got != ABI_VERSION:
            _lib = None
            raise FiError(f"{LIB_PATH} reports C-ABI version {got}, this host mirror was written against {ABI_VERSION} "
                          "(include/fedicra_hip.h FI_ABI_VERSION): rebuild with `make -C fedicra_amd/csrc`")
        for name in EXPORTS:
            getattr(_lib, name).restype = C.c_int
        _lib.fi_conv2d_wgrad_workspace.restype = C.c_long
        _lib.fi_conv3d_wgrad_fused_workspace.restype = C.c_long
        _lib.fi_tree_mst_workspace.restype = C.c_long
        _lib.fi_conv2d_stats_xcorr_workspace.restype = C.c_long
        _lib.fi_conv3d_first_wgrad_workspace.restype = C.c_long
        _lib.fi_conv3d_point_wgrad_workspace.restype = C.c_long
    return _lib


def _chk(rc: int, what: str):
    if rc != 0:
        raise FiError(f"{what} failed with code {rc}" + (" (hipError_t)" if rc > 0 else " (FI_ERR_*)"))


def dt(t: torch.dtype) -> int:
    if t == torch.float32:
        return FI_F32
    if t == torch.bfloat16:
        return FI_BF16
    if t == torch.float16:
        return FI_F16
    raise FiError(f"unsupported dtype {t}")


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t):
    if not t.is_cuda:
        raise FiError("fedicra_amd ops need device tensors (no CPU path)")
    return t


# ------------------------------------------------------------------ per-kernel timing (bench.py roofline leg)
class KernelProfile:
    """HIP-event timing of individual C-ABI launches on the stream they are issued on.  Only active
    between profile_begin()/profile_end(); zero cost otherwise.  Each record carries the launch's
    ALGORITHMIC flops and bytes (DESIGN.md 'roofline accounting') so that bench.py can price it.

    Two things keep the host out of the numbers (an eager launch through ctypes costs ~8 us of host time, more than
    most of these kernels run for, so a plain bracket measures the GPU waiting for Python):
      * block(): at the start of every profiled iteration the stream is held by a spin kernel long enough for the
        host to enqueue the whole iteration; the GPU then runs launches and event records back to back;
      * the cost of the bracket itself (calibrate()) is subtracted from every interval."""

    BLOCK_CYCLES = 60_000_000          # torch.cuda._sleep argument: ~30 ms, several times one iteration's enqueue time

    def __init__(self):
        self.rows = []
        self.overhead_ms = 0.0

    def block(self):
        torch.cuda._sleep(self.BLOCK_CYCLES)

    def calibrate(self):
        """Bracket cost = interval around ONE tiny kernel minus the per-launch cost of that kernel when 20 of them
        share a bracket (slope).  What stays in a measured interval is the kernel plus one back-to-back dispatch gap
        (~1.5 us), which rocprofv3's begin/end stamps do not contain."""
        x = torch.zeros(1, device="cuda")
        tiny = lambda: scale(x, x, 1.0)
        tiny()

        def bracket(k, reps):
            ev = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(k):
                    tiny()
                e1.record()
                ev.append((e0, e1))
            torch.cuda.synchronize()
            v = sorted(a.elapsed_time(b) for a, b in ev)
            return v[len(v) // 2]

        self.block()
        t1 = bracket(1, 60)
        self.block()
        t20 = bracket(20, 15)
        per = (t20 - t1) / 19.0
        self.overhead_ms = max(t1 - per, 0.0)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, flops, nbytes, e0, e1, xflops in self.rows:
            a = agg.setdefault(key, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "xflops": 0.0})
            a["calls"] += 1
            a["ms"] += max(e0.elapsed_time(e1) - self.overhead_ms, 0.0)
            a["flops"] += flops                          # ALGORITHMIC work of the layer (the reference-defined count)
            a["bytes"] += nbytes
            a["xflops"] += xflops                        # work the launch EXECUTES (differs where an algebraic form replaces the layer)
        return agg


_prof = None


def profile_begin(subtract_overhead=True):
    """subtract_overhead=False: raw event intervals (what bench.py's roofline object reports, so that it can be checked
    against a rocprofv3 kernel trace without any correction)."""
    global _prof
    p = KernelProfile()
    if subtract_overhead:
        p.calibrate()
    _prof = p
    return _prof


def profile_end():
    global _prof
    p, _prof = _prof, None
    return p


def profile_block():
    """Called at the start of an iteration (ops.begin_iteration): no-op unless a profile is being taken."""
    if _prof is not None:
        _prof.block()


class _Timed:
    __slots__ = ("key", "flops", "bytes", "e0", "xflops")

    def __init__(self, key, flops, nbytes, xflops):
        self.key, self.flops, self.bytes, self.xflops = key, flops, nbytes, xflops

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def __exit__(self, *a):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        _prof.rows.append((self.key, self.flops, self.bytes, self.e0, e1, self.xflops))


class _NoTime:
    def __enter__(self):
        pass

    def __exit__(self, *a):
        pass


_NOTIME = _NoTime()


def _timed(kind, shape_key, flops, nbytes, executed_flops=None):
    if _prof is None:
        return _NOTIME
    return _Timed((kind,) + tuple(shape_key), float(flops), float(nbytes), float(flops if executed_flops is None else executed_flops))


def _esz(t):
    return t.element_size()


# ------------------------------------------------------------------ thin wrappers
def conv_weight_chunk16(dtype, ksize, cin, cout):
    """True when a filter of this shape gets the chunk-major second operand (fi_pack_weights mode 2 / 3; for the dgrad
    operand pass the conv's cout as cin and vice versa)."""
    return bool(lib().fi_conv_weight_chunk16(dt(dtype), int(ksize), int(cin), int(cout)))


def _attach_w16(d, w):
    """The chunk-major sibling of a packed operand travels as a python attribute of the operand tensor (ops._packed,
    flat.FlatStoreMixin._fi_refresh_packs set it): FiConv.w16 / w16_rows."""
    w16 = getattr(w, "_fi_w16", None)
    if w16 is not None:
        d.w16 = w16.data_ptr()
        d.w16_rows = int(getattr(w, "_fi_w16_rows", 0))


def conv2d_fwd(x0, x1, w, bias, y0, y1, stats, *, ksize, acc0=False, acc1=False, y_f32=False, tag="conv_fwd", cout=None):
    """x*: [N,H,W,C] dense NHWC; w: packed [Cout][k*k][Cin] (any shape, dense) in x0.dtype.  y0 = None with `cout` and
    `stats` given: statistics-only launch (nothing stored)."""
    _dev(x0)
    N, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    co0 = int(cout) if y0 is None else y0.shape[3]
    co1 = 0 if y1 is None else y1.shape[3]
    d = FiConv(dt(x0.dtype), N, H, W, ksize, c0, c1, co0, co1, int(acc0), int(acc1), int(y_f32))
    _attach_w16(d, w)
    cin, cout, px = c0 + c1, co0 + co1, N * H * W
    with _timed(tag, (str(x0.dtype)[6:], N, H, W, cin, cout, ksize), 2.0 * px * cin * cout * ksize * ksize,
                px * cin * _esz(x0) + (0 if y0 is None else px * cout * _esz(y0)) + cin * cout * ksize * ksize * _esz(x0)):
        _chk(lib().fi_conv2d_fwd(C.byref(d), ptr(x0), ptr(x1), ptr(w), ptr(bias), ptr(y0), ptr(y1), ptr(stats),
                                 stream()), "fi_conv2d_fwd")


def conv_tuning(v2=-1, nf=0, ck=0, wgs_per_cu=0):
    """fi_conv_tuning: measurement / test hook (which forward kernel is launched; -1 = the library's per-layer choice)."""
    _chk(lib().fi_conv_tuning(int(v2), int(nf), int(ck), int(wgs_per_cu)), "fi_conv_tuning")


def in_xform(coef, slope, *, pool=False, drop=None, seed_group_stride=0, group0=0):
    """FiInXform for a source that holds a raw conv output: coef = fp32 [2][G][C] from bn_finalize_groups (None: the source
    is used as it is); drop = (mode, p, seed, mask, seed_offset) as ops._drop_spec returns it (RNG element mode only).
    group0: the launch covers the images of groups group0, group0 + 1, ... only (its group 0 reads coefficient row group0; no
    dropout draws in that form: their seeds are numbered from the launch's first group)."""
    if coef is None:
        return None
    G, Cc = coef.shape[1], coef.shape[2]
    mode, p, seed, mask, soff = drop if drop is not None else (DROP_NONE, 0.0, 0, None, None)
    if mode not in (DROP_NONE, DROP_RNG_ELEM) or mask is not None:
        raise FiError("the fused forward draws its dropout masks on the device (RNG element mode only)")
    if group0 and (mode != DROP_NONE or not 0 <= group0 < G):
        raise FiError("in_xform: a group offset goes with no dropout and an existing group")
    t = FiInXform(coef[0, group0].data_ptr(), coef[1, group0].data_ptr(), float(slope), int(pool), mode, float(p), seed & 0xFFFFFFFFFFFFFFFF,
                  seed_group_stride & 0xFFFFFFFFFFFFFFFF, None if soff is None else soff.data_ptr())
    t._keep = (coef, soff)
    return t


def conv2d_fwd_fused(x0, t0, x1, t1, w, bias, y, stats, *, ksize, groups, cout=None, shared0=False, tag="conv_fwd"):
    """fi_conv2d_fwd_fused: x* dense NHWC (source 0 at twice the resolution when t0.pool; holding ONE group's images when
    shared0); stats fp64 [G][SLOTS][Cout][2] (or None); y None = statistics-only launch."""
    _dev(x0)
    N, H, W, c0 = x0.shape
    if shared0:
        N *= groups
    if t0 is not None and t0.pool:
        H, W = H // 2, W // 2
    c1 = 0 if x1 is None else x1.shape[3]
    co = int(cout) if y is None else y.shape[3]
    d = FiConv(dt(x0.dtype), N, H, W, ksize, c0, c1, co, 0, 0, 0, 0)
    _attach_w16(d, w)
    cin, px = c0 + c1, N * H * W
    gi = N // groups
    with _timed(tag, (str(x0.dtype)[6:], N, H, W, cin, co, ksize, "fused"), 2.0 * px * cin * co * ksize * ksize,
                x0.numel() * _esz(x0) + (0 if x1 is None else x1.numel() * _esz(x1)) + (0 if y is None else px * co * _esz(y))
                + cin * co * ksize * ksize * _esz(x0)):
        _chk(lib().fi_conv2d_fwd_fused(C.byref(d), None if t0 is None else C.byref(t0), None if t1 is None else C.byref(t1),
                                       int(gi), int(bool(shared0)), ptr(x0), ptr(x1), ptr(w), ptr(bias), ptr(y), ptr(stats),
                                       C.c_long(0 if stats is None else stats.numel() // groups), stream()),
             "fi_conv2d_fwd_fused")


_xcorr_ws = {}


def conv2d_stats_xcorr(x0, t0, w, bias, stats, *, groups, cout, workspace=None, tag="conv_stats_xcorr"):
    """fi_conv2d_stats_xcorr: the statistics a statistics-only fi_conv2d_fwd_fused launch would add, from the input's
    autocorrelation (csrc/xcorr.hip).  x0 dense NHWC [N][H][W][64] 16-bit (raw + t0, or the activation itself); w the 3x3
    forward operand; stats fp64 [G][SLOTS][cout][2] zeroed.  Returns False when the shape is not covered (nothing launched).
    The workspace (~190 MB for 84 x 128^2 -> 512) is cached per (device, shape): stream-ordered reuse, like the arena."""
    _dev(x0)
    N, H, W, c0 = x0.shape
    d = FiConv(dt(x0.dtype), N, H, W, 3, c0, 0, int(cout), 0, 0, 0, 0)
    gi = N // groups
    need = lib().fi_conv2d_stats_xcorr_workspace(C.byref(d), int(gi))
    if need == FI_ERR_UNSUPPORTED:
        return False
    if need < 0:
        _chk(int(need), "fi_conv2d_stats_xcorr_workspace")
    if workspace is None:
        # per stream: two clients hosted by one process run their LC forwards on different streams at the same time
        key = (x0.device.index, torch.cuda.current_stream().cuda_stream, N, H, W, int(cout), gi, x0.dtype)
        workspace = _xcorr_ws.get(key)
        if workspace is None or workspace.numel() < need:
            workspace = _xcorr_ws[key] = torch.empty(need, dtype=torch.uint8, device=x0.device)
    px = N * H * W
    with _timed(tag, (str(x0.dtype)[6:], N, H, W, c0, int(cout), 3, "xcorr"), 2.0 * px * c0 * cout * 9,
                x0.numel() * _esz(x0) + c0 * cout * 9 * _esz(x0), executed_flops=2.0 * px * 13 * c0 * c0):
        rc = lib().fi_conv2d_stats_xcorr(C.byref(d), None if t0 is None else C.byref(t0), int(gi), ptr(x0), ptr(w), ptr(bias),
                                         ptr(stats), C.c_long(stats.numel() // groups), ptr(workspace),
                                         C.c_long(workspace.numel()), stream())
    if rc == FI_ERR_UNSUPPORTED:
        return False
    _chk(rc, "fi_conv2d_stats_xcorr")
    return True


def bn_act_pool_groups(y, coef, slope, z, groups, pool=True):
    """fi_bn_act_pool_groups: y raw [N][2H][2W][C] (pool) or [N][H][W][C], coef fp32 [2][G][C] -> z [N][H][W][C] =
    [maxpool2](act(BN(y))) with group g's coefficient rows for its images."""
    N, Hi, Wi, Cc = _dev(y).shape
    Ho, Wo = (Hi // 2, Wi // 2) if pool else (Hi, Wi)
    with _timed("bn_act_pool" if pool else "bn_act_groups", (str(y.dtype)[6:], N, Hi, Wi, Cc), 0.0, y.numel() * _esz(y) + z.numel() * _esz(z)):
        _chk(lib().fi_bn_act_pool_groups(dt(y.dtype), ptr(y), ptr(coef[0]), ptr(coef[1]), C.c_float(slope), ptr(z), N, Ho, Wo, Cc,
                                         N // groups, int(bool(pool)), stream()), "fi_bn_act_pool_groups")


class FiBnRunItem(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("stats_group_stride", C.c_long), ("count", C.c_double), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p), ("momentum", C.c_float), ("groups", C.c_int),
                ("C", C.c_int)]


def bn_running_groups_multi(items):
    """items: [(stats, groups, count, rmean, rvar, nbt, momentum, shared)] -- the running-statistics half of
    bn_finalize_groups for every listed BatchNorm in one launch (per 32 layers)."""
    if not items:
        return
    arr = (FiBnRunItem * len(items))()
    for k, (stats, groups, count, rmean, rvar, nbt, momentum, shared) in enumerate(items):
        _dev(stats)
        arr[k] = FiBnRunItem(stats.data_ptr(), 0 if shared else stats.numel() // groups, float(count), rmean.data_ptr(),
                             rvar.data_ptr(), 0 if nbt is None else nbt.data_ptr(), float(momentum), int(groups), rmean.numel())
    _chk(lib().fi_bn_running_groups_multi(arr, len(items), stream()), "fi_bn_running_groups_multi")


def bn_finalize_groups(stats, groups, count, gamma, beta, rmean, rvar, nbt, momentum, eps, coef, shared=False):
    """stats fp64 [G][SLOTS][C][2] (shared: ONE accumulator set every group reads) -> coef fp32 [2][G][C]; running
    statistics moved G times in group order."""
    _chk(lib().fi_bn_finalize_groups(ptr(_dev(stats)), C.c_long(0 if shared else stats.numel() // groups), int(groups),
                                     C.c_double(count),
                                     ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), ptr(nbt), C.c_float(momentum),
                                     C.c_float(eps), ptr(coef), gamma.numel(), stream()), "fi_bn_finalize_groups")


def conv2d_wgrad(x0, x1, dy, dw, dbias, *, ksize, deterministic=True):
    _dev(x0)
    N, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    d = FiConv(dt(x0.dtype), N, H, W, ksize, c0, c1, dy.shape[3], 0, 0, 0, 0)
    cin, cout, px = c0 + c1, dy.shape[3], N * H * W
    ws = None
    nbytes = 0
    if deterministic:
        nbytes = lib().fi_conv2d_wgrad_workspace(C.byref(d))
        if nbytes < 0:
            _chk(int(nbytes), "fi_conv2d_wgrad_workspace")
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x0.device)     # caller-owned workspace
    with _timed("conv_wgrad", (str(x0.dtype)[6:], N, H, W, cin, cout, ksize), 2.0 * px * cin * cout * ksize * ksize,
                px * cin * _esz(x0) + px * cout * _esz(dy) + cin * cout * ksize * ksize * 4):
        _chk(lib().fi_conv2d_wgrad(C.byref(d), ptr(x0), ptr(x1), ptr(dy), ptr(dw), ptr(dbias), ptr(ws),
                                   C.c_long(nbytes), stream()), "fi_conv2d_wgrad")


def conv2d_wgrad_partial(x0, x1, dy, want_bias, *, ksize):
    """Stage 1 only.  Returns (workspace tensor, slices, stride) for a later fi_wgrad_reduce_multi."""
    _dev(x0)
    N, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    d = FiConv(dt(x0.dtype), N, H, W, ksize, c0, c1, dy.shape[3], 0, 0, 0, 0)
    cin, cout, px = c0 + c1, dy.shape[3], N * H * W
    nbytes = lib().fi_conv2d_wgrad_workspace(C.byref(d))
    if nbytes < 0:
        _chk(int(nbytes), "fi_conv2d_wgrad_workspace")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x0.device)
    slices, stride = C.c_int(0), C.c_long(0)
    with _timed("conv_wgrad", (str(x0.dtype)[6:], N, H, W, cin, cout, ksize), 2.0 * px * cin * cout * ksize * ksize,
                px * cin * _esz(x0) + px * cout * _esz(dy) + cin * cout * ksize * ksize * 4):
        _chk(lib().fi_conv2d_wgrad_partial(C.byref(d), ptr(x0), ptr(x1), ptr(dy), int(want_bias), ptr(ws),
                                           C.c_long(nbytes), C.byref(slices), C.byref(stride), stream()),
             "fi_conv2d_wgrad_partial")
    return ws, slices.value, stride.value


WGRAD_ROW = 11       # FI_WGRAD_ROW
CE_SLOTS = 16        # FI_CE_SLOTS


def wgrad_reduce_multi(table, n, nblocks, nblocks3d=0):
    """nblocks3d > 0: rows with word 9 < 0 get their second launch (fi_wgrad_permute3d_multi) right behind the reduce."""
    with _timed("wgrad_reduce", (n,), 0, 0):
        _chk(lib().fi_wgrad_reduce_multi(ptr(_dev(table)), int(n), int(nblocks), stream()), "fi_wgrad_reduce_multi")
        if nblocks3d > 0:
            _chk(lib().fi_wgrad_permute3d_multi(ptr(_dev(table)), int(n), int(nblocks3d), stream()), "fi_wgrad_permute3d_multi")


def pack_weights3d_multi(table, ntensors, nblocks, dtype):
    """fi_pack_weights3d_multi: table int64 [ntensors][6] on the device (ops3d._MultiPack3D builds it)."""
    _chk(lib().fi_pack_weights3d_multi(ptr(_dev(table)), int(ntensors), int(nblocks), dt(dtype), stream()), "fi_pack_weights3d_multi")


def pack_weights(src, dst, cout, kk, cin, mode):
    _chk(lib().fi_pack_weights(ptr(_dev(src)), ptr(dst), cout, kk, cin, mode, dt(dst.dtype), stream()),
         "fi_pack_weights")


def pack_weights_multi(table, ntensors, dtype):
    _chk(lib().fi_pack_weights_multi(ptr(_dev(table)), int(ntensors), dt(dtype), stream()), "fi_pack_weights_multi")


def bn_fused_fwd(y, z, stats, gamma, beta, rmean, rvar, nbt, momentum, eps, training, coef, slope, drop=None):
    d = _bnact(_dev(y), slope, drop)
    with _timed("bn_act_fwd", (str(y.dtype)[6:],) + tuple(y.shape), 0, 2 * y.numel() * _esz(y)):
        _chk(lib().fi_bn_fused_fwd(C.byref(d), ptr(y), ptr(z), ptr(stats), ptr(gamma), ptr(beta), ptr(rmean),
                                   ptr(rvar), ptr(nbt), C.c_float(momentum), C.c_float(eps), int(training),
                                   ptr(coef), stream()), "fi_bn_fused_fwd")


def bn_finalize(stats, count, gamma, beta, rmean, rvar, nbt, momentum, eps, training, scale, shift, mean, invstd):
    _chk(lib().fi_bn_finalize(ptr(stats), C.c_double(count), ptr(_dev(gamma)), ptr(beta), ptr(rmean), ptr(rvar),
                              ptr(nbt), C.c_float(momentum), C.c_float(eps), int(training), ptr(scale), ptr(shift),
                              ptr(mean), ptr(invstd), gamma.numel(), stream()), "fi_bn_finalize")


def _bnact(y, slope, drop):
    N, H, W, Cc = y.shape
    mode, p, seed, mask, soff = drop if drop is not None else (DROP_NONE, 0.0, 0, None, None)
    return FiBnAct(dt(y.dtype), N * H * W, Cc, H * W, slope, mode, p, seed & 0xFFFFFFFFFFFFFFFF,
                   None if mask is None else mask.data_ptr(), None if soff is None else soff.data_ptr())


def bn_act_fwd(y, scale, shift, z, slope, drop=None):
    d = _bnact(_dev(y), slope, drop)
    with _timed("bn_act_fwd", (str(y.dtype)[6:],) + tuple(y.shape), 0, 2 * y.numel() * _esz(y)):
        _chk(lib().fi_bn_act_fwd(C.byref(d), ptr(y), ptr(scale), ptr(shift), ptr(z), stream()), "fi_bn_act_fwd")


def bn_act_bwd_reduce(dz, y, scale, shift, mean, invstd, sums, slope, drop=None):
    d = _bnact(_dev(y), slope, drop)
    with _timed("bn_act_bwd_reduce", (str(y.dtype)[6:],) + tuple(y.shape), 0, 2 * y.numel() * _esz(y)):
        _chk(lib().fi_bn_act_bwd_reduce(C.byref(d), ptr(dz), ptr(y), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                                        ptr(sums), stream()), "fi_bn_act_bwd_reduce")


def bn_act_bwd_apply(dz, y, scale, shift, mean, invstd, sums, training, dy, dgamma, dbeta, slope, drop=None,
                     accumulate_param=False):
    d = _bnact(_dev(y), slope, drop)
    with _timed("bn_act_bwd_apply", (str(y.dtype)[6:],) + tuple(y.shape), 0, 3 * y.numel() * _esz(y)):
        _chk(lib().fi_bn_act_bwd_apply(C.byref(d), ptr(dz), ptr(y), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                                       ptr(sums), int(training), ptr(dy), ptr(dgamma), ptr(dbeta),
                                       int(accumulate_param), stream()), "fi_bn_act_bwd_apply")


def instnorm_fwd_batched(y, z, stats, ones, zeros, rm, rv, eps, coef):
    """InstanceNorm3d(affine=False) + ReLU of all samples in one launch: y / z [N, D, H, W, C] dense, stats fp64 [N, slots*C*2],
    coef fp32 [N, 4, C] (scale, shift, mean, invstd rows per sample)."""
    d = _bnact(_dev(y)[0], 0.0, None)
    N = y.shape[0]
    with _timed("bn_act_fwd", (str(y.dtype)[6:],) + tuple(y.shape), 0, 2 * y.numel() * _esz(y)):
        _chk(lib().fi_bn_fused_fwd_batched(C.byref(d), N, C.c_long(y[0].numel()), C.c_long(stats.stride(0)), C.c_long(coef.stride(0)),
                                           ptr(y), ptr(z), ptr(stats), ptr(ones), ptr(zeros), ptr(rm), ptr(rv), C.c_float(eps),
                                           ptr(coef), stream()), "fi_bn_fused_fwd_batched")


def instnorm_bwd_batched(dz, y, coef, sums, dy):
    """Both backward passes (reduce, apply) of the batched InstanceNorm3d + ReLU: sums fp64 [N, slots*C*2] zeroed."""
    d = _bnact(_dev(y)[0], 0.0, None)
    N = y.shape[0]
    args = (C.byref(d), N, C.c_long(y[0].numel()), C.c_long(sums.stride(0)), C.c_long(coef.stride(0)), ptr(dz), ptr(y),
            ptr(coef[0, 0]), ptr(coef[0, 1]), ptr(coef[0, 2]), ptr(coef[0, 3]), ptr(sums))
    with _timed("bn_act_bwd_reduce", (str(y.dtype)[6:],) + tuple(y.shape), 0, 2 * y.numel() * _esz(y)):
        _chk(lib().fi_bn_act_bwd_reduce_batched(*args, stream()), "fi_bn_act_bwd_reduce_batched")
    with _timed("bn_act_bwd_apply", (str(y.dtype)[6:],) + tuple(y.shape), 0, 3 * y.numel() * _esz(y)):
        _chk(lib().fi_bn_act_bwd_apply_batched(*args, 1, ptr(dy), stream()), "fi_bn_act_bwd_apply_batched")


def maxpool2_fwd(x, y):
    N, H, W, Cc = _dev(x).shape
    with _timed("maxpool_fwd", (str(x.dtype)[6:],) + tuple(x.shape), 0, 1.25 * x.numel() * _esz(x)):
        _chk(lib().fi_maxpool2_fwd(dt(x.dtype), ptr(x), ptr(y), N, H, W, Cc, stream()), "fi_maxpool2_fwd")


def maxpool2_bwd(x, dy, dx, accumulate=False):
    N, H, W, Cc = _dev(x).shape
    with _timed("maxpool_bwd", (str(x.dtype)[6:],) + tuple(x.shape), 0, 2.25 * x.numel() * _esz(x)):
        _chk(lib().fi_maxpool2_bwd(dt(x.dtype), ptr(x), ptr(dy), ptr(dx), N, H, W, Cc, int(accumulate), stream()),
             "fi_maxpool2_bwd")


def maxpool2_bwd_add(x, dy, add, dx):
    """dx = add + scatter(dy): the pooling gradient joins the gradient of the tensor's other consumer in one pass."""
    N, H, W, Cc = _dev(x).shape
    with _timed("maxpool_bwd", (str(x.dtype)[6:],) + tuple(x.shape), 0, 3.25 * x.numel() * _esz(x)):
        _chk(lib().fi_maxpool2_bwd_add(dt(x.dtype), ptr(x), ptr(dy), ptr(add), ptr(dx), N, H, W, Cc, stream()),
             "fi_maxpool2_bwd_add")


def upsample2x_fwd(x, y):
    N, h, w, Cc = _dev(x).shape
    with _timed("upsample_fwd", (str(x.dtype)[6:],) + tuple(x.shape), 0, 5 * x.numel() * _esz(x)):
        _chk(lib().fi_upsample2x_fwd(dt(x.dtype), ptr(x), ptr(y), N, h, w, Cc, stream()), "fi_upsample2x_fwd")


def conv1x1_up2x_fwd(x, t0, w, bias, y, *, groups=1):
    """fi_conv1x1_up2x_fwd: y [N,2h,2w,cout] = bilinear x2 of conv1x1(act(BN(x)) or x).  Returns False (nothing launched) when the
    shape is not covered: the caller makes the two launches."""
    _dev(x)
    N, h, wd, cin = x.shape
    cout = y.shape[3]
    px = N * h * wd
    with _timed("conv_up_fwd", (str(x.dtype)[6:], N, h, wd, cin, cout, 1) + (("fused",) if t0 is not None else ()),
                2.0 * px * cin * cout, x.numel() * _esz(x) + y.numel() * _esz(y) + cin * cout * _esz(x)):
        rc = lib().fi_conv1x1_up2x_fwd(dt(x.dtype), N, h, wd, cin, cout, None if t0 is None else C.byref(t0), int(N // groups),
                                       ptr(x), ptr(w), ptr(bias), ptr(y), stream())
    if rc == FI_ERR_UNSUPPORTED:
        return False
    _chk(rc, "fi_conv1x1_up2x_fwd")
    return True


def upfuse_tuning(rows):
    _chk(lib().fi_upfuse_tuning(int(rows)), "fi_upfuse_tuning")


def upsample2x_bwd(dy, dx, accumulate=False):
    N, h, w, Cc = _dev(dx).shape
    with _timed("upsample_bwd", (str(dx.dtype)[6:],) + tuple(dx.shape), 0, 5 * dx.numel() * _esz(dx)):
        _chk(lib().fi_upsample2x_bwd(dt(dx.dtype), ptr(dy), ptr(dx), N, h, w, Cc, int(accumulate), stream()),
             "fi_upsample2x_bwd")


def maxpool3d_fwd(x, y):
    N, D, H, W, Cc = _dev(x).shape
    _chk(lib().fi_maxpool3d_fwd(dt(x.dtype), ptr(x), ptr(y), N, D, H, W, Cc, stream()), "fi_maxpool3d_fwd")


def maxpool3d_bwd(x, dy, dx):
    N, D, H, W, Cc = _dev(x).shape
    _chk(lib().fi_maxpool3d_bwd(dt(x.dtype), ptr(x), ptr(dy), ptr(dx), N, D, H, W, Cc, stream()), "fi_maxpool3d_bwd")


def maxpool3d_bwd_add(x, dy, add, dx):
    N, D, H, W, Cc = _dev(x).shape
    _chk(lib().fi_maxpool3d_bwd_add(dt(x.dtype), ptr(x), ptr(dy), ptr(add), ptr(dx), N, D, H, W, Cc, stream()), "fi_maxpool3d_bwd_add")


def upsample3d2x_fwd(x, y):
    N, d, h, w, Cc = _dev(x).shape
    _chk(lib().fi_upsample3d2x_fwd(dt(x.dtype), ptr(x), ptr(y), N, d, h, w, Cc, stream()), "fi_upsample3d2x_fwd")


def upsample3d2x_bwd(dy, dx):
    N, d, h, w, Cc = _dev(dx).shape
    _chk(lib().fi_upsample3d2x_bwd(dt(dx.dtype), ptr(dy), ptr(dx), N, d, h, w, Cc, stream()), "fi_upsample3d2x_bwd")


def ce_fwd(logits, labels, ignore_index, acc):
    M, Cc = _dev(logits).numel() // logits.shape[-1], logits.shape[-1]
    _chk(lib().fi_ce_fwd(ptr(logits), ptr(labels), C.c_long(M), Cc, ignore_index, ptr(acc), stream()), "fi_ce_fwd")


def ce_finalize(acc, loss):
    _chk(lib().fi_ce_finalize(ptr(_dev(acc)), ptr(loss), stream()), "fi_ce_finalize")


def ce_bwd(logits, labels, ignore_index, acc, gscale, dlogits):
    M, Cc = _dev(logits).numel() // logits.shape[-1], logits.shape[-1]
    _chk(lib().fi_ce_bwd(ptr(logits), ptr(labels), C.c_long(M), Cc, ignore_index, ptr(acc), ptr(gscale), ptr(dlogits),
                         dt(dlogits.dtype), stream()), "fi_ce_bwd")


def pdice_fwd(probs, labels, ignore_index, acc):
    B, Cc = _dev(probs).shape[0], probs.shape[-1]
    HW = probs.numel() // (B * Cc)
    _chk(lib().fi_pdice_fwd(ptr(probs), ptr(labels), B, C.c_long(HW), Cc, int(ignore_index), ptr(acc), stream()),
         "fi_pdice_fwd")


def pdice_finalize(acc, weight, ncls, loss):
    _chk(lib().fi_pdice_finalize(ptr(_dev(acc)), ptr(weight), int(ncls), ptr(loss), stream()), "fi_pdice_finalize")


def pdice_bwd(probs, labels, ignore_index, acc, weight, gscale, dprobs):
    B, Cc = _dev(probs).shape[0], probs.shape[-1]
    HW = probs.numel() // (B * Cc)
    _chk(lib().fi_pdice_bwd(ptr(probs), ptr(labels), B, C.c_long(HW), Cc, int(ignore_index), ptr(acc), ptr(weight),
                            ptr(gscale), ptr(dprobs), stream()), "fi_pdice_bwd")


def dice_counts(logits, gt, counts):
    M, Cc = _dev(logits).numel() // logits.shape[-1], logits.shape[-1]
    _chk(lib().fi_dice_counts(ptr(logits), ptr(gt), C.c_long(M), Cc, ptr(counts), stream()), "fi_dice_counts")


def tree_grid_weights(fm, weight):
    B, Cc, H, W = _dev(fm).shape
    _chk(lib().fi_tree_grid_weights(ptr(fm), B, Cc, H, W, ptr(weight), stream()), "fi_tree_grid_weights")


def tree_mst(weight, H, W, edges):
    B = _dev(weight).shape[0]
    per = lib().fi_tree_mst_workspace(H, W)
    ws = torch.empty(per * B, dtype=torch.uint8, device=weight.device)
    _chk(lib().fi_tree_mst(ptr(weight), B, H, W, ptr(edges), ptr(ws), C.c_long(per * B), stream()), "fi_tree_mst")


def tree_bfs(edges, H, W, sidx, spar, schild, levels):
    B = _dev(edges).shape[0]
    adj = torch.empty((B, H * W, 4), dtype=torch.int32, device=edges.device)
    _chk(lib().fi_tree_bfs(ptr(edges), B, H, W, ptr(sidx), ptr(spar), ptr(schild), ptr(levels), ptr(adj), stream()),
         "fi_tree_bfs")


def tree_edge_weights(embed, sidx, spar, inv_sigma, w):
    B, Ce, V = _dev(embed).shape
    _chk(lib().fi_tree_edge_weights(ptr(embed), ptr(sidx), ptr(spar), B, Ce, V, C.c_float(inv_sigma), ptr(w), stream()),
         "fi_tree_edge_weights")


def tree_edge_weights_bwd(embed, sidx, spar, schild, w, gw, inv_sigma, gembed):
    B, Ce, V = _dev(embed).shape
    _chk(lib().fi_tree_edge_weights_bwd(ptr(embed), ptr(sidx), ptr(spar), ptr(schild), ptr(w), ptr(gw), B, Ce, V,
                                        C.c_float(inv_sigma), ptr(gembed), stream()), "fi_tree_edge_weights_bwd")


def tree_aggr_up(x, w, sidx, schild, levels, out):
    B, Cc, V = _dev(out).shape
    _chk(lib().fi_tree_aggr_up(ptr(x), ptr(w), ptr(sidx), ptr(schild), ptr(levels), B, Cc, V, ptr(out), stream()),
         "fi_tree_aggr_up")


def tree_prop_down(xs, w, sidx, spar, levels, out):
    B, Cc, V = _dev(out).shape
    _chk(lib().fi_tree_prop_down(ptr(xs), ptr(w), ptr(sidx), ptr(spar), ptr(levels), B, Cc, V, ptr(out), stream()),
         "fi_tree_prop_down")


def tree_grad_rec(in_data, in_grad, out_data, w, sidx, spar, levels, grad):
    B, Cd, V = _dev(in_data).shape
    Cg = in_grad.shape[1]
    _chk(lib().fi_tree_grad_rec(ptr(in_data), ptr(in_grad), ptr(out_data), ptr(w), ptr(sidx), ptr(spar), ptr(levels), B, Cd,
                                Cg, V, ptr(grad), stream()), "fi_tree_grad_rec")


TREE_MAPS, TREE_TERMS = 4, 3       # FI_TREE_MAPS, FI_TREE_TERMS


class FiTreeMap(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("stride", C.c_long * 4), ("C", C.c_int), ("h", C.c_int), ("w", C.c_int)]


def _tree_maps(pairs):
    """pairs: [(a, b, C, h, w)]: forward (source in any layout, dense resized map), backward (grad of the resized map, grad of
    the source; h, w are the SOURCE's; both dense)."""
    arr = (FiTreeMap * max(len(pairs), 1))()
    for k, (a, b, Ck, h, w) in enumerate(pairs):
        arr[k] = FiTreeMap(_dev(a).data_ptr(), _dev(b).data_ptr(), (C.c_long * 4)(*[int(v) for v in a.stride()]), int(Ck), int(h), int(w))
    return arr


def tree_prep_fwd(preds, prob, maps, roi_src, rois, count, N, H, W):
    """preds / map sources: fp32 [N,C,h,w] in ANY layout (strides are passed); prob / resized maps dense NCHW; roi_src uint8
    [N,h,w] dense or None."""
    pairs = [(s_, d_, s_.shape[1], s_.shape[2], s_.shape[3]) for s_, d_ in maps]
    Cc = 0 if preds is None else preds.shape[1]
    pst = (C.c_long * 4)(*([0] * 4 if preds is None else [int(v) for v in preds.stride()]))
    _chk(lib().fi_tree_prep_fwd(ptr(preds), pst, ptr(prob), int(N), int(Cc), int(H), int(W), _tree_maps(pairs), len(pairs), ptr(roi_src),
                                0 if roi_src is None else roi_src.shape[1], 0 if roi_src is None else roi_src.shape[2],
                                ptr(rois), ptr(count), stream()), "fi_tree_prep_fwd")


def tree_prep_bwd(prob, dprob, dpreds, maps, N, H, W):
    """maps: [(grad w.r.t. the resized map [N,C,H,W], grad w.r.t. the source [N,C,h,w] -- written)]."""
    pairs = [(g_, o_, o_.shape[1], o_.shape[2], o_.shape[3]) for g_, o_ in maps]
    Cc = 0 if dpreds is None else dpreds.shape[1]
    _chk(lib().fi_tree_prep_bwd(ptr(prob), ptr(dprob), ptr(dpreds), int(N), int(Cc), int(H), int(W), _tree_maps(pairs), len(pairs),
                                stream()), "fi_tree_prep_bwd")


def _ptr_array(ts):
    return (C.c_void_p * max(len(ts), 1))(*[0 if t is None else t.data_ptr() for t in ts])


def tree_masked_l1_fwd(prob, as_list, rois, count, weight, acc, loss):
    N, Cc, H, W = _dev(prob).shape
    _chk(lib().fi_tree_masked_l1_fwd(ptr(prob), _ptr_array(as_list), len(as_list), ptr(rois), N, Cc, H, W, ptr(count),
                                     C.c_float(weight), ptr(acc), ptr(loss), stream()), "fi_tree_masked_l1_fwd")


def tree_masked_l1_bwd(prob, as_list, rois, count, weight, gout, dprob, das):
    N, Cc, H, W = _dev(prob).shape
    _chk(lib().fi_tree_masked_l1_bwd(ptr(prob), _ptr_array(as_list), len(as_list), ptr(rois), N, Cc, H, W, ptr(count),
                                     C.c_float(weight), ptr(gout), ptr(dprob), _ptr_array(das), stream()), "fi_tree_masked_l1_bwd")


def tv_loss_fwd(p, eroded, idx_e, idx_d, positive, acc):
    planes = _dev(p).numel() // (p.shape[-2] * p.shape[-1])
    _chk(lib().fi_tv_loss_fwd(ptr(p), C.c_long(planes), p.shape[-2], p.shape[-1], ptr(eroded), ptr(idx_e), ptr(idx_d), ptr(positive),
                              ptr(acc), stream()), "fi_tv_loss_fwd")


def tv_loss_bwd(idx_e, idx_d, positive, gout, shape, scratch, dp):
    planes = dp.numel() // (shape[-2] * shape[-1])
    _chk(lib().fi_tv_loss_bwd(ptr(_dev(idx_e)), ptr(idx_d), ptr(positive), ptr(gout), C.c_long(planes), shape[-2], shape[-1],
                              ptr(scratch), ptr(dp), stream()), "fi_tv_loss_bwd")


CRF_SLOTS = 16       # FI_CRF_SLOTS


def gatedcrf_fwd(y_nhwc, feat_nhwc, radius, weights, sigma_xy, sigma_sample, prod, acc):
    """weights / sigma_xy / sigma_sample: python sequences, one entry per kernel (sigma <= 0: modality not used)."""
    N, H, W, Cc = _dev(y_nhwc).shape
    F_ = feat_nhwc.shape[3]
    nk = len(weights)
    arr = lambda v: (C.c_float * nk)(*[float(t) for t in v])
    _chk(lib().fi_gatedcrf_fwd(ptr(y_nhwc), ptr(_dev(feat_nhwc)), N, H, W, Cc, F_, int(radius), nk, arr(weights),
                               arr(sigma_xy), arr(sigma_sample), ptr(prod), ptr(acc), stream()), "fi_gatedcrf_fwd")


def augment2d(src_img, src_lab, ip, dp, out_img, out_lab, img_cval, lab_cval):
    n, Cc, H, W = _dev(src_img).shape
    B = ip.shape[0]
    assert src_img.dtype == torch.float32 and src_lab.dtype == torch.uint8 and ip.dtype == torch.int32
    assert dp.dtype == torch.float64 and tuple(out_img.shape) == (B, Cc, H, W) and tuple(out_lab.shape) == (B, H, W)
    _chk(lib().fi_augment2d(ptr(src_img), ptr(src_lab), ptr(_dev(ip)), ptr(_dev(dp)), ptr(out_img), ptr(out_lab), B, Cc, H, W,
                            C.c_float(img_cval), int(lab_cval), stream()), "fi_augment2d")


def seg_borders(logits_hwc, gt_u8, k, pred_list, gt_list, counts):
    H, W, Cc = _dev(logits_hwc).shape
    _chk(lib().fi_seg_borders(ptr(logits_hwc), ptr(gt_u8), H, W, Cc, int(k), ptr(pred_list), ptr(gt_list), ptr(counts),
                              stream()), "fi_seg_borders")


def surface_distances(from_list, to_list, counts, from_index, to_index, W, out):
    _chk(lib().fi_surface_distances(ptr(_dev(from_list)), ptr(to_list), ptr(counts), int(from_index), int(to_index),
                                    int(W), int(from_list.numel()), ptr(out), stream()), "fi_surface_distances")


def adamw_hyper(step, hyper, lr_state, beta1, beta2, wd):
    _chk(lib().fi_adamw_hyper(ptr(_dev(step)), ptr(hyper), ptr(lr_state), C.c_float(beta1), C.c_float(beta2),
                              C.c_float(wd), stream()), "fi_adamw_hyper")


def lr_poly_advance(it, lr_state, base_lr, max_iter):
    _chk(lib().fi_lr_poly_advance(ptr(_dev(it)), ptr(lr_state), C.c_double(base_lr), C.c_double(max_iter), stream()),
         "fi_lr_poly_advance")


def adamw_step(p, g, m, v, hyper, beta1, beta2, eps, shadow=None):
    with _timed("adamw_step", (p.numel(),), 0, 28 * p.numel()):
        _chk(lib().fi_adamw_step(ptr(_dev(p)), ptr(g), ptr(m), ptr(v), C.c_long(p.numel()), ptr(hyper),
                                 C.c_float(beta1), C.c_float(beta2), C.c_float(eps), ptr(shadow), stream()),
             "fi_adamw_step")


def sgd_step(p, g, buf, lr_state, momentum, wd, skip_hyper=None):
    _chk(lib().fi_sgd_step(ptr(_dev(p)), ptr(g), ptr(buf), C.c_long(p.numel()), ptr(lr_state), C.c_float(momentum),
                           C.c_float(wd), ptr(skip_hyper), stream()), "fi_sgd_step")


def amp_unscale(grads, scale_t, found_inf):
    _chk(lib().fi_amp_unscale(ptr(_dev(grads)), C.c_long(grads.numel()), ptr(scale_t), ptr(found_inf), stream()),
         "fi_amp_unscale")


def amp_guard(step, hyper, found_inf):
    _chk(lib().fi_amp_guard(ptr(_dev(step)), ptr(hyper), ptr(found_inf), stream()), "fi_amp_guard")


def amp_update(scale_t, tracker, found_inf, growth, backoff, interval):
    _chk(lib().fi_amp_update(ptr(_dev(scale_t)), ptr(tracker), ptr(found_inf), C.c_float(growth), C.c_float(backoff),
                             int(interval), stream()), "fi_amp_update")


def scale(x, y, a, divide=False):
    _chk(lib().fi_scale(ptr(_dev(x)), ptr(y), C.c_long(x.numel()), C.c_float(a), int(divide), stream()), "fi_scale")


def axpy(acc, x, a):
    _chk(lib().fi_axpy(ptr(_dev(acc)), ptr(x), C.c_long(x.numel()), C.c_float(a), stream()), "fi_axpy")


def _taps_array(taps):
    arr = (C.c_void_p * 3)()
    for i, t in enumerate(taps):
        arr[i] = t.data_ptr()
    return arr


def conv3d_fwd(x0, x1, w_taps, bias, y, stats, *, ksize, y_f32=False):
    """x*: [N,D,H,W,C] dense; w_taps: packed 2D operands per depth tap; y zeroed by the caller; stats [N, slots*C*2] or None."""
    N, D, H, W, c0 = _dev(x0).shape
    c1 = 0 if x1 is None else x1.shape[4]
    d = FiConv(dt(x0.dtype), N, H, W, ksize, c0, c1, y.shape[4], 0, 1, 0, int(y_f32))
    stride = 0 if stats is None else stats.stride(0)
    cin, co, vox = c0 + c1, y.shape[4], N * D * H * W
    with _timed("conv3d_fwd", (str(x0.dtype)[6:], N, D, H, W, cin, co, ksize, "taps"), 2.0 * vox * cin * co * ksize ** 3,
                vox * cin * _esz(x0) + vox * co * _esz(y) + cin * co * ksize ** 3 * _esz(x0)):
        _chk(lib().fi_conv3d_fwd(C.byref(d), D, ptr(x0), ptr(x1), _taps_array(w_taps), ptr(bias), ptr(y), ptr(stats),
                                 C.c_long(stride), stream()), "fi_conv3d_fwd")


def conv3d_first_fwd(x, w27, bias, y, stats):
    """Conv3d(1 -> 16, 3^3, pad 1): x [N,D,H,W,1] 16-bit, w27 fp32 [16,27] (kd, kh, kw), y [N,D,H,W,16]; stats [N, slots*16*2] / None."""
    N, D, H, W, _ = _dev(x).shape
    vox = N * D * H * W
    with _timed("conv3d_fwd", (str(x.dtype)[6:], N, D, H, W, 1, 16, 3, "first"), 2.0 * vox * 16 * 27, vox * 17 * _esz(x) + 16 * 27 * _esz(x)):
        _chk(lib().fi_conv3d_first_fwd(dt(x.dtype), N, D, H, W, ptr(x), ptr(_dev(w27)), ptr(bias), ptr(y), ptr(stats),
                                       C.c_long(0 if stats is None else stats.stride(0)), stream()), "fi_conv3d_first_fwd")


def conv3d_first_wgrad(x, dy, dw27, dbias):
    """dw27 fp32 [16,27] and dbias fp32 [16] are added to."""
    N, D, H, W, _ = _dev(x).shape
    vox = N * D * H * W
    nbytes = int(lib().fi_conv3d_first_wgrad_workspace(N, D, H, W))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    with _timed("conv3d_wgrad", (str(x.dtype)[6:], N, D, H, W, 1, 16, 3, "first"), 2.0 * vox * 16 * 27, vox * 17 * _esz(x) + 16 * 27 * _esz(x)):
        _chk(lib().fi_conv3d_first_wgrad(dt(x.dtype), N, D, H, W, ptr(x), ptr(_dev(dy)), ptr(dw27), ptr(dbias), ptr(ws),
                                         C.c_long(nbytes), stream()), "fi_conv3d_first_wgrad")


def conv3d_point_fwd(x, w2, bias, y):
    """Conv3d(16 -> cout <= 4, 1x1x1): x [..., 16] 16-bit dense, w2 fp32 [cout, 16], y fp32 [..., cout]."""
    vox, co = _dev(x).numel() // 16, w2.shape[0]
    with _timed("conv3d_fwd", (str(x.dtype)[6:],) + tuple(x.shape[:4]) + (16, co, 1, "point"), 2.0 * vox * 16 * co,
                vox * 16 * _esz(x) + vox * co * 4):
        _chk(lib().fi_conv3d_point_fwd(dt(x.dtype), C.c_long(vox), co, ptr(x), ptr(_dev(w2)), ptr(bias), ptr(y), stream()),
             "fi_conv3d_point_fwd")


def conv3d_point_dgrad(dy, w2, dx):
    vox, co = _dev(dx).numel() // 16, w2.shape[0]
    with _timed("conv3d_dgrad", (str(dx.dtype)[6:],) + tuple(dx.shape[:4]) + (co, 16, 1, "point"), 2.0 * vox * 16 * co,
                vox * 16 * _esz(dx) + vox * co * 4):
        _chk(lib().fi_conv3d_point_dgrad(dt(dx.dtype), C.c_long(vox), co, ptr(_dev(dy)), ptr(w2), ptr(dx), stream()),
             "fi_conv3d_point_dgrad")


def conv3d_point_wgrad(x, dy, dw, dbias):
    """dw fp32 [cout * 16] / dbias fp32 [cout] are added to (either may be None)."""
    vox = _dev(x).numel() // 16
    co = dy.numel() // vox
    nbytes = int(lib().fi_conv3d_point_wgrad_workspace())
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    with _timed("conv3d_wgrad", (str(x.dtype)[6:],) + tuple(x.shape[:4]) + (16, co, 1, "point"), 2.0 * vox * 16 * co,
                vox * 16 * _esz(x) + vox * co * 4):
        _chk(lib().fi_conv3d_point_wgrad(dt(x.dtype), C.c_long(vox), co, ptr(x), ptr(_dev(dy)), ptr(dw), ptr(dbias), ptr(ws),
                                         C.c_long(nbytes), stream()), "fi_conv3d_point_wgrad")


def conv3d_fwd_fused(x0, x1, w_all, bias, y, stats, *, ksize):
    """One-launch 3x3x3 convolution (w_all [Cout][9][3][cin]); y written.  -> False when the shape is not covered."""
    N, D, H, W, c0 = _dev(x0).shape
    c1 = 0 if x1 is None else x1.shape[4]
    d = FiConv(dt(x0.dtype), N, H, W, ksize, c0, c1, y.shape[4], 0, 0, 0, 0)
    stride = 0 if stats is None else stats.stride(0)
    cin, co, vox = c0 + c1, y.shape[4], N * D * H * W
    with _timed("conv3d_fwd", (str(x0.dtype)[6:], N, D, H, W, cin, co, ksize), 2.0 * vox * cin * co * ksize ** 3,
                vox * (cin + co) * _esz(x0) + cin * co * ksize ** 3 * _esz(x0)):
        rc = lib().fi_conv3d_fwd_fused(C.byref(d), D, ptr(x0), ptr(x1), ptr(w_all), ptr(bias), ptr(y), ptr(stats),
                                       C.c_long(stride), stream())
    if rc == FI_ERR_UNSUPPORTED:
        return False
    _chk(rc, "fi_conv3d_fwd_fused")
    return True


def conv3d_tuning(stream_on=-1):
    """fi_conv3d_tuning: measurement / test hook (1 = the depth-streaming kernel for the thin 128^3 layers, 0 = the general form)."""
    _chk(lib().fi_conv3d_tuning(int(stream_on)), "fi_conv3d_tuning")


def conv3d_dgrad_fused(dy, wt_all, d0, d1, *, ksize):
    N, D, H, W, cout = _dev(dy).shape
    d = FiConv(dt(dy.dtype), N, H, W, ksize, cout, 0, d0.shape[4], 0 if d1 is None else d1.shape[4], 0, 0, 0)
    cin, vox = d0.shape[4] + (0 if d1 is None else d1.shape[4]), N * D * H * W
    with _timed("conv3d_dgrad", (str(dy.dtype)[6:], N, D, H, W, cout, cin, ksize), 2.0 * vox * cin * cout * ksize ** 3,
                vox * (cin + cout) * _esz(dy) + cin * cout * ksize ** 3 * _esz(dy)):
        rc = lib().fi_conv3d_dgrad_fused(C.byref(d), D, ptr(dy), ptr(wt_all), ptr(d0), ptr(d1), stream())
    if rc == FI_ERR_UNSUPPORTED:
        return False
    _chk(rc, "fi_conv3d_dgrad_fused")
    return True


def conv3d_dgrad(dy, wt_taps, d0, d1, *, ksize):
    N, D, H, W, cout = _dev(dy).shape
    d = FiConv(dt(dy.dtype), N, H, W, ksize, cout, 0, d0.shape[4], 0 if d1 is None else d1.shape[4], 1, 1, 0)
    cin, vox = d0.shape[4] + (0 if d1 is None else d1.shape[4]), N * D * H * W
    with _timed("conv3d_dgrad", (str(dy.dtype)[6:], N, D, H, W, cout, cin, ksize, "taps"), 2.0 * vox * cin * cout * ksize ** 3,
                vox * (cin + cout) * _esz(dy) + cin * cout * ksize ** 3 * _esz(dy)):
        _chk(lib().fi_conv3d_dgrad(C.byref(d), D, ptr(dy), _taps_array(wt_taps), ptr(d0), ptr(d1), stream()), "fi_conv3d_dgrad")


def conv3d_wgrad(x0, x1, dy, dw_taps, dbias, *, ksize):
    N, D, H, W, c0 = _dev(x0).shape
    c1 = 0 if x1 is None else x1.shape[4]
    d = FiConv(dt(x0.dtype), N, H, W, ksize, c0, c1, dy.shape[4], 0, 0, 0, 0)
    lib().fi_conv3d_wgrad_workspace.restype = C.c_long
    nbytes = lib().fi_conv3d_wgrad_workspace(C.byref(d), D)
    if nbytes < 0:
        _chk(int(nbytes), "fi_conv3d_wgrad_workspace")
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x0.device)      # caller-owned workspace
    cin, co, vox = c0 + c1, dy.shape[4], N * D * H * W
    with _timed("conv3d_wgrad", (str(x0.dtype)[6:], N, D, H, W, cin, co, ksize), 2.0 * vox * cin * co * ksize ** 3,
                vox * (cin + co) * _esz(x0) + cin * co * ksize ** 3 * 4):
        _chk(lib().fi_conv3d_wgrad(C.byref(d), D, ptr(x0), ptr(x1), ptr(dy), ptr(dw_taps), ptr(dbias), ptr(ws), C.c_long(nbytes),
                                   stream()), "fi_conv3d_wgrad")


def conv3d_wgrad_fused(x0, x1, dy, dw_all, dbias, *, ksize):
    """One-launch 3x3x3 filter gradient: dw_all fp32 [cout][9][3][cin] (added to).  -> False when the shape is not covered."""
    N, D, H, W, c0 = _dev(x0).shape
    c1 = 0 if x1 is None else x1.shape[4]
    d = FiConv(dt(x0.dtype), N, H, W, ksize, c0, c1, dy.shape[4], 0, 0, 0, 0)
    nbytes = lib().fi_conv3d_wgrad_fused_workspace(C.byref(d), D)
    if nbytes == FI_ERR_UNSUPPORTED:
        return False
    if nbytes < 0:
        _chk(int(nbytes), "fi_conv3d_wgrad_fused_workspace")
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x0.device)      # caller-owned workspace
    cin, co, vox = c0 + c1, dy.shape[4], N * D * H * W
    with _timed("conv3d_wgrad", (str(x0.dtype)[6:], N, D, H, W, cin, co, ksize, "fused"), 2.0 * vox * cin * co * ksize ** 3,
                vox * (cin + co) * _esz(x0) + cin * co * ksize ** 3 * 4):
        _chk(lib().fi_conv3d_wgrad_fused(C.byref(d), D, ptr(x0), ptr(x1), ptr(dy), ptr(dw_all), ptr(dbias), ptr(ws),
                                         C.c_long(nbytes), stream()), "fi_conv3d_wgrad_fused")
    return True


def conv3d_wgrad_fused_partial(x0, x1, dy, want_bias, *, ksize):
    """Stage 1 of conv3d_wgrad_fused only.  -> (workspace tensor, slices, stride) for a later fi_wgrad_reduce_multi (table word 9 =
    cin), or None when the shape is not covered."""
    N, D, H, W, c0 = _dev(x0).shape
    c1 = 0 if x1 is None else x1.shape[4]
    d = FiConv(dt(x0.dtype), N, H, W, ksize, c0, c1, dy.shape[4], 0, 0, 0, 0)
    nbytes = lib().fi_conv3d_wgrad_fused_workspace(C.byref(d), D)
    if nbytes == FI_ERR_UNSUPPORTED:
        return None
    if nbytes < 0:
        _chk(int(nbytes), "fi_conv3d_wgrad_fused_workspace")
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x0.device)
    cin, co, vox = c0 + c1, dy.shape[4], N * D * H * W
    slices, stride = C.c_int(0), C.c_long(0)
    with _timed("conv3d_wgrad", (str(x0.dtype)[6:], N, D, H, W, cin, co, ksize, "fused"), 2.0 * vox * cin * co * ksize ** 3,
                vox * (cin + co) * _esz(x0) + cin * co * ksize ** 3 * 4):
        _chk(lib().fi_conv3d_wgrad_fused_partial(C.byref(d), D, ptr(x0), ptr(x1), ptr(dy), int(want_bias), ptr(ws), C.c_long(nbytes),
                                                 C.byref(slices), C.byref(stride), stream()), "fi_conv3d_wgrad_fused_partial")
    return ws, slices.value, stride.value


def convtranspose2x_fwd(x, w_packed, bias_taps, packed, y, N, D, H, W, cin, cout, three_d):
    _chk(lib().fi_convtranspose2x_fwd(dt(_dev(x).dtype), N, D, H, W, cin, cout, int(three_d), ptr(x), ptr(w_packed),
                                      ptr(bias_taps), ptr(packed), ptr(y), stream()), "fi_convtranspose2x_fwd")


def convtranspose2x_dgrad(dy, wt_packed, packed, dx, N, D, H, W, cin, cout, three_d):
    _chk(lib().fi_convtranspose2x_dgrad(dt(_dev(dy).dtype), N, D, H, W, cin, cout, int(three_d), ptr(dy), ptr(wt_packed),
                                        ptr(packed), ptr(dx), stream()), "fi_convtranspose2x_dgrad")


def convtranspose2x_wgrad(x, dy_packed, dw, dbias_taps, N, D, H, W, cin, cout, three_d):
    fn = lib().fi_convtranspose2x_wgrad_workspace
    fn.restype = C.c_long
    nbytes = fn(dt(_dev(x).dtype), N, D, H, W, cin, cout, int(three_d))
    if nbytes < 0:
        _chk(int(nbytes), "fi_convtranspose2x_wgrad_workspace")
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x.device)      # caller-owned workspace
    _chk(lib().fi_convtranspose2x_wgrad(dt(x.dtype), N, D, H, W, cin, cout, int(three_d), ptr(x), ptr(dy_packed), ptr(dw),
                                        ptr(dbias_taps), ptr(ws), C.c_long(nbytes), stream()), "fi_convtranspose2x_wgrad")


def depth_to_space2x(src, dst, N, D, H, W, Cc, three_d, inverse=False):
    _chk(lib().fi_depth_to_space2x(dt(src.dtype), ptr(_dev(src)), ptr(dst), int(N), int(D), int(H), int(W), int(Cc), int(three_d),
                                   int(inverse), stream()), "fi_depth_to_space2x")


def groupnorm_fwd(x, z, gamma, beta, mean, invstd, N, pixels, Cc, G, eps, relu):
    _chk(lib().fi_groupnorm_fwd(dt(x.dtype), ptr(_dev(x)), ptr(z), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), int(N),
                                C.c_long(pixels), int(Cc), int(G), C.c_float(eps), int(relu), stream()), "fi_groupnorm_fwd")


def groupnorm_bwd(dz, x, z, gamma, mean, invstd, dx, dgamma, dbeta, N, pixels, Cc, G, relu):
    _chk(lib().fi_groupnorm_bwd(dt(x.dtype), ptr(_dev(dz)), ptr(x), ptr(z), ptr(gamma), ptr(mean), ptr(invstd), ptr(dx),
                                ptr(dgamma), ptr(dbeta), int(N), C.c_long(pixels), int(Cc), int(G), int(relu), stream()),
         "fi_groupnorm_bwd")


def channel_stats(x, stats, pixels, Cc):
    _chk(lib().fi_channel_stats(dt(x.dtype), ptr(_dev(x)), ptr(stats), C.c_long(pixels), int(Cc), stream()), "fi_channel_stats")


def fedopt_step(mode, cur, agg, m, v, eta, beta1, beta2, tau):
    """mode 0 FedAdagrad, 1 FedAdam, 2 FedYogi; the python-float constants are rounded to fp32 the way numpy does."""
    import numpy as np
    f = lambda x: C.c_float(float(np.float32(x)))
    _chk(lib().fi_fedopt_step(int(mode), ptr(_dev(cur)), ptr(agg), ptr(m), ptr(v), C.c_long(cur.numel()), f(eta), f(beta1),
                              f(1.0 - beta1), f(beta2), f(1.0 - beta2), f(tau), stream()), "fi_fedopt_step")


def ala_update(w, temp, grad, local, glob, eta, skip=None):
    _chk(lib().fi_ala_update(ptr(_dev(w)), ptr(temp), ptr(grad), ptr(local), ptr(glob), C.c_long(w.numel()),
                             C.c_float(eta), ptr(skip), stream()), "fi_ala_update")


def global_avgmax(x, avg, mx, amax):
    N, H, W, Cc = _dev(x).shape
    S = lib().fi_global_avgmax_ranges(dt(x.dtype), N, H * W, Cc)
    if S >= 2:                                               # large maps: pixel ranges on separate workgroups
        ws = torch.empty(N * S * 3 * Cc, dtype=torch.float32, device=x.device)
        _chk(lib().fi_global_avgmax_split(dt(x.dtype), ptr(x), ptr(avg), ptr(mx), ptr(amax), N, H * W, Cc, ptr(ws),
                                          C.c_long(ws.numel() * 4), stream()), "fi_global_avgmax_split")
        return
    _chk(lib().fi_global_avgmax(dt(x.dtype), ptr(x), ptr(avg), ptr(mx), ptr(amax), N, H * W, Cc, stream()),
         "fi_global_avgmax")


def channel_gate_fwd(x, h, y):
    N, H, W, Cc = _dev(x).shape
    _chk(lib().fi_channel_gate_fwd(dt(x.dtype), ptr(x), ptr(h), ptr(y), N, H * W, Cc, stream()), "fi_channel_gate_fwd")


def channel_gate_bwd(x, dy, h, amax, davg, dmx, dx, dh):
    N, H, W, Cc = _dev(x).shape
    _chk(lib().fi_channel_gate_bwd(dt(x.dtype), ptr(x), ptr(dy), ptr(h), ptr(amax), ptr(davg), ptr(dmx), ptr(dx),
                                   ptr(dh), N, H * W, Cc, stream()), "fi_channel_gate_bwd")


def pcs_gate_fwd(avg, mx, who, w1a, w1b, w2a, w2b, h, hidden):
    B, Cc = _dev(avg).shape
    _chk(lib().fi_pcs_gate_fwd(ptr(avg), ptr(mx), ptr(who), ptr(w1a), ptr(w1b), ptr(w2a), ptr(w2b), ptr(h), ptr(hidden), B, Cc,
                               w1a.shape[1], stream()), "fi_pcs_gate_fwd")


def pcs_gate_bwd(dh, h, hidden, w2a, w2b, davg, dmx):
    B, Cc = _dev(h).shape
    _chk(lib().fi_pcs_gate_bwd(ptr(dh), ptr(h), ptr(hidden), ptr(w2a), ptr(w2b), ptr(davg), ptr(dmx), B, Cc, stream()),
         "fi_pcs_gate_bwd")


def lc_loss_fwd(h, others, loss_ce, alpha, G, out, dcoef):
    _chk(lib().fi_lc_loss_fwd(ptr(_dev(h)), ptr(others), ptr(loss_ce), C.c_float(alpha), int(G), h.numel(), ptr(out), ptr(dcoef),
                              stream()), "fi_lc_loss_fwd")


def lc_loss_bwd(dcoef, g, alpha, dh):
    _chk(lib().fi_lc_loss_bwd(ptr(_dev(dcoef)), ptr(g), C.c_float(alpha), ptr(dh), dcoef.numel(), stream()), "fi_lc_loss_bwd")


def cast(src, dst):
    _chk(lib().fi_cast(ptr(_dev(src)), dt(src.dtype), ptr(dst), dt(dst.dtype), C.c_long(src.numel()), stream()),
         "fi_cast")


def nchw_to_nhwc(src, dst):
    N, Cc, H, W = _dev(src).shape
    _chk(lib().fi_nchw_to_nhwc(ptr(src), ptr(dst), dt(dst.dtype), N, Cc, H, W, stream()), "fi_nchw_to_nhwc")


def nhwc_to_nchw(src, dst):
    N, Cc, H, W = dst.shape
    _chk(lib().fi_nhwc_to_nchw(ptr(_dev(src)), dt(src.dtype), ptr(dst), N, Cc, H, W, stream()), "fi_nhwc_to_nchw")


def probe_tr16(inp, offs, out):
    _chk(lib().fi_probe_tr16(ptr(_dev(inp)), ptr(offs), ptr(out), stream()), "fi_probe_tr16")
