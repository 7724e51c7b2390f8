"""Host-side mirror of the reference's operator surface for the OT hot path.

Same names, positional arguments, dtypes and return shapes as the reference's Python functions, so
the parity tests read like the reference's own call sites; each function is a thin shim that
hands raw device pointers to the C-ABI of include/pats_amd.h (libpats_amd.so, hand-written HIP for
gfx950).  PyTorch is used for device memory and streams only.  Tensors must live on a HIP device
("cuda" in torch-ROCm); there is no CPU path and no fallback.

Reference call sites (paths relative to zju3dv/pats):
  log_sinkhorn_iterations / log_optimal_transport / log_optimal_transport2   models/modules.py:137-182
  cost (einsum + scale)            models/first_layer.py:110-114 second_layer.py:100-104 third_layer.py:156-158
  Compute_positions_and_ranges     utils/utils.py:1527-1537
  Iterative_expand_matrix          utils/utils.py:1179-1297
  est_position (first / second)    models/first_layer.py:159-178  models/second_layer.py:240-259
  split_patches                    utils/utils.py:152-181
  Compute_imgs / tensor_resize     utils/utils.py:1343-1393  setup/library.cpp:47-66
  Compute_result / third label     models/third_layer.py:161-170,184-217
"""
import ctypes

import numpy as np
import torch

from . import _lib

_L = _lib.lib
_check = _lib.check
_L()   # load libpats_amd.so at import: a missing HIP extension fails here, loudly


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's current HIP stream as the C-ABI's pats_stream_t.  (torch.cuda.current_stream() builds a Stream object per call:
    9 us each, 0.8 ms of a pair walked chunk by chunk; the raw getter is what torch's own extensions use.)"""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("pats_amd: %s is on %s; the HIP path needs a GPU tensor (no CPU fallback)"
                           % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError("pats_amd: %s must be %s, got %s" % (name, dtype, t.dtype))
    return t.contiguous()


def _map(t, name, dtype=torch.float32):
    """A [B,C,H,W] feature map in either memory format -> (tensor whose storage a kernel can walk, channels_last?).
    A torch.channels_last tensor is used AS IT LIES (the channels-last gathers read it in full 64-byte granules);
    anything else is made NCHW-contiguous."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("pats_amd: %s is on %s; the HIP path needs a GPU tensor (no CPU fallback)" % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError("pats_amd: %s must be %s, got %s" % (name, dtype, t.dtype))
    if t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last):
        return t, True
    return t.contiguous(), False


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


_WS = {}
_WS_MAX = 64 << 20
_WS_ON = [False]


class workspace_cache:
    """Context: inside it the scratch block of a C call (up to 64 MB) is kept per (device, stream) and shared by consecutive
    calls instead of being allocated per call - the calls of a stream run in order and none reads its scratch after it returns.
    pipeline.forward_chunks_device uses it (a pair walked chunk by chunk made 300 allocator calls, a third of them these).
    Off by default: it is a host-side saving for many small calls on few streams; the throughput path (big tensors, their own
    allocations) keeps the allocator.  (profiles/r06_ws_cache_ab.txt: an apparent 4x cost to bench.py's masked-stream leg turned
    out to be a first-process-on-the-box effect, not the cache.)"""

    def __enter__(self):
        self.prev = _WS_ON[0]
        _WS_ON[0] = True

    def __exit__(self, *exc):
        _WS_ON[0] = self.prev


def _workspace(nbytes, device):
    """Scratch for ONE C call on the current stream (see workspace_cache)."""
    n = max(int(nbytes), 1)
    if not _WS_ON[0] or n > _WS_MAX or _raw_stream is None or torch.cuda.is_current_stream_capturing():
        return torch.empty(n, dtype=torch.uint8, device=device)
    key = (device.index, _raw_stream(torch.cuda.current_device()))
    t = _WS.get(key)
    if t is None or t.numel() < n:
        t = _WS[key] = torch.empty(min(_WS_MAX, max(2 * n, 1 << 20)), dtype=torch.uint8, device=device)
    return t


def _scalar_dev(x, device):
    """0-d tensor / python float -> device float32[1] without a device sync."""
    if isinstance(x, torch.Tensor):
        return x.detach().reshape(1).to(device=device, dtype=torch.float32)
    return torch.full((1,), float(x), dtype=torch.float32, device=device)


def set_sinkhorn_mode(mode):
    """'auto' | 'log' | 'kernel' (include/pats_amd.h PATS_SINKHORN_*). Returns the previous mode."""
    names = {"auto": 0, "log": 1, "kernel": 2}
    prev = _L().pats_set_sinkhorn_mode(names[mode])
    return {v: k for k, v in names.items()}[prev]


def set_fine_fused(on):
    """Fine level of cost_ot (variant 2, 145 x 145): True = the fused cost -> OT kernel (no score matrix in HBM), False (default)
    = MFMA cost kernel + Sinkhorn kernel.  Same bits; returns the previous setting."""
    return bool(_L().pats_set_fine_fused(1 if on else 0))


def sinkhorn_fallbacks(reset=True):
    """Problems on the current device whose linear-domain solve left the guard band and were redone
    with log-sum-exp sweeps since the last reset (pats_sinkhorn_fallbacks; synchronises)."""
    import ctypes
    n = ctypes.c_int64(0)
    rc = _L().pats_sinkhorn_fallbacks(ctypes.byref(n), 1 if reset else 0)
    if rc != 0:
        raise RuntimeError(_L().pats_last_error().decode())
    return int(n.value)


def set_gnn_redo(mode):
    """'inline' (default) | 'deferred' (include/pats_amd.h pats_set_gnn_redo_mode): in deferred mode the GNN layers queue no gated
    fp32 redo chain; an activation beyond the fp16 range raises a sticky device flag instead and the outputs of that call are not
    valid - read gnn_overflows() where you synchronise anyway and repeat the work under 'inline' if it says True.  Returns the
    previous mode."""
    names = {"inline": 0, "deferred": 1}
    prev = _L().pats_set_gnn_redo_mode(names[mode])
    return {v: k for k, v in names.items()}[prev]


def gnn_overflows(reset=True):
    """True if a GNN layer launched under set_gnn_redo('deferred') on the current device left the fp16 range since the last reset
    (pats_gnn_overflows; synchronises)."""
    n = ctypes.c_int64(0)
    _check(_L().pats_gnn_overflows(ctypes.byref(n), 1 if reset else 0), "gnn_overflows")
    return bool(n.value)


# ------------------------------------------------------------------------------------------------
# cost build
# ------------------------------------------------------------------------------------------------
def cost(mdesc0, mdesc1, out=None):
    """0.1 * (einsum('bdn,bdm->bnm', mdesc0, mdesc1) / D**.5)   (first_layer.py:110-114).
    out: optional preallocated [b, n, m] float32 result (as torch's `out=`)."""
    d0, d1 = _dev(mdesc0, "mdesc0"), _dev(mdesc1, "mdesc1")
    b, D, n = d0.shape
    if d1.shape[0] != b or d1.shape[1] != D:
        raise RuntimeError("cost: descriptor shapes %s / %s do not match" % (tuple(d0.shape), tuple(d1.shape)))
    m = d1.shape[2]
    if out is None:
        out = torch.empty((b, n, m), dtype=torch.float32, device=d0.device)
    elif tuple(out.shape) != (b, n, m) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != d0.device:
        raise RuntimeError("cost: out must be a contiguous float32 [%d, %d, %d] tensor on %s" % (b, n, m, d0.device))
    _check(_L().pats_cost_f32(_ptr(d0), _ptr(d1), b, D, n, m, _ptr(out), _stream()), "cost")
    return out


# ------------------------------------------------------------------------------------------------
# Sinkhorn / OT
# ------------------------------------------------------------------------------------------------
def log_sinkhorn_iterations(Z, log_mu, log_nu, iters: int):
    Z, log_mu, log_nu = _dev(Z, "Z"), _dev(log_mu, "log_mu"), _dev(log_nu, "log_nu")
    b, M, N = Z.shape
    if tuple(log_mu.shape) != (b, M) or tuple(log_nu.shape) != (b, N):
        raise RuntimeError("log_sinkhorn_iterations: marginal shapes do not match Z")
    out = torch.empty_like(Z)
    nb = _L().pats_sinkhorn_workspace_bytes(b, M, N)
    ws = _workspace(nb, Z.device)
    _check(_L().pats_sinkhorn_f32(_ptr(Z), b, M, N, _ptr(log_mu), _ptr(log_nu), int(iters), _ptr(out),
                                  _ptr(ws), nb, _stream()), "log_sinkhorn_iterations")
    return out


def log_optimal_transport(scores, alpha, ns, iters: int):
    scores = _dev(scores, "scores")
    b, m, n = scores.shape
    ns = _dev(ns, "ns")
    if ns.numel() != b * n:
        raise RuntimeError("log_optimal_transport: ns must have %d entries per batch" % n)
    ns = ns.reshape(b, n)
    a = _scalar_dev(alpha, scores.device)
    Z = torch.empty((b, m + 1, n + 1), dtype=torch.float32, device=scores.device)
    nb = _L().pats_ot_workspace_bytes(b, m + 1, n + 1)
    ws = _workspace(nb, scores.device)
    _check(_L().pats_log_optimal_transport_f32(_ptr(scores), b, m, n, _ptr(a), _ptr(ns), int(iters),
                                               _ptr(Z), _ptr(ws), nb, _stream()), "log_optimal_transport")
    return Z


def log_optimal_transport2(scores, one, ns, iters: int, bias_k: float = 0.0):
    """bias_k = 2 (outdoor) / 3 (indoor) folds the caller's dustbin `+= log(k)` of
    second_layer.py:107-112 into the epilogue; 0 returns exactly modules.py:165-182."""
    scores = _dev(scores, "scores")
    b, m, n = scores.shape
    ns = _dev(ns, "ns")
    if ns.numel() != b * (n - 1):
        raise RuntimeError("log_optimal_transport2: ns must have %d entries per batch" % (n - 1))
    ns = ns.reshape(b, n - 1)
    o = _scalar_dev(one, scores.device)
    Z = torch.empty((b, m, n), dtype=torch.float32, device=scores.device)
    nb = _L().pats_ot2_workspace_bytes(b, m, n)
    ws = _workspace(nb, scores.device)
    _check(_L().pats_log_optimal_transport2_f32(_ptr(scores), b, m, n, _ptr(o), _ptr(ns), int(iters),
                                                float(bias_k), _ptr(Z), _ptr(ws), nb, _stream()),
           "log_optimal_transport2")
    return Z


def set_cost_ot_mid_event(event):
    """Measurement hook: the next two-kernel cost_ot call records `event` (a torch.cuda.Event that has been recorded once, so
    that its handle exists; None clears) between its cost-build launch and its Sinkhorn launch."""
    h = None if event is None else ctypes.c_void_p(event.cuda_event)
    _check(_L().pats_set_cost_ot_mid_event(h), "set_cost_ot_mid_event")


def cost_ot(mdesc0, mdesc1, variant, scalar, ns, iters: int, bias_k: float = 0.0, return_flags=False, count=None):
    """descriptors -> log-plan (cost build + OT on one stream, score matrix never returned).
    return_flags (variant 2): also est_position's if_nomatching2 [b, m-1] (bool) from the OT epilogue ->
    (Z, col_nomatch); hand it to est_position_second(col_nomatch=...).
    count (fine level, with return_flags): DEVICE int64 [1] - the tensors are a capacity, only the first `count` problems are
    solved (the rows of the others are left as allocated)."""
    d0, d1 = _dev(mdesc0, "mdesc0"), _dev(mdesc1, "mdesc1")
    b, D, n = d0.shape
    m = d1.shape[2]
    ns = _dev(ns, "ns").reshape(b, m if variant == 1 else m - 1)
    s = _scalar_dev(scalar, d0.device)
    shape = (b, n + 1, m + 1) if variant == 1 else (b, n, m)
    Z = torch.empty(shape, dtype=torch.float32, device=d0.device)
    nb = _L().pats_cost_ot_workspace_bytes(b, D, n, m, variant)
    ws = _workspace(nb, d0.device)
    if return_flags:
        if variant != 2:
            raise RuntimeError("cost_ot: return_flags needs variant 2 (the coarse level gets them from colmass_sqrt)")
        flags = torch.empty((b, m - 1), dtype=torch.bool, device=d0.device)
        if count is not None:
            cnt = _dev(count, "count", torch.int64)
            _check(_L().pats_cost_ot_flags_counted_f32(_ptr(d0), _ptr(d1), b, _ptr(cnt), D, n, m, int(variant), _ptr(s), _ptr(ns),
                                                       int(iters), float(bias_k), _ptr(Z), _ptr(flags.view(torch.uint8)), _ptr(ws),
                                                       nb, _stream()), "cost_ot")
            return Z, flags
        _check(_L().pats_cost_ot_flags_f32(_ptr(d0), _ptr(d1), b, D, n, m, int(variant), _ptr(s), _ptr(ns), int(iters),
                                           float(bias_k), _ptr(Z), _ptr(flags.view(torch.uint8)), _ptr(ws), nb, _stream()),
               "cost_ot")
        return Z, flags
    if count is not None:
        raise RuntimeError("cost_ot: count needs return_flags=True (the fine level's counted launch)")
    _check(_L().pats_cost_ot_f32(_ptr(d0), _ptr(d1), b, D, n, m, int(variant), _ptr(s), _ptr(ns),
                                 int(iters), float(bias_k), _ptr(Z), _ptr(ws), nb, _stream()), "cost_ot")
    return Z


# ------------------------------------------------------------------------------------------------
# post-OT
# ------------------------------------------------------------------------------------------------
def colmass_sqrt(Z, return_flags=False):
    """sqrt(exp(Z[:, :-1, :-1]).sum(1) + 1e-8)   (first_layer.py:117-118).
    return_flags: the same pass over the columns also yields est_position's if_nomatching2 (first_layer.py:163,167)
    -> (scales, col_nomatch [b, N-1] bool)."""
    Z = _dev(Z, "Z")
    b, M, N = Z.shape
    out = torch.empty((b, N - 1), dtype=torch.float32, device=Z.device)
    if return_flags:
        flags = torch.empty((b, N - 1), dtype=torch.bool, device=Z.device)
        _check(_L().pats_colmass_flags_f32(_ptr(Z), b, M, N, _ptr(out), _ptr(flags.view(torch.uint8)), _stream()),
               "colmass_sqrt")
        return out, flags
    _check(_L().pats_colmass_sqrt_f32(_ptr(Z), b, M, N, _ptr(out), _stream()), "colmass_sqrt")
    return out


def dustbin_bias_(Z, k):
    """In place: Z[:, :, -1] += log(k); Z[:, -1, :] += log(k)   (second_layer.py:107-112)."""
    if not Z.is_contiguous():
        raise RuntimeError("dustbin_bias_: Z must be contiguous (in-place op)")
    _dev(Z, "Z")
    b, M, N = Z.shape
    _check(_L().pats_dustbin_bias_inplace_f32(_ptr(Z), b, M, N, float(k), _stream()), "dustbin_bias_")
    return Z


def exp(Z):
    Z = _dev(Z, "Z")
    out = torch.empty_like(Z)
    _check(_L().pats_exp_f32(_ptr(Z), Z.numel(), _ptr(out), _stream()), "exp")
    return out


def argmax(Z):
    """(scores.max(2).indices, scores.max(1).indices), first index on ties (first_layer.py:162)."""
    Z = _dev(Z, "Z")
    b, M, N = Z.shape
    r = torch.empty((b, M), dtype=torch.int64, device=Z.device)
    c = torch.empty((b, N), dtype=torch.int64, device=Z.device)
    _check(_L().pats_argmax_f32(_ptr(Z), b, M, N, _ptr(r), _ptr(c), _stream()), "argmax")
    return r, c


# ------------------------------------------------------------------------------------------------
# patch-area expansion
# ------------------------------------------------------------------------------------------------
_POS_CACHE = {}


def Compute_positions_and_ranges(height, width, device):
    """utils/utils.py:1527-1537.  The returned tensors carry the grid as `_pats_grid` so
    Iterative_expand_matrix does not have to read them back from the device.  They depend on (height, width, device)
    only and are built once (treat them as read-only): the per-call host-to-device copies of the first version were
    stream-ordered pageable copies, i.e. the host waited for everything queued before them."""
    key = (int(height), int(width), str(torch.device(device)))
    hit = _POS_CACHE.get(key)
    if hit is not None:
        return hit
    k = torch.arange(height * width)
    positions = torch.stack([(k // width).float(), (k % width).float()], dim=1)
    max_shape = max(height, width)
    kk = torch.arange(max_shape).float()
    ranges = torch.where(kk[None, :] <= kk[:, None], kk[None, :].expand(max_shape, -1),
                         torch.full((max_shape, max_shape), 1e7))
    positions, ranges = positions.to(device), ranges.to(device)
    positions._pats_grid = (int(height), int(width))
    ranges._pats_grid = (int(height), int(width))
    _POS_CACHE[key] = (positions, ranges)
    return positions, ranges


def _grid_of(positions, ranges):
    """(h, w) of the index tables handed to Iterative_expand_matrix.  The expansion kernel does not READ the two tensors: it forms
    positions[k] = (k // w, k % w) and ranges[i] = [0 .. i, 1e7 ...] itself (utils.py:1527-1537 - what every call site of the
    reference passes, first_layer.py:173-175 / second_layer.py:254-256).  The reference does honour other contents (a shifted
    `ranges` changes 311 of 1 152 rectangle bounds of the golden case, tests/golden/positions_ranges.npz), so tensors that do
    not come from Compute_positions_and_ranges above are CHECKED against those tables once (one device comparison) and
    anything else is refused instead of being silently replaced."""
    g = getattr(positions, "_pats_grid", None) or getattr(ranges, "_pats_grid", None)
    if g is not None:
        return g
    # foreign tensors: recover (h, w) from the values (one device read), then hold both to the canonical tables
    if positions.dim() != 2 or positions.shape[1] != 2 or ranges.dim() != 2 or ranges.shape[0] != ranges.shape[1]:
        raise RuntimeError("Iterative_expand_matrix: positions must be [h*w,2] and ranges [max(h,w),max(h,w)]")
    w = int((positions[:, 0] == 0).sum().item())
    if w <= 0 or positions.shape[0] % w != 0:
        raise RuntimeError("Iterative_expand_matrix: positions is not the table of Compute_positions_and_ranges")
    h = positions.shape[0] // w
    cp, cr = Compute_positions_and_ranges(h, w, positions.device)
    if tuple(ranges.shape) != tuple(cr.shape) or not torch.equal(positions.float(), cp) or not torch.equal(ranges.float().to(cr.device), cr):
        raise RuntimeError("Iterative_expand_matrix: only the index tables Compute_positions_and_ranges builds are supported "
                           "(positions[k] = (k // w, k % w), ranges[i] = [0 .. i, 1e7 ...]); the tensors handed in differ")
    try:
        positions._pats_grid = ranges._pats_grid = (h, w)      # checked once
    except Exception:                                          # noqa: BLE001
        pass
    return h, w


def Iterative_expand_matrix(scores_in, scalex, scaley, limitation, ranges, positions,
                            lower_bound=1e-3, upper_bound=1e7, iter_num=15, width=20, height=15,
                            type="distance", input_is_log=False, row_nomatch=None, count=None):
    """utils/utils.py:1179-1297.  Returns (whole_cost, core_cost, average_point, x_scale, y_scale,
    bound) with the reference's shapes/dtypes.  `width`/`height`/`upper_bound`/`type` are accepted
    and ignored exactly as the reference ignores them (it re-derives width/height at :1181)."""
    P = _dev(scores_in, "scores_in")
    b, M, N = P.shape
    sx = _dev(scalex, "scalex").reshape(b, -1)
    sy = _dev(scaley, "scaley").reshape(b, -1)
    if sx.shape[1] != N - 1 or sy.shape[1] != N - 1:
        raise RuntimeError("Iterative_expand_matrix: scale tensors must have %d entries" % (N - 1))
    h, w = _grid_of(positions, ranges)
    if isinstance(limitation, torch.Tensor):
        lim3 = getattr(limitation, "_pats_lim3", None)
        if lim3 is None:
            lim3 = int(limitation[3].item())
    else:
        lim3 = int(limitation[3])
    m = M - 1
    dev = P.device
    whole = torch.empty((b, m), dtype=torch.float32, device=dev)
    core = torch.empty((b, m), dtype=torch.float32, device=dev)
    avg = torch.empty((b, m, 2), dtype=torch.float32, device=dev)
    xs = torch.empty((b, m), dtype=torch.float32, device=dev)
    ys = torch.empty((b, m), dtype=torch.float32, device=dev)
    bound = torch.empty((b, m, 4), dtype=torch.int64, device=dev)
    rn = _ptr(row_nomatch.view(torch.uint8)) if row_nomatch is not None else _ptr(None)
    if count is not None:           # not in the reference's signature: a device-side batch count (throughput mode)
        _check(_L().pats_iterative_expand_counted_f32(_ptr(P), int(bool(input_is_log)), b, _ptr(_dev(count, "count", torch.int64)),
                                                      M, N, _ptr(sx), _ptr(sy), lim3, h, w, float(lower_bound), int(iter_num),
                                                      _ptr(whole), _ptr(core), _ptr(avg), _ptr(xs), _ptr(ys), _ptr(bound), rn,
                                                      _stream()), "Iterative_expand_matrix")
        return whole, core, avg, xs, ys, bound
    _check(_L().pats_iterative_expand_f32(_ptr(P), int(bool(input_is_log)), b, M, N, _ptr(sx), _ptr(sy),
                                          lim3, h, w, float(lower_bound), int(iter_num), _ptr(whole),
                                          _ptr(core), _ptr(avg), _ptr(xs), _ptr(ys), _ptr(bound), rn,
                                          _stream()), "Iterative_expand_matrix")
    return whole, core, avg, xs, ys, bound


def _est_position(scores, scale_x, scale_y, H, W, patch_scale, iter_num, lower_bound, col_nomatch=None, count=None):
    """est_position (first_layer.py:159-178 / second_layer.py:240-259) without a separate argmax pass: the row flag
    `scores.max(2).indices[:, :-1] == h*w` comes out of the expansion kernel (which holds every row anyway), the
    column flag from the caller (OT epilogue / colmass pass) or, failing that, from one pass over the columns."""
    b, M, N = scores.shape
    h, w = H // patch_scale, W // patch_scale
    if col_nomatch is None:
        col_nomatch = torch.empty((b, N - 1), dtype=torch.bool, device=scores.device)
        _check(_L().pats_colmass_flags_f32(_ptr(_dev(scores, "scores")), b, M, N, _ptr(None),
                                           _ptr(col_nomatch.view(torch.uint8)), _stream()), "est_position")
    if_nomatching1 = torch.empty((b, M - 1), dtype=torch.bool, device=scores.device)
    positions1, ranges1 = Compute_positions_and_ranges(h, w, scores.device)
    limitation1 = [0, h, 0, w]
    trust_score, _, average_point1, x_scale, y_scale, _ = Iterative_expand_matrix(
        scores, scale_x.reshape(b, -1, 1), scale_y.reshape(b, -1, 1), limitation1, ranges1, positions1,
        height=h, width=w, iter_num=iter_num, lower_bound=lower_bound, input_is_log=True, row_nomatch=if_nomatching1, count=count)
    return trust_score, average_point1, x_scale, y_scale, if_nomatching1, col_nomatch


def est_position_first(scores, scale_src, image_shape, patch_scale, col_nomatch=None):
    """FirstLayer.est_position (first_layer.py:159-178): scores is the LOG plan; exp() is fused
    into the expansion kernel's load.  col_nomatch: if_nomatching2 when the caller already has it
    (colmass_sqrt(return_flags=True))."""
    H, W = image_shape
    return _est_position(scores, scale_src, scale_src, H, W, patch_scale, 15, 1e-5, col_nomatch)


def est_position_second(scores, scale_x, scale_y, image_shape, patch_scale, col_nomatch=None, count=None):
    """SecondLayer.est_position (second_layer.py:240-259).  col_nomatch: if_nomatching2 from
    cost_ot(..., return_flags=True).  count: device-side batch count (throughput mode), as for cost_ot."""
    H, W = image_shape
    return _est_position(scores, scale_x, scale_y, H, W, patch_scale, 8, 1e-3, col_nomatch, count)


# ------------------------------------------------------------------------------------------------
# chunk planner (host)
# ------------------------------------------------------------------------------------------------
def split_patches(sum_cycle, height, width, max_once_used=350):
    """utils/utils.py:152-181.  One D->H copy of the cumsum (none if it already is a CPU tensor /
    numpy array, e.g. one row of a batch fetched once for many pairs), then host C++."""
    if isinstance(sum_cycle, np.ndarray):
        sc = np.ascontiguousarray(sum_cycle, dtype=np.int32)
    else:
        sc = sum_cycle.detach().to("cpu", torch.int32).contiguous().numpy()
    if sc.shape[0] != height * width:
        raise RuntimeError("split_patches: sum_cycle must have height*width entries")
    second = np.zeros((height + 1, 2), np.int64)
    third = np.zeros((height + 1, 2), np.int64)
    n = _L().pats_split_patches(sc.ctypes.data_as(ctypes.c_void_p), int(height), int(width),
                                int(max_once_used), second.ctypes.data_as(ctypes.c_void_p),
                                third.ctypes.data_as(ctypes.c_void_p))
    if n < 1:
        _check(-n, "split_patches")
    return n, second[:n].tolist(), third[:n].tolist()


def split_patches_device(sum_cycle, height, width, max_once_used=350):
    """split_patches (utils/utils.py:152-181) for a batch of pairs WITHOUT leaving the device:
    sum_cycle [pairs, height*width] int32 -> (cycle_num [pairs] int32, second_layer_set [pairs,height+1,2],
    third_layer_set [pairs,height+1,2] int64), rows past cycle_num zeroed.  No host read."""
    sc = _dev(sum_cycle, "sum_cycle", torch.int32)
    pairs = sc.shape[0]
    if sc.dim() != 2 or sc.shape[1] != height * width:
        raise RuntimeError("split_patches_device: sum_cycle must be [pairs, height*width]")
    second = torch.empty((pairs, height + 1, 2), dtype=torch.int64, device=sc.device)
    third = torch.empty((pairs, height + 1, 2), dtype=torch.int64, device=sc.device)
    num = torch.empty((pairs,), dtype=torch.int32, device=sc.device)
    _check(_L().pats_split_patches_device(_ptr(sc), pairs, int(height), int(width), int(max_once_used), _ptr(second),
                                          _ptr(third), _ptr(num), _stream()), "split_patches_device")
    return num, second, third


# ------------------------------------------------------------------------------------------------
# subdivision gather
# ------------------------------------------------------------------------------------------------
def tensor_resize(input_tensor, bound, validate=True):
    """tensor_resize.tensor_resize(input, bound)  (setup/library.cpp:47-66,92-93).
    input [n,C,Hp,Wp] float32, bound [K,5] int64 -> new [K,C,96,96] float32 on input.device.
    Raises RuntimeError for an empty or out-of-range crop like the reference (validate=True)."""
    inp = _dev(input_tensor, "input_tensor")
    bnd = _dev(bound, "bound", torch.int64)
    if inp.dim() != 4 or bnd.dim() != 2 or bnd.shape[1] != 5:
        raise RuntimeError("tensor_resize: expected input [n,C,H,W] and bound [K,5]")
    n_img, C, Hp, Wp = inp.shape
    K = bnd.shape[0]
    out = torch.empty((K, C, 96, 96), dtype=torch.float32, device=inp.device)
    status = torch.zeros((1,), dtype=torch.int32, device=inp.device) if validate else None
    _check(_L().pats_tensor_resize_f32(_ptr(inp), n_img, C, Hp, Wp, _ptr(bnd), K, _ptr(out),
                                       _ptr(status), _stream()), "tensor_resize")
    if validate and K > 0 and int(status.item()) != 0:
        # library.cpp:56-60: narrow() outside the tensor / an empty crop into upsample_bilinear2d is a
        # c10::Error there (RuntimeError / IndexError in Python).  One host read per call (the reference
        # makes five per crop); validate=False skips it and leaves such crops zero-filled.
        raise RuntimeError("tensor_resize: a crop is empty or outside the %dx%d input (start/length out of range, "
                           "or image index >= %d)" % (Hp, Wp, n_img))
    return out


def Compute_imgs(x_scale, y_scale, average_point, if_nomatching, left, right, sequence_num=0,
                 output_path=None, if_view=False, margin=128, width=20, height=15, patch_scale=32, known_count=None,
                 validate=False):
    """utils/utils.py:1343-1393 - same 5-tuple as the reference."""
    return Compute_imgs_ex(x_scale, y_scale, average_point, if_nomatching, left, right, sequence_num,
                           output_path, if_view, margin, width, height, patch_scale, known_count, validate)[:5]


def Compute_imgs_ex(x_scale, y_scale, average_point, if_nomatching, left, right, sequence_num=0,
                    output_path=None, if_view=False, margin=128, width=20, height=15, patch_scale=32,
                    known_count=None, validate=False):
    """Compute_imgs plus the [K,5] bound tensor the reference hands to tensor_resize (utils.py:1382).
    utils/utils.py:1343-1393, any batch of images (PATS.forward uses 1, first_layer.py:135; a batch
    yields the crops of all images in (image, patch) order - `sequence = img * 10000 + patch`, :1374-1377).
    Returns (new_left [K,96,96,3], new_right [K,96,96,3], x_scale_new [1,N,2], y_scale_new
    [1,N,2], average_new [1,N,2], bound5 [K,5]).  One host read (K) sizes the outputs, as the reference's
    boolean-mask indexing does - unless the caller already knows the matched patches per image
    (`known_count`, e.g. the last entry of the cumsum it fetched for split_patches): then no sync;
    validate=True checks those counts against the device-side ones (one host read).
    known_count="device" never touches the host: outputs are sized for the capacity n_img*N, only the first
    K_total rows are written, and the tuple gains (K_img [n_img], K_total [1]) int64 DEVICE tensors."""
    if margin != 128 or patch_scale != 32:
        raise RuntimeError("Compute_imgs: margin=128 / patch_scale=32 are what the path uses")
    nb = left.shape[0]                          # images in the batch; crops come out ordered (image, patch)
    dev = x_scale.device
    Np = width * height
    xs = _dev(x_scale.float(), "x_scale").reshape(nb, Np)
    ys = _dev(y_scale.float(), "y_scale").reshape(nb, Np)
    ap = _dev(average_point.float(), "average_point").reshape(nb, Np, 2)
    ifn = _as_flags(if_nomatching, "if_nomatching").reshape(nb, Np)         # bool viewed as bytes: no copy kernel
    leftf = _dev(left.float(), "left")
    rightf = _dev(right.float(), "right")
    H, W = leftf.shape[1], leftf.shape[2]
    on_device = isinstance(known_count, str) and known_count == "device"
    counts = None
    if known_count is not None and not on_device:
        if isinstance(known_count, torch.Tensor):
            known_count = known_count.tolist()          # a (GPU) tensor of counts: one host read, like int(tensor)
        counts = [int(k) for k in np.atleast_1d(np.asarray(known_count)).ravel()]
        if len(counts) != nb:
            raise RuntimeError("Compute_imgs: known_count needs one entry per image")
    bound5 = torch.empty((nb * Np, 5), dtype=torch.int64, device=dev)
    Kd = torch.empty((nb,), dtype=torch.int64, device=dev)
    Kt = torch.empty((1,), dtype=torch.int64, device=dev)
    xsn = torch.empty((nb, Np, 2), dtype=torch.float32, device=dev)
    ysn = torch.empty((nb, Np, 2), dtype=torch.float32, device=dev)
    avn = torch.empty((nb, Np, 2), dtype=torch.float32, device=dev)
    # all images in one launch: block i offsets its compacted bounds by the matches of the images before it
    _check(_L().pats_compute_imgs_bounds_batch_f32(_ptr(xs), _ptr(ys), _ptr(ap), _ptr(ifn), nb, Np, height, width,
                                                   _ptr(bound5), _ptr(Kd), _ptr(Kt), _ptr(xsn), _ptr(ysn), _ptr(avn),
                                                   _stream()), "Compute_imgs(bounds)")
    status = torch.zeros((1,), dtype=torch.int32, device=dev) if validate else None
    if on_device:
        cap = nb * Np
        new_left = torch.empty((cap, 96, 96, 3), dtype=torch.float32, device=dev)
        new_right = torch.empty((cap, 96, 96, 3), dtype=torch.float32, device=dev)
        _check(_L().pats_left_crops_counted_f32(_ptr(leftf), nb, H, W, _ptr(bound5), cap, _ptr(Kt), height, width,
                                                _ptr(new_left), _stream()), "Compute_imgs(left)")
        _check(_L().pats_tensor_resize_hwc_counted_f32(_ptr(rightf), nb, H, W, margin, _ptr(bound5), cap, _ptr(Kt),
                                                       _ptr(new_right), _ptr(status), _stream()), "Compute_imgs(right)")
        if validate and int(status.item()) != 0:
            raise RuntimeError("Compute_imgs: a right crop is empty or outside the padded image")
        return new_left, new_right, xsn, ysn, avn, bound5, Kd, Kt
    if counts is None:
        counts = [int(k) for k in Kd.tolist()]                  # the host read
    elif validate:
        got = [int(k) for k in Kd.tolist()]
        if got != counts:
            raise RuntimeError("Compute_imgs: known_count %s does not match the matched patches per image %s" % (counts, got))
    K = sum(counts)
    new_left = torch.empty((K, 96, 96, 3), dtype=torch.float32, device=dev)
    new_right = torch.empty((K, 96, 96, 3), dtype=torch.float32, device=dev)
    _check(_L().pats_left_crops_f32(_ptr(leftf), nb, H, W, _ptr(bound5), K, height, width, _ptr(new_left),
                                    _stream()), "Compute_imgs(left)")
    _check(_L().pats_tensor_resize_hwc_f32(_ptr(rightf), nb, H, W, margin, _ptr(bound5), K,
                                           _ptr(new_right), _ptr(status), _stream()),
           "Compute_imgs(right)")
    if validate and K > 0 and int(status.item()) != 0:
        raise RuntimeError("Compute_imgs: a right crop is empty or outside the padded image")
    new_left = new_left.to(left.dtype) if left.dtype != torch.float32 else new_left
    return new_left, new_right, xsn, ysn, avn, bound5[:K]


# ------------------------------------------------------------------------------------------------
# third level
# ------------------------------------------------------------------------------------------------
def Compute_result(scores, W, T, scale_x, scale_y, p_s, p_t, device=None, outdoor=True,
                   input_is_log=False):
    """ThirdLayer.Compute_result (third_layer.py:184-217) + the label rule (:161-170).
    Returns (mkpts0_f, mkpts1_f, whole_loss, label, if_matching1)."""
    if W != 8 or T != 5:
        raise RuntimeError("Compute_result: the path uses W=8, T=5 (third_layer.py:108-111)")
    S = _dev(scores, "scores")
    P = S.shape[0]
    if tuple(S.shape[1:]) != (65, 65):
        raise RuntimeError("Compute_result: scores must be [P,65,65]")
    sx = _dev(scale_x, "scale_x").reshape(P, 64)
    sy = _dev(scale_y, "scale_y").reshape(P, 64)
    ps = _dev(p_s.to(torch.int64), "p_s", torch.int64).reshape(P, 2)
    pt = _dev(p_t.to(torch.int64), "p_t", torch.int64).reshape(P, 2)
    dev = S.device
    m0 = torch.empty((P, 16, 2), dtype=torch.float32, device=dev)
    m1 = torch.empty((P, 16, 2), dtype=torch.float32, device=dev)
    wl = torch.empty((P, 16), dtype=torch.float32, device=dev)
    label = torch.empty((P * 16, 2), dtype=torch.float32, device=dev)
    ifm = torch.empty((P, 16), dtype=torch.uint8, device=dev)
    ws = torch.empty((1,), dtype=torch.int32, device=dev)          # whole_loss' cross-problem count: no allocation in the call
    _check(_L().pats_compute_result_ws_f32(_ptr(S), int(bool(input_is_log)), P, _ptr(sx), _ptr(sy), _ptr(ps),
                                           _ptr(pt), int(bool(outdoor)), _ptr(m0), _ptr(m1), _ptr(wl),
                                           _ptr(label), _ptr(ifm), _ptr(ws), 4, _stream()), "Compute_result")
    return m0, m1, wl, label, ifm.bool()


def third_level(feat_f0_unfold, feat_f1_unfold, scale, mkpts0_c, mkpts1_c, outdoor=True, iters=100,
                return_plan=False, count=None):
    """The third layer's whole OT step in one launch (third_layer.py:153-170):
        scale_x = scale_y = sqrt(scale + 1e-8)
        scores  = exp(log_optimal_transport2(0.1 * einsum(f0, f1) / 128**.5, 1, scale, 100))
        mkpts0_f, mkpts1_f, _ = Compute_result(scores, 8, 5, scale_x, scale_y, mkpts0_c, mkpts1_c)
        label / if_matching1 as :161-170
    Returns (mkpts0_f, mkpts1_f, label, if_matching1[, Z]).  The 65x65 plans stay on chip.
    count: a DEVICE int64 [1] holding the number of problems that exist (throughput mode: the tensors are sized for a
    capacity, nothing is read back); rows past it are not written, and sqrt(scale + 1e-8) is formed in the kernel."""
    f0, f1 = _dev(feat_f0_unfold, "feat_f0_unfold"), _dev(feat_f1_unfold, "feat_f1_unfold")
    P, D, n = f0.shape
    if n != 65 or tuple(f1.shape) != (P, D, 65):
        raise RuntimeError("third_level: descriptors must be [P,D,65]")
    sc = _dev(scale, "scale").reshape(P, 64)
    ps = _dev(mkpts0_c.to(torch.int64), "mkpts0_c", torch.int64).reshape(P, 2)
    pt = _dev(mkpts1_c.to(torch.int64), "mkpts1_c", torch.int64).reshape(P, 2)
    dev = f0.device
    if count is not None:
        if return_plan:
            raise RuntimeError("third_level: return_plan is not available with a device-side count")
        cnt = _dev(count, "count", torch.int64).reshape(1)
        m0 = torch.empty((P, 16, 2), dtype=torch.float32, device=dev)
        m1 = torch.empty((P, 16, 2), dtype=torch.float32, device=dev)
        label = torch.empty((P * 16, 2), dtype=torch.float32, device=dev)
        ifm = torch.empty((P, 16), dtype=torch.uint8, device=dev)
        _check(_L().pats_third_level_counted_f32(_ptr(f0), _ptr(f1), P, _ptr(cnt), D, _ptr(sc), _ptr(None), _ptr(None), _ptr(ps),
                                                 _ptr(pt), int(iters), int(bool(outdoor)), _ptr(m0), _ptr(m1), _ptr(label),
                                                 _ptr(ifm), _stream()), "third_level")
        return m0, m1, label, ifm.view(torch.bool)
    sxy = torch.sqrt(sc + 1e-8)
    m0 = torch.empty((P, 16, 2), dtype=torch.float32, device=dev)
    m1 = torch.empty((P, 16, 2), dtype=torch.float32, device=dev)
    label = torch.empty((P * 16, 2), dtype=torch.float32, device=dev)
    ifm = torch.empty((P, 16), dtype=torch.uint8, device=dev)
    Z = torch.empty((P, 65, 65), dtype=torch.float32, device=dev) if return_plan else None
    _check(_L().pats_third_level_f32(_ptr(f0), _ptr(f1), P, D, _ptr(sc), _ptr(sxy), _ptr(sxy), _ptr(ps),
                                     _ptr(pt), int(iters), int(bool(outdoor)), _ptr(m0), _ptr(m1),
                                     _ptr(label), _ptr(ifm), _ptr(Z), _stream()), "third_level")
    return (m0, m1, label, ifm.bool(), Z) if return_plan else (m0, m1, label, ifm.bool())


def fine_descriptors(desc0_, title, rubbish, out=None, count=None):
    """second_layer.py:71-86: desc0_ = the three maps of ResNet2.forward2 on the stacked crops
    ([2B,64,48,48], [2B,64,24,24], [2B,128,12,12]); title [B,8] = compress_1(desc_l); rubbish [B,264]
    = compress_2(desc_l).  Returns desc [2,B,264,145] (desc[0], desc[1] feed the GNN).
    Maps in torch.channels_last memory format (all three) take the channels-last gather: same bits, 0.67x the HBM bytes."""
    maps = [_map(t, "desc0_[%d]" % i) for i, t in enumerate(desc0_)]
    if len({cl for _, cl in maps}) != 1:       # mixed formats: fall back to the NCHW kernel on contiguous copies
        maps = [(t.contiguous(), False) for t, _ in maps]
    (f0, nhwc), (f1, _), (f2, _) = maps
    B = f0.shape[0] // 2
    if tuple(f0.shape[1:]) != (64, 48, 48) or tuple(f1.shape) != (2 * B, 64, 24, 24) or \
            tuple(f2.shape) != (2 * B, 128, 12, 12):
        raise RuntimeError("fine_descriptors: unexpected feature-map shapes")
    ti = _dev(title, "title").reshape(B, 8)
    ru = _dev(rubbish, "rubbish").reshape(B, 264)
    desc = torch.empty((2, B, 264, 145), dtype=torch.float32, device=f0.device) if out is None else _dev(out, "out")
    if tuple(desc.shape) != (2, B, 264, 145) or (out is not None and desc.data_ptr() != out.data_ptr()):
        raise RuntimeError("fine_descriptors: out must be a contiguous [2,B,264,145] tensor")
    if count is not None:      # device-side row count: the tensors are a capacity (throughput mode)
        _check(_L().pats_fine_descriptors_counted_f32(_ptr(f0), _ptr(f1), _ptr(f2), _ptr(ti), _ptr(ru), B,
                                                      _ptr(_dev(count, "count", torch.int64)), int(bool(nhwc)), _ptr(desc),
                                                      _stream()), "fine_descriptors")
        return desc
    fn = _L().pats_fine_descriptors_nhwc_f32 if nhwc else _L().pats_fine_descriptors_f32
    _check(fn(_ptr(f0), _ptr(f1), _ptr(f2), _ptr(ti), _ptr(ru), B, _ptr(desc), _stream()), "fine_descriptors")
    return desc


def third_descriptors(feat_f0, feat_f1, mkpts0_c, mkpts1_c, b_ids, kenc, rubbish, count=None, out=None):
    """third_layer.py:121-146.  Returns (feat_f0_unfold, feat_f1_unfold [P,128,65], mkpts0_c,
    mkpts1_c [P,2] int64 rounded to the 4-px lattice as the reference reassigns them).
    count: DEVICE int64 [1], the number of points that exist (the tensors are a capacity; rows past it are not written);
    out: optional (o0, o1) to write into.
    Maps in torch.channels_last memory format take the channels-last gather: same bits, under half the HBM bytes."""
    (f0, nhwc), (f1, nhwc1) = _map(feat_f0, "feat_f0"), _map(feat_f1, "feat_f1")
    if nhwc != nhwc1:
        f0, f1, nhwc = f0.contiguous(), f1.contiguous(), False
    B = f0.shape[0]
    if tuple(f0.shape[1:]) != (128, 52, 52) or f1.shape != f0.shape:
        raise RuntimeError("third_descriptors: feature maps must be [B,128,52,52]")
    m0 = _dev(mkpts0_c.float(), "mkpts0_c").reshape(-1, 2)
    m1 = _dev(mkpts1_c.float(), "mkpts1_c").reshape(-1, 2)
    P = m0.shape[0]
    bi = _dev(b_ids.to(torch.int64), "b_ids", torch.int64).reshape(P)
    ke = _dev(kenc, "kenc").reshape(128, 64)
    ru = _dev(rubbish, "rubbish").reshape(B, 128, 144)
    dev = f0.device
    if out is not None:
        o0, o1 = _dev(out[0], "out[0]"), _dev(out[1], "out[1]")
        if tuple(o0.shape) != (P, 128, 65) or o1.shape != o0.shape or o0.data_ptr() != out[0].data_ptr():
            raise RuntimeError("third_descriptors: out must be two contiguous [P,128,65] tensors")
    else:
        o0 = torch.empty((P, 128, 65), dtype=torch.float32, device=dev)
        o1 = torch.empty((P, 128, 65), dtype=torch.float32, device=dev)
    ps = torch.empty((P, 2), dtype=torch.int64, device=dev)
    pt = torch.empty((P, 2), dtype=torch.int64, device=dev)
    cnt = _dev(count, "count", torch.int64).reshape(1) if count is not None else None
    if nhwc:
        _check(_L().pats_third_descriptors_nhwc_f32(_ptr(f0), _ptr(f1), _ptr(m0), _ptr(m1), _ptr(bi), _ptr(ke), _ptr(ru),
                                                    P, _ptr(cnt), B, _ptr(o0), _ptr(o1), _ptr(ps), _ptr(pt), _stream()),
               "third_descriptors")
        return o0, o1, ps, pt
    if count is not None:
        _check(_L().pats_third_descriptors_counted_f32(_ptr(f0), _ptr(f1), _ptr(m0), _ptr(m1), _ptr(bi), _ptr(ke), _ptr(ru),
                                                       P, _ptr(cnt), B, _ptr(o0), _ptr(o1), _ptr(ps), _ptr(pt), _stream()),
               "third_descriptors")
        return o0, o1, ps, pt
    _check(_L().pats_third_descriptors_f32(_ptr(f0), _ptr(f1), _ptr(m0), _ptr(m1), _ptr(bi), _ptr(ke), _ptr(ru),
                                           P, B, _ptr(o0), _ptr(o1), _ptr(ps), _ptr(pt), _stream()),
           "third_descriptors")
    return o0, o1, ps, pt


# ------------------------------------------------------------------------------------------------
# the steps either side of the OT path
# ------------------------------------------------------------------------------------------------
def _as_flags(t, name):
    """bool / uint8 tensor -> contiguous uint8 view sharing storage when it already is contiguous bool."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("pats_amd: %s must be a GPU tensor (no CPU fallback)" % name)
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)
    if t.dtype != torch.uint8:
        raise RuntimeError("pats_amd: %s must be bool or uint8, got %s" % (name, t.dtype))
    return t.contiguous()


def _merge(merge_new, patch_num, trust_score, original_image_shape, if_nomatching1_L1, if_nomatching1_L2,
           scores_back, validate):
    if trust_score.dtype != torch.float32 or not trust_score.is_cuda or not trust_score.is_contiguous():
        raise RuntimeError("merge_patches: trust_score must be a contiguous float32 GPU tensor (it is updated in place)")
    if if_nomatching1_L2.dtype != torch.bool or not if_nomatching1_L2.is_contiguous():
        raise RuntimeError("merge_patches: if_nomatching1_L2 must be a contiguous bool tensor (it is updated in place)")
    if scores_back.dtype != torch.float64 or not scores_back.is_contiguous():
        raise RuntimeError("merge_patches: scores_back must be a contiguous float64 tensor (it is updated in place)")
    B = trust_score.shape[0]
    if trust_score.numel() != B * 144 or if_nomatching1_L2.numel() != B * 144 or int(patch_num) != B:
        raise RuntimeError("merge_patches: trust_score / if_nomatching1_L2 must be [patch_num,144]")
    H, W = int(original_image_shape[0]), int(original_image_shape[1])
    l1 = _as_flags(if_nomatching1_L1, "if_nomatching1_L1")
    bt = l1.shape[0]
    if l1.numel() != bt * (H // 32) * (W // 32) or scores_back.numel() != l1.numel() * 144:
        raise RuntimeError("merge_patches: if_nomatching1_L1 must be [batch, H/32*W/32], scores_back [batch, H/32*W/32, 16, 9]")
    if validate and int((l1 == 0).sum()) != B:          # the reference's masked assignment raises here (:160 / :209)
        raise IndexError("merge_patches: %d unmasked coarse patches but %d rows of trust_score" % (int((l1 == 0).sum()), B))
    dev = trust_score.device
    out = torch.empty((B, 144), dtype=torch.bool, device=dev)
    nws = _L().pats_merge_workspace_bytes(B, H, W, bt)
    ws = _workspace(nws, dev)
    _check(_L().pats_merge_patches(1 if merge_new else 0, B, _ptr(trust_score), H, W, bt, _ptr(l1),
                                   _ptr(if_nomatching1_L2.view(torch.uint8)), _ptr(scores_back), _ptr(out.view(torch.uint8)),
                                   _ptr(ws), nws, _stream()), "merge_patches")
    return out


def merge_patches_new(patch_num, trust_score, original_image_shape, if_nomatching1_L1, if_nomatching1_L2, scores_back,
                      validate=True):
    """SecondLayer.merge_patches_new (second_layer.py:193-240).  trust_score, if_nomatching1_L2 and
    scores_back are updated in place as in the reference; returns (if_nomatching [B,144], scores_back)."""
    out = _merge(True, patch_num, trust_score, original_image_shape, if_nomatching1_L1, if_nomatching1_L2,
                 scores_back, validate)
    return out, scores_back


def merge_patches_old(patch_num, trust_score, original_image_shape, if_nomatching1_L1, if_nomatching1_L2, scores_back,
                      validate=True):
    """SecondLayer.merge_patches_old (second_layer.py:137-191); hands back a zeroed scores_back (:191)."""
    out = _merge(False, patch_num, trust_score, original_image_shape, if_nomatching1_L1, if_nomatching1_L2,
                 scores_back, validate)
    return out, torch.zeros_like(scores_back)


def third_inputs(if_nomatching, pts, capacity=None, sync=True):
    """pats.py:53-58: (mkpts0_c [P,2], mkpts1_c [P,2], b_ids [P]) of the surviving L2 cells - the arguments
    PATS.forward passes to ThirdLayer (third_input[:, :2] * 2, third_input[:, 2:4] * 2, third_input[:, -1]).
    One host read of P (the reference's boolean-mask indexing syncs at the same point).
    sync=False makes none: returns (mkpts0_c [cap,2], mkpts1_c [cap,2], b_ids [cap], P [1] int64 DEVICE count) with
    `capacity` rows (default B*144), only the first min(P, cap) written."""
    f = _as_flags(if_nomatching, "if_nomatching")
    B = f.shape[0]
    p = _dev(pts, "pts").reshape(B, 144, 2)
    if f.numel() != B * 144:
        raise RuntimeError("third_inputs: if_nomatching must be [B,144]")
    dev = p.device
    cap = B * 144 if capacity is None else int(capacity)
    mk0 = torch.empty((cap, 2), dtype=torch.float32, device=dev)
    mk1 = torch.empty((cap, 2), dtype=torch.float32, device=dev)
    bi = torch.empty((cap,), dtype=torch.int64, device=dev)
    cnt = torch.empty((1,), dtype=torch.int64, device=dev)                  # always written by the scan
    nws = _L().pats_compact_workspace_bytes(B * 144)
    ws = _workspace(nws, dev)
    _check(_L().pats_third_inputs_f32(_ptr(f), _ptr(p), B, _ptr(mk0), _ptr(mk1), _ptr(bi), cap, _ptr(cnt), _ptr(ws), nws,
                                      _stream()), "third_inputs")
    if not sync:
        return mk0, mk1, bi, cnt
    P = int(cnt.item())
    return mk0[:P], mk1[:P], bi[:P]


def refine_scatter(if_nomatching, pts, mkpts1_f, label):
    """pats.py:59-67: (if_nomatching16 [B,2304] bool, pts16 [B,2304,2]) on the 48x48 sub-cell grid.
    `label` is ThirdLayer's [P*16,2] tensor (column 0 is read) or a 1-D [P*16] tensor."""
    f = _as_flags(if_nomatching, "if_nomatching")
    B = f.shape[0]
    p = _dev(pts, "pts").reshape(B, 144, 2)
    mk = _dev(mkpts1_f, "mkpts1_f").reshape(-1, 16, 2)
    P = mk.shape[0]
    lb = _dev(label, "label")
    stride = 2 if lb.dim() == 2 else 1
    if lb.numel() != P * 16 * stride:
        raise RuntimeError("refine_scatter: label must be [P*16,2] or [P*16]")
    dev = p.device
    f16 = torch.empty((B, 2304), dtype=torch.bool, device=dev)
    p16 = torch.empty((B, 2304, 2), dtype=torch.float32, device=dev)
    nws = _L().pats_compact_workspace_bytes(B * 144)
    ws = _workspace(nws, dev)
    _check(_L().pats_refine_scatter_f32(_ptr(f), _ptr(p), _ptr(mk), _ptr(lb), stride, B, P, _ptr(f16.view(torch.uint8)),
                                        _ptr(p16), _ptr(ws), nws, _stream()), "refine_scatter")
    return f16, p16


def get_result(batch_size, if_nomatching, average_point, scale, patch_size, left_choice, layer_num=2, validate=True,
               sync=True):
    """utils.get_result (utils.py:189-213) for the two-level call of pats.py:73: returns (matches_l,
    matches_r) [M,2].  scale[1] may be the reference's [K,n1,2] tensor or a [K,2] / [K,1,2] tensor holding
    one scale per row (what pats.py:70 repeats over the sub-cells).
    sync=False makes no host read: returns (matches_l [cap,2], matches_r [cap,2], M [1] int64 DEVICE count),
    only the first M rows written (throughput mode; the reference reads M when it masks)."""
    if layer_num != 2 or len(if_nomatching) != 2:
        raise RuntimeError("get_result: only the reference's layer_num=2 call is implemented")
    f0, f1 = _as_flags(if_nomatching[0], "if_nomatching[0]"), _as_flags(if_nomatching[1], "if_nomatching[1]")
    z0, z1 = [int(v) for v in patch_size[0]], [int(v) for v in patch_size[1]]
    n0, n1 = z0[1] * z0[2], z1[1] * z1[2]
    bs, rows1 = int(batch_size), f1.shape[0]
    if f0.numel() != bs * n0 or f1.numel() != rows1 * n1:
        raise RuntimeError("get_result: if_nomatching shapes do not match patch_size")
    a0, a1 = _dev(average_point[0], "average_point[0]"), _dev(average_point[1], "average_point[1]")
    s0, s1 = _dev(scale[0], "scale[0]"), _dev(scale[1], "scale[1]")
    if a0.numel() != bs * n0 * 2 or s0.numel() != bs * n0 * 2 or a1.numel() != rows1 * n1 * 2:
        raise RuntimeError("get_result: average_point / scale shapes do not match")
    if s1.numel() == rows1 * n1 * 2:
        stride = 2
    elif s1.numel() == rows1 * 2:
        stride = 0
    else:
        raise RuntimeError("get_result: scale[1] must be [K,n1,2] or [K,2]")
    c0, c1 = _as_flags(left_choice[0], "left_choice[0]"), _as_flags(left_choice[1], "left_choice[1]")
    if c0.numel() != bs or c1.numel() != rows1:
        raise RuntimeError("get_result: left_choice shapes do not match")
    if validate and int((f0 == 0).sum()) != rows1:
        raise IndexError("get_result: %d surviving level-0 cells but %d level-1 rows" % (int((f0 == 0).sum()), rows1))
    dev = a0.device
    cap = rows1 * n1
    ml = torch.empty((cap, 2), dtype=torch.float32, device=dev)
    mr = torch.empty((cap, 2), dtype=torch.float32, device=dev)
    cnt = torch.empty((1,), dtype=torch.int64, device=dev)                  # always written by the scan
    nws = _L().pats_get_result_workspace_bytes(bs * n0, rows1, n1)
    ws = _workspace(nws, dev)
    ps0, ps1 = (ctypes.c_int * 3)(*z0), (ctypes.c_int * 3)(*z1)
    _check(_L().pats_get_result_f32(bs, _ptr(f0), _ptr(f1), rows1, _ptr(a0), _ptr(a1), _ptr(s0), _ptr(s1), stride, ps0, ps1,
                                    _ptr(c0), _ptr(c1), _ptr(ml), _ptr(mr), cap, _ptr(cnt), _ptr(ws), nws, _stream()),
           "get_result")
    if not sync:
        return ml, mr, cnt
    M = int(cnt.item())
    return ml[:M], mr[:M]


# ------------------------------------------------------------------------------------------------
# throughput mode: the chunk loop of PATS.forward for a batch of pairs, no host read (csrc/batch.hip)
# ------------------------------------------------------------------------------------------------
class ChunkRows:
    """The row table pats_chunk_rows_device builds (include/pats_amd.h): device tensors only."""
    __slots__ = ("pairs", "h", "w", "Cmax", "rows_cap", "sum_cycle", "cycle_num", "second", "third", "masks", "chunk_base",
                 "crop_base", "row_cell", "row_forced", "row_crop", "row_slot", "status")


def max_chunks(height, width, max_once_used):
    """Upper bound of split_patches' chunk count: a chunk closes when the cumulative match count passes a multiple of
    max_once_used (utils.py:157), at most once per grid row."""
    return min(height + 1, (height * width - 1) // int(max_once_used) + 1)


def chunk_rows(if_nomatching1, height, width, max_once_used, Cmax=None, rows_cap=None):
    """first_layer.py:130-146 + pats.py:33-39 for a batch of pairs on the device: cumulative match counts, chunk plans,
    chunk masks and the fine level's row table in (chunk, pair, cell) order.  if_nomatching1 [pairs, h*w] bool."""
    f = _as_flags(if_nomatching1, "if_nomatching1")
    pairs, N = f.shape[0], height * width
    if f.numel() != pairs * N:
        raise RuntimeError("chunk_rows: if_nomatching1 must be [pairs, height*width]")
    r = ChunkRows()
    r.pairs, r.h, r.w = pairs, int(height), int(width)
    r.Cmax = max_chunks(height, width, max_once_used) if Cmax is None else int(Cmax)
    r.rows_cap = pairs * (N + (r.Cmax - 1) * width) if rows_cap is None else int(rows_cap)
    dev = f.device
    i32, i64, u8 = torch.int32, torch.int64, torch.uint8
    r.sum_cycle = torch.empty((pairs, N), dtype=i32, device=dev)
    r.cycle_num = torch.empty((pairs,), dtype=i32, device=dev)
    r.second = torch.empty((pairs, height + 1, 2), dtype=i64, device=dev)
    r.third = torch.empty((pairs, height + 1, 2), dtype=i64, device=dev)
    r.masks = torch.empty((r.Cmax, pairs, N), dtype=torch.bool, device=dev)
    r.chunk_base = torch.empty((r.Cmax + 1,), dtype=i64, device=dev)
    r.crop_base = torch.empty((pairs + 1,), dtype=i64, device=dev)
    r.row_cell = torch.empty((r.rows_cap,), dtype=i32, device=dev)
    r.row_forced = torch.empty((r.rows_cap,), dtype=u8, device=dev)
    r.row_crop = torch.empty((r.rows_cap,), dtype=i32, device=dev)
    r.row_slot = torch.empty((r.Cmax, pairs * N), dtype=i32, device=dev)
    r.status = torch.empty((1,), dtype=i32, device=dev)
    nws = _L().pats_chunk_rows_workspace_bytes(pairs, r.Cmax)
    ws = _workspace(nws, dev)
    _check(_L().pats_chunk_rows_device(_ptr(f), pairs, int(height), int(width), int(max_once_used), r.Cmax, r.rows_cap,
                                       _ptr(r.sum_cycle), _ptr(r.cycle_num), _ptr(r.second), _ptr(r.third),
                                       _ptr(r.masks.view(u8)), _ptr(r.chunk_base), _ptr(r.crop_base), _ptr(r.row_cell),
                                       _ptr(r.row_forced), _ptr(r.row_crop), _ptr(r.row_slot), _ptr(r.status), _ptr(ws), nws,
                                       _stream()), "chunk_rows")
    return r


def merge_patches_batch(merge_new, rows, trust_score, original_image_shape, if_nomatching1_L2, scores_back=None):
    """SecondLayer.merge_patches_new / _old (second_layer.py:137-238) for every chunk of every pair of a ChunkRows table,
    chunk blocks in order, pats.py:38-39 applied.  trust_score / if_nomatching1_L2 [rows_cap,144] are updated in place;
    scores_back [pairs, N, 16, 9] float64 (zeros if None, pats.py:32).  Returns if_nomatching [rows_cap,144] bool."""
    if trust_score.dtype != torch.float32 or not trust_score.is_cuda or not trust_score.is_contiguous():
        raise RuntimeError("merge_patches_batch: trust_score must be a contiguous float32 GPU tensor (it is updated in place)")
    if if_nomatching1_L2.dtype != torch.bool or not if_nomatching1_L2.is_contiguous():
        raise RuntimeError("merge_patches_batch: if_nomatching1_L2 must be a contiguous bool tensor (it is updated in place)")
    H, W = int(original_image_shape[0]), int(original_image_shape[1])
    if trust_score.numel() != rows.rows_cap * 144 or if_nomatching1_L2.numel() != rows.rows_cap * 144 or \
            H // 32 != rows.h or W // 32 != rows.w:
        raise RuntimeError("merge_patches_batch: tensors must be [rows_cap,144] on the table's grid")
    dev = trust_score.device
    fresh = scores_back is None
    if fresh:
        scores_back = torch.empty((rows.pairs, rows.h * rows.w, 16, 9), dtype=torch.float64, device=dev)   # cleared by the call
    elif scores_back.dtype != torch.float64 or not scores_back.is_contiguous() or \
            scores_back.numel() != rows.pairs * rows.h * rows.w * 144:
        raise RuntimeError("merge_patches_batch: scores_back must be a contiguous float64 [pairs, N, 16, 9] tensor")
    out = torch.empty((rows.rows_cap, 144), dtype=torch.bool, device=dev)
    nws = _L().pats_merge_batch_workspace_bytes(rows.pairs, H, W)
    ws = _workspace(nws, dev)
    _check(_L().pats_merge_patches_batch(1 if merge_new else 0, rows.Cmax, rows.pairs, H, W, rows.rows_cap,
                                         _ptr(rows.chunk_base), _ptr(rows.row_cell), _ptr(rows.row_slot), _ptr(rows.row_forced),
                                         _ptr(trust_score), _ptr(if_nomatching1_L2.view(torch.uint8)), _ptr(scores_back),
                                         int(fresh), _ptr(out.view(torch.uint8)), _ptr(ws), nws, _stream()), "merge_patches_batch")
    return out


def merge_patches_chunk(merge_new, rows, c, row_origin, trust_score, original_image_shape, if_nomatching1_L2, scores_back,
                        first=False):
    """merge_patches_new / _old for chunk `c` of a ChunkRows table on that chunk's own tensors (trust_score / if_nomatching1_L2
    [B,144] = table rows row_origin .. row_origin + B, updated in place): PATS.forward's chunk loop walked chunk by chunk
    (pats.py:33-39) with one launch per chunk and no host read.  scores_back [pairs,N,16,9] float64 is handed from call to
    call (first=True clears it, pats.py:32).  Returns if_nomatching [B,144] bool (pats.py:38-39 applied)."""
    B = trust_score.shape[0]
    H, W = int(original_image_shape[0]), int(original_image_shape[1])
    if trust_score.dtype != torch.float32 or not trust_score.is_contiguous() or if_nomatching1_L2.dtype != torch.bool or \
            not if_nomatching1_L2.is_contiguous() or trust_score.numel() != B * 144 or if_nomatching1_L2.numel() != B * 144:
        raise RuntimeError("merge_patches_chunk: trust_score / if_nomatching1_L2 must be contiguous [B,144] float32 / bool")
    if scores_back.dtype != torch.float64 or not scores_back.is_contiguous() or scores_back.numel() != rows.pairs * rows.h * rows.w * 144:
        raise RuntimeError("merge_patches_chunk: scores_back must be a contiguous float64 [pairs, N, 16, 9] tensor")
    dev = trust_score.device
    out = torch.empty((B, 144), dtype=torch.bool, device=dev)
    nws = _L().pats_merge_batch_workspace_bytes(rows.pairs, H, W)
    ws = _workspace(nws, dev)
    _check(_L().pats_merge_patches_chunks(1 if merge_new else 0, rows.Cmax, int(c), int(c) + 1, rows.pairs, H, W, int(row_origin), B,
                                          _ptr(rows.chunk_base), _ptr(rows.row_cell), _ptr(rows.row_slot), _ptr(rows.row_forced),
                                          _ptr(trust_score), _ptr(if_nomatching1_L2.view(torch.uint8)), _ptr(scores_back),
                                          int(bool(first)), _ptr(out.view(torch.uint8)), _ptr(ws), nws, _stream()),
           "merge_patches_chunk")
    return out


def _carve(device, *sizes):
    """One allocation for several internal tensors of a fused call: returns (arena, [device addresses])."""
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (int(n) + 255) & ~255
    arena = torch.empty(max(total, 1), dtype=torch.uint8, device=device)
    base = arena.data_ptr()
    return arena, [ctypes.c_void_p(base + o) for o in offs]


def chunk_fine_tail(f0, f1, one, ns, scale_x, scale_y, iters, bias_k, merge_new, rows, c, row_origin, image_shape, scores_back, first,
                    wait_before_merge=None, record_after_merge=None):
    """Everything between the fine network callback and the third one for ONE chunk of a pair walked chunk by chunk, in one C
    call (pats_chunk_fine_tail_f32, csrc/chunk_walk.cpp): second_layer.py:100-122 (cost build, log_optimal_transport2 + ln k,
    est_position, the chunk's merge through the row table `rows`) and pats.py:37-39,53-58 (tail rows, the third level's inputs over
    the capacity 144 B).  Returns (merged [B,144] bool, points [B,144,2], mkpts0_c [144 B,2], mkpts1_c, b_ids [144 B], P [1] int64
    on the device); the log plan, flags and the other expansion outputs stay internal.
    wait_before_merge / record_after_merge: torch.cuda.Event objects (already recorded once, so that their handles exist) - the
    stream waits for the first right before the merge and re-records the second right behind it."""
    d0, d1 = _dev(f0, "f0"), _dev(f1, "f1")
    B = d0.shape[0]
    if tuple(d0.shape) != (B, 264, 145) or d1.shape != d0.shape:
        raise RuntimeError("chunk_fine_tail: descriptors must be [B,264,145]")
    nsv, sx, sy = _dev(ns, "ns").reshape(B, 144), _dev(scale_x, "scale_x").reshape(B, 144), _dev(scale_y, "scale_y").reshape(B, 144)
    H, W = int(image_shape[0]), int(image_shape[1])
    dev = d0.device
    merged = torch.empty((B, 144), dtype=torch.bool, device=dev)
    pts = torch.empty((B, 144, 2), dtype=torch.float32, device=dev)
    mk = torch.empty((2, B * 144, 2), dtype=torch.float32, device=dev)
    bi = torch.empty((B * 144,), dtype=torch.int64, device=dev)
    cnt = torch.empty((1,), dtype=torch.int64, device=dev)
    nws = _L().pats_chunk_fine_tail_workspace_bytes(B, rows.pairs, H, W)
    n1 = B * 144
    arena, (Z, cflag, trust, core, xs, ys, bound, rflag, ws) = _carve(dev, B * 145 * 145 * 4, n1, n1 * 4, n1 * 4, n1 * 4, n1 * 4, n1 * 4 * 8,
                                                                      n1, nws)
    _check(_L().pats_chunk_fine_tail_f32(_ptr(d0), _ptr(d1), B, _ptr(_scalar_dev(one, dev)), _ptr(nsv), int(iters), float(bias_k), _ptr(sx),
                                         _ptr(sy), 1 if merge_new else 0, rows.Cmax, int(c), rows.pairs, H, W, int(row_origin),
                                         _ptr(rows.chunk_base), _ptr(rows.row_cell), _ptr(rows.row_slot), _ptr(rows.row_forced),
                                         _ptr(scores_back), int(bool(first)), Z, cflag, trust, core, _ptr(pts), xs, ys, bound, rflag,
                                         _ptr(merged.view(torch.uint8)), _ptr(mk[0]), _ptr(mk[1]), _ptr(bi), _ptr(cnt),
                                         ctypes.c_void_p(wait_before_merge.cuda_event if wait_before_merge is not None else 0),
                                         ctypes.c_void_p(record_after_merge.cuda_event if record_after_merge is not None else 0),
                                         ws, nws, _stream()),
           "chunk_fine_tail")
    return merged, pts, mk[0], mk[1], bi, cnt


def chunk_third_tail(feat0, feat1, P, scale, p_s, p_t, iters, outdoor, merged, points, chunk_mask, h, w, pts_new, scales):
    """Everything behind the third network callback for ONE chunk, in one C call (pats_chunk_third_tail_f32): third_layer.py:153-170
    over the capacity 144 B with the count P on the device, pats.py:59-67 (scatter onto the sub-cell grid) and :68-78 (get_result with
    the chunk's mask [h w] as the level-0 flags; pts_new / scales = Compute_imgs' [1, h w, 2] tensors).  Returns (matches_l, matches_r
    [2304 B, 2], M [1] int64 on the device): the first M rows are the chunk's matches in the reference's order."""
    a, b = _dev(feat0, "feat0"), _dev(feat1, "feat1")
    Pc = a.shape[0]
    B = merged.shape[0]
    if tuple(a.shape) != (Pc, 128, 65) or b.shape != a.shape or Pc != B * 144:
        raise RuntimeError("chunk_third_tail: descriptors must be [144 B,128,65]")
    sc = _dev(scale, "scale").reshape(Pc, 64)
    ps = _dev(p_s.to(torch.int64), "p_s", torch.int64).reshape(Pc, 2)
    pt = _dev(p_t.to(torch.int64), "p_t", torch.int64).reshape(Pc, 2)
    dev = a.device
    ml = torch.empty((B * 2304, 2), dtype=torch.float32, device=dev)
    mr = torch.empty((B * 2304, 2), dtype=torch.float32, device=dev)
    cnt = torch.empty((1,), dtype=torch.int64, device=dev)
    nws = _L().pats_chunk_third_tail_workspace_bytes(B, int(h), int(w))
    arena, (m0, m1, label, ifm, f16, p16, mrow, ws) = _carve(dev, Pc * 32 * 4, Pc * 32 * 4, Pc * 32 * 4, Pc * 16, B * 2304, B * 2304 * 8,
                                                             B * 2304 * 4, nws)
    _check(_L().pats_chunk_third_tail_f32(_ptr(a), _ptr(b), Pc, _ptr(_dev(P, "P", torch.int64)), _ptr(sc), _ptr(ps), _ptr(pt), int(iters),
                                          int(bool(outdoor)), _ptr(_as_flags(merged, "merged")), _ptr(_dev(points, "points")), B,
                                          _ptr(_as_flags(chunk_mask, "chunk_mask")), int(h), int(w), _ptr(_dev(pts_new, "pts_new")),
                                          _ptr(_dev(scales, "scales")), _ptr(_ones(B, dev)), m0, m1, label, ifm, f16, p16, _ptr(ml), _ptr(mr),
                                          mrow, _ptr(cnt), ws, nws, _stream()), "chunk_third_tail")
    return ml, mr, cnt


_ONES = {}


def _ones(n, device):
    key = (int(n), str(device))
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones((int(n),), dtype=torch.uint8, device=device)
    return t


def get_result_chunks(rows, if_nomatching16, pts_new, pts16, scales, patch_size=((32, None, None), (2, 48, 48))):
    """get_result (utils.py:189-213) for every (chunk, pair) of a ChunkRows table in one call, as PATS.forward issues it per
    chunk (pats.py:68-73): pts_new / scales = Compute_imgs' per-pair [pairs,N,2] tensors, pts16 [rows_cap,2304,2] /
    if_nomatching16 [rows_cap,2304] from refine_scatter.  No host read.  Returns (matches_l [cap,2], matches_r [cap,2],
    match_row [cap] int32, M [1] int64 device): the first M rows are valid, match_row -> rows.row_cell // N = pair."""
    f16 = _as_flags(if_nomatching16, "if_nomatching16")
    z0 = [int(patch_size[0][0]), rows.h, rows.w]
    z1 = [int(v) for v in patch_size[1]]
    n1 = z1[1] * z1[2]
    if f16.numel() != rows.rows_cap * n1:
        raise RuntimeError("get_result_chunks: if_nomatching16 must be [rows_cap, %d]" % n1)
    a0, a1, s0 = _dev(pts_new, "pts_new"), _dev(pts16, "pts16"), _dev(scales, "scales")
    N = rows.h * rows.w
    if a0.numel() != rows.pairs * N * 2 or s0.numel() != rows.pairs * N * 2 or a1.numel() != rows.rows_cap * n1 * 2:
        raise RuntimeError("get_result_chunks: pts_new / scales must be [pairs,N,2], pts16 [rows_cap,%d,2]" % n1)
    dev = a0.device
    cap = rows.rows_cap * n1
    ml = torch.empty((cap, 2), dtype=torch.float32, device=dev)
    mr = torch.empty((cap, 2), dtype=torch.float32, device=dev)
    mrow = torch.empty((cap,), dtype=torch.int32, device=dev)
    cnt = torch.empty((1,), dtype=torch.int64, device=dev)
    nws = _L().pats_get_result_workspace_bytes(rows.Cmax * rows.pairs * N, rows.rows_cap, n1)
    ws = _workspace(nws, dev)
    ps0, ps1 = (ctypes.c_int * 3)(*z0), (ctypes.c_int * 3)(*z1)
    _check(_L().pats_get_result_chunks_f32(rows.Cmax, rows.pairs, _ptr(rows.masks.view(torch.uint8)), _ptr(f16), rows.rows_cap,
                                           _ptr(a0), _ptr(a1), _ptr(s0), ps0, ps1, _ptr(_ones(rows.Cmax * rows.pairs, dev)),
                                           _ptr(_ones(rows.rows_cap, dev)), _ptr(ml), _ptr(mr), _ptr(mrow), cap, _ptr(cnt),
                                           _ptr(ws), nws, _stream()), "get_result_chunks")
    return ml, mr, mrow, cnt


def matches_by_pair(rows, matches_l, matches_r, match_row, M, out=None, P=None):
    """The matches of a batch (get_result_chunks) grouped by pair ON THE DEVICE: (matches_l, matches_r) with every pair's
    list contiguous in the reference's order, and pair_off [pairs + 1] int64 (pair p = rows pair_off[p] .. pair_off[p + 1]).
    What batch.split_by_pair reads back is then only pair_off.  out: optional (out_l, out_r, pair_off) to write into.
    P (device int64 [1], the step's third-level problem count): pair_off is then the first pairs + 1 entries of a pairs + 4
    buffer whose tail is (M, P, table status) - returned as a fourth element: the whole hand-over in ONE device-to-host copy."""
    dev = matches_l.device
    n_off = rows.pairs + 1 + (3 if P is not None else 0)
    if out is None:
        out = (torch.empty_like(matches_l), torch.empty_like(matches_r), torch.empty((n_off,), dtype=torch.int64, device=dev))
    ol, orr, off = out[:3]
    if off.numel() != n_off or not off.is_contiguous():
        raise RuntimeError("matches_by_pair: pair_off must be a contiguous int64 tensor of %d entries" % n_off)
    nws = _L().pats_matches_by_pair_workspace_bytes(rows.Cmax, rows.pairs)
    ws = _workspace(nws, dev)
    if P is None:
        _check(_L().pats_matches_by_pair_f32(_ptr(matches_l), _ptr(matches_r), _ptr(match_row), _ptr(M), _ptr(rows.row_cell),
                                             _ptr(rows.chunk_base), rows.Cmax, rows.pairs, rows.h * rows.w, _ptr(ol), _ptr(orr), _ptr(off),
                                             _ptr(ws), nws, _stream()), "matches_by_pair")
        return ol, orr, off
    _check(_L().pats_matches_by_pair_summary_f32(_ptr(matches_l), _ptr(matches_r), _ptr(match_row), _ptr(M), _ptr(rows.row_cell),
                                                 _ptr(rows.chunk_base), rows.Cmax, rows.pairs, rows.h * rows.w, _ptr(ol), _ptr(orr),
                                                 _ptr(off), _ptr(_dev(P, "P", torch.int64)), _ptr(rows.status), _ptr(ws), nws, _stream()),
           "matches_by_pair")
    return ol, orr, off[:rows.pairs + 1], off


def masked_stream(cus):
    """A torch stream (torch.cuda.ExternalStream over a HIP stream of this library) whose kernels may only run on the
    compute units listed in `cus` (indices 0..255 on MI355X) - see pats_stream_create_cu_mask.  The stream lives as long as
    the process (it is not destroyed: torch may still hold references to it)."""
    n_cu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for c in cus:
        if not 0 <= int(c) < n_cu:
            raise RuntimeError("masked_stream: CU %d outside 0..%d" % (c, n_cu - 1))
        mask[int(c) // 32] |= 1 << (int(c) % 32)
    handle = ctypes.c_void_p()
    _check(_L().pats_stream_create_cu_mask(mask, words, ctypes.byref(handle)), "masked_stream")
    return torch.cuda.ExternalStream(handle.value)


def profile_marker(tag=0):
    """An empty, uniquely named kernel on the current stream: brackets a region of a kernel trace."""
    _check(_L().pats_profile_marker(int(tag), _stream()), "profile_marker")


def attention(query, key, value, return_prob=True):
    """models/modules.py:84-88: returns (x [b,dim,heads,n], prob [b,heads,n,m]) like the reference;
    return_prob=False skips materialising prob (MultiHeadedAttention discards it, :103) and returns
    (x, None)."""
    q, k, v = _dev(query, "query"), _dev(key, "key"), _dev(value, "value")
    if q.dim() != 4 or k.dim() != 4 or v.shape != k.shape or k.shape[:3] != q.shape[:3]:
        raise RuntimeError("attention: query [b,dim,heads,n], key/value [b,dim,heads,m]")
    b, dim, heads, n = q.shape
    m = k.shape[3]
    out = torch.empty((b, dim, heads, n), dtype=torch.float32, device=q.device)
    prob = torch.empty((b, heads, n, m), dtype=torch.float32, device=q.device) if return_prob else None
    _check(_L().pats_attention_f32(_ptr(q), _ptr(k), _ptr(v), b, dim, heads, n, m, _ptr(out), _ptr(prob), _stream()),
           "attention")
    return out, prob


# ------------------------------------------------------------------------------------------------
# the GNN layer around the attention core
# ------------------------------------------------------------------------------------------------
class _PropagationWeights(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("wq_t", "bq", "wk_t", "bk", "wv_t", "bv", "wm_t", "bm", "w1_t", "b1",
                                               "bn_a", "bn_b", "w2_t", "b2")]


class PropagationParams:
    """Device-resident weights of one AttentionalPropagation layer (modules.py:107-113) in the layout the C-ABI takes
    (Conv1d matrices transposed, BatchNorm folded for eval mode).  Build it once per layer:
        PropagationParams(layer.state_dict())          # a reference module's own parameters, or any dict with its names
    """

    def __init__(self, state, device="cuda", eps=1e-5, prefix=""):
        def get(name):
            t = state[prefix + name]
            t = torch.from_numpy(np.asarray(t)) if not isinstance(t, torch.Tensor) else t
            return t.detach().to(device=device, dtype=torch.float32)

        def mat_t(name):                              # [C_out, C_in, 1] -> [C_in][C_out]
            w = get(name)
            return w.reshape(w.shape[0], -1).t().contiguous()
        self.eps = float(eps)
        self._packed = {}
        self.C = get("attn.merge.bias").shape[0]
        self.t = {"wq_t": mat_t("attn.proj.0.weight"), "bq": get("attn.proj.0.bias").contiguous(),
                  "wk_t": mat_t("attn.proj.1.weight"), "bk": get("attn.proj.1.bias").contiguous(),
                  "wv_t": mat_t("attn.proj.2.weight"), "bv": get("attn.proj.2.bias").contiguous(),
                  "wm_t": mat_t("attn.merge.weight"), "bm": get("attn.merge.bias").contiguous(),
                  "w1_t": mat_t("mlp.0.weight"), "b1": get("mlp.0.bias").contiguous(),
                  "w2_t": mat_t("mlp.3.weight"), "b2": get("mlp.3.bias").contiguous()}
        gamma, beta = get("mlp.1.weight"), get("mlp.1.bias")
        scale = gamma / torch.sqrt(get("mlp.1.running_var") + self.eps)
        self.bn = {False: (scale.contiguous(), (beta - get("mlp.1.running_mean") * scale).contiguous()),   # eval: folded
                   True: (gamma.contiguous(), beta.contiguous())}                                       # train: gamma / beta

    def packed(self, heads=4):
        """The Conv1d matrices split into fp16 hi + lo halves in MFMA fragment order (pats_propagation_pack_f32), made once and
        kept beside the weights; None if the library has no packed form for this shape."""
        key = int(heads)
        if key not in self._packed:
            nb = _L().pats_propagation_packed_bytes(self.C, key)
            buf = None
            if nb:
                buf = torch.empty((nb,), dtype=torch.uint8, device=self.t["wq_t"].device)
                w = self.struct(False)
                with torch.cuda.device(buf.device):
                    _check(_L().pats_propagation_pack_f32(ctypes.byref(w), self.C, key, _ptr(buf), nb, _stream()), "propagation_pack")
            self._packed[key] = buf
        return self._packed[key]

    def struct(self, bn_train):
        a, b = self.bn[bool(bn_train)]
        w = _PropagationWeights()
        for k, v in self.t.items():
            setattr(w, k, v.data_ptr())
        w.bn_a, w.bn_b = a.data_ptr(), b.data_ptr()
        return w


def attentional_propagation(x, source, params, heads=4, bn_train=False, residual=None, count=None, out=None, count_off=0):
    """AttentionalPropagation.forward(x, source) (modules.py:114-117) -> delta [b,C,n]; with `residual` (= x in
    AttentionalGNN.forward, :131-133) the sum residual + delta.  bn_train: BatchNorm on batch statistics - what the
    third layer's GNN does under PATS.eval() (pats.py:112-120).  Six GEMM launches + the attention kernel."""
    x, source = _dev(x, "x"), _dev(source, "source")
    b, C, n = x.shape
    m = source.shape[2]
    if source.shape[0] != b or source.shape[1] != C or C != params.C:
        raise RuntimeError("attentional_propagation: x %s / source %s / weights C=%d do not match"
                           % (tuple(x.shape), tuple(source.shape), params.C))
    res = _dev(residual, "residual") if residual is not None else None
    if out is None:
        out = torch.empty_like(x)
    elif tuple(out.shape) != tuple(x.shape) or not out.is_contiguous() or out.dtype != torch.float32 or not out.is_cuda:
        raise RuntimeError("attentional_propagation: out must be a contiguous float32 GPU tensor shaped like x")
    nb = _L().pats_attentional_propagation_workspace_bytes(b, C, n, m)
    ws = _workspace(nb, x.device)
    w = params.struct(bn_train)
    pk = params.packed(heads)      # the third level's shape: one fused kernel (gnn_fused.hip); any other: packed-weights convolutions
    if pk is not None and count is not None:
        _check(_L().pats_attentional_propagation_packed_counted_f32(_ptr(x), _ptr(source), b, _ptr(_dev(count, "count", torch.int64)),
                                                                    int(count_off), C,
                                                                    int(heads), n, m, ctypes.byref(w), _ptr(pk), int(bool(bn_train)),
                                                                    float(params.eps), _ptr(res), _ptr(out), _ptr(ws), nb, _stream()),
               "attentional_propagation")
        return out
    if pk is not None:
        _check(_L().pats_attentional_propagation_packed_f32(_ptr(x), _ptr(source), b, C, int(heads), n, m, ctypes.byref(w), _ptr(pk),
                                                            int(bool(bn_train)), float(params.eps), _ptr(res), _ptr(out), _ptr(ws),
                                                            nb, _stream()), "attentional_propagation")
        return out
    _check(_L().pats_attentional_propagation_f32(_ptr(x), _ptr(source), b, C, int(heads), n, m, ctypes.byref(w),
                                                 int(bool(bn_train)), float(params.eps), _ptr(res), _ptr(out), _ptr(ws), nb,
                                                 _stream()), "attentional_propagation")
    return out


UNSUPPORTED = 2            # include/pats_amd.h PATS_ERR_UNSUPPORTED
GNN_STACK_ROWS = 4096     # rows of a descriptor set per pats_attentional_gnn_packed_f32 call (its workspace is ~1.9 MB a row)


def _gnn_packed_stack(desc0, desc1, layers, names, heads, count=None, out=None):
    """The whole stack in the fine level's one-kernel form (csrc/gnn_fine.hip), or None if the library has no such form for
    this shape (anything but [b, 264, 145], 4 heads)."""
    b, C, n = desc0.shape
    if tuple(desc1.shape) != (b, C, n) or b == 0 or not _L().pats_attentional_gnn_packed_workspace_bytes(1, C, int(heads), n):
        return None
    packed = [p.packed(heads) for p in layers]
    if any(pk is None for pk in packed) or any(p.C != C for p in layers):
        return None
    L = len(layers)
    structs = [p.struct(False) for p in layers]
    w_arr = (ctypes.c_void_p * max(L, 1))(*[ctypes.addressof(w) for w in structs])
    pk_arr = (ctypes.c_void_p * max(L, 1))(*[pk.data_ptr() for pk in packed])
    cross = (ctypes.c_int * max(L, 1))(*[1 if nm == "cross" else 0 for nm in names])
    out0, out1 = out if out is not None else (torch.empty_like(desc0), torch.empty_like(desc1))
    for lo in range(0, b, GNN_STACK_ROWS):
        hi = min(b, lo + GNN_STACK_ROWS)
        nb = _L().pats_attentional_gnn_packed_workspace_bytes(hi - lo, C, int(heads), n)
        ws = _workspace(nb, desc0.device)
        rc = _L().pats_attentional_gnn_packed_f32(_ptr(desc0[lo:hi]), _ptr(desc1[lo:hi]), hi - lo, _ptr(count), lo, C, int(heads), n, L, w_arr,
                                                  pk_arr, cross,
                                                  float(layers[0].eps) if L else 1e-5, _ptr(out0[lo:hi]), _ptr(out1[lo:hi]), _ptr(ws), nb,
                                                  _stream())
        if rc == UNSUPPORTED:
            return None
        _check(rc, "attentional_gnn_packed")
    return out0, out1


GNN_LAYER_ROWS = 32768    # rows per pass of the layer-by-layer path over a big batch (a layer's workspace is ~7 tensors)


def attentional_gnn(desc0, desc1, layers, names, heads=4, bn_train=False, count=None, out=None):
    """AttentionalGNN.forward (modules.py:127-134): layers = [PropagationParams, ...], names = ['self', 'cross', ...].
    At the fine level's shape ([b, 264, 145], eval-mode BatchNorm) the whole stack runs in the one-kernel layer's own descriptor
    form (pats_attentional_gnn_packed_f32); any other shape: layer by layer (eval mode: in blocks of GNN_LAYER_ROWS rows - rows are
    independent problems, a cross layer couples row i of one set with row i of the other only).
    count: optional device int64 [1] - only rows < count are problems (throughput mode: the launches cover a capacity); honoured
    by the one-kernel layers (fine and third level's shapes, eval mode), rows past it are zeros (fine) / left untouched (third).
    out: optional (out0, out1), contiguous float32 GPU tensors shaped like the inputs."""
    layers, names = list(layers), list(names)
    desc0, desc1 = _dev(desc0, "desc0"), _dev(desc1, "desc1")
    if out is not None and (len(out) != 2 or any(tuple(o.shape) != tuple(desc0.shape) or not o.is_contiguous() for o in out)):
        raise RuntimeError("attentional_gnn: out must be two contiguous tensors shaped like the descriptors")
    if count is not None:
        count = _dev(count, "count", torch.int64)
    if not bn_train and len(layers) == len(names):
        got = _gnn_packed_stack(desc0, desc1, layers, names, heads, count, out)
        if got is not None:
            return got
    b = desc0.shape[0]
    if len(layers) == 0:
        if out is not None:
            out[0].copy_(desc0)
            out[1].copy_(desc1)
            return out[0], out[1]
        return desc0, desc1
    # batch statistics couple all rows of a launch: no blocking there
    step = b if (bn_train or b <= GNN_LAYER_ROWS) else GNN_LAYER_ROWS
    if not bn_train and tuple(desc0.shape[1:]) == (264, 145):
        step = min(step, 2 * GNN_STACK_ROWS)     # (the fine level's kernels keep 0.5 MB of projections per problem in the workspace)
    if step == b and out is None:
        res = None
    else:
        res = out if out is not None else (torch.empty_like(desc0), torch.empty_like(desc1))
    # Eval-mode BatchNorm and no device-side count: a layer treats every problem on its own and both descriptor sets go through the SAME
    # weights, so the two sets run as ONE batch of 2 b problems - one launch chain a layer instead of two ('cross': the sources are the
    # sets swapped).  Same kernels, same per-problem arithmetic: the same bits; at the coarse level of one pair (2 x [448, 300]) the
    # 18 layers were 432 launches of ~23 us for almost no work (6 of a pair's 26.6 ms with the heads inside).
    if not bn_train and count is None:
        for lo in range(0, max(b, 1), max(step, 1)):
            hi = min(b, lo + step)
            bb = hi - lo
            X = torch.cat([desc0[lo:hi], desc1[lo:hi]])
            for p, name in zip(layers, names):
                src = torch.cat([X[bb:], X[:bb]]) if name == "cross" else X
                X = attentional_propagation(X, src, p, heads, False, residual=X)
            if res is None:
                return X[:bb], X[bb:]
            res[0][lo:hi].copy_(X[:bb])
            res[1][lo:hi].copy_(X[bb:])
        return res[0], res[1]
    for lo in range(0, max(b, 1), max(step, 1)):
        hi = min(b, lo + step)
        c0, c1 = desc0[lo:hi], desc1[lo:hi]
        for li, (p, name) in enumerate(zip(layers, names)):
            src0, src1 = (c1, c0) if name == "cross" else (c0, c1)
            last = li == len(layers) - 1 and res is not None
            n0 = attentional_propagation(c0, src0, p, heads, bn_train, residual=c0, count=count, count_off=lo, out=res[0][lo:hi] if last else None)
            n1 = attentional_propagation(c1, src1, p, heads, bn_train, residual=c1, count=count, count_off=lo, out=res[1][lo:hi] if last else None)
            c0, c1 = n0, n1
        if res is None:
            return c0, c1
    return res[0], res[1]


# ------------------------------------------------------------------------------------------------
# the descriptor heads either side of the GNN: Conv1d(k=1) alone (final_proj) and chained (MLP / KeypointEncoder)
# ------------------------------------------------------------------------------------------------
def _pad8(t, dim):
    """Zero channels up to a multiple of 8 along `dim` (the contraction's operand slabs are 8 channels)."""
    k = t.shape[dim]
    if k % 8 == 0:
        return t
    shape = list(t.shape)
    shape[dim] = (-k) % 8
    return torch.cat([t, t.new_zeros(shape)], dim=dim).contiguous()


def conv1d(x, weight, bias=None, in_scale=None, in_shift=None, residual=None, weight_t=None):
    """nn.Conv1d(kernel_size=1) as the path uses it (final_proj: first_layer.py:34-36,105, second_layer.py:40-42,91;
    the layers of MLP, modules.py:57-69): x [b,K,n], weight [M,K,1] or [M,K] (the module's own layout), bias [M] or
    None -> [b,M,n].  in_scale / in_shift [K]: x is max(0, x * scale + shift) while it is staged - the BatchNorm1d + ReLU
    of the previous MLP layer, folded.  residual [b,M,n] is added to the result.
    weight_t: the weight already in the C-ABI's layout ([K8][M]: transposed, input channels zero-padded to a multiple of
    8 - MLPParams keeps it), so that no transpose / pad kernel runs per call."""
    x = _dev(x, "x")
    weight = _dev(weight, "weight")
    b, K, n = x.shape
    w2 = weight.reshape(weight.shape[0], -1)
    M = w2.shape[0]
    if w2.shape[1] != K:
        raise RuntimeError("conv1d: weight %s does not take %d input channels" % (tuple(weight.shape), K))
    if weight_t is not None:
        w_t = _dev(weight_t, "weight_t")
        if tuple(w_t.shape) != (K + (-K) % 8, M):
            raise RuntimeError("conv1d: weight_t must be [%d,%d]" % (K + (-K) % 8, M))
    else:
        w_t = _pad8(w2.t().contiguous(), 0)             # [K][M], zero rows for the padding channels
    xp = _pad8(x, 1)
    sc = sh = None
    if in_scale is not None:
        sc = _pad8(_dev(in_scale, "in_scale").reshape(-1), 0)      # padded channels: max(0, 0 * 0 + 0) = 0
        sh = _pad8(_dev(in_shift, "in_shift").reshape(-1), 0)
        if sc.numel() != xp.shape[1] or sh.numel() != xp.shape[1]:
            raise RuntimeError("conv1d: in_scale / in_shift must have %d entries" % K)
    bs = _dev(bias, "bias").reshape(-1) if bias is not None else None
    if bs is not None and bs.numel() != M:
        raise RuntimeError("conv1d: bias must have %d entries" % M)
    res = _dev(residual, "residual") if residual is not None else None
    if res is not None and tuple(res.shape) != (b, M, n):
        raise RuntimeError("conv1d: residual %s is not [%d,%d,%d]" % (tuple(res.shape), b, M, n))
    y = torch.empty((b, M, n), dtype=torch.float32, device=x.device)
    if b == 0 or n == 0:
        return y
    nb = _L().pats_conv1x1_workspace_bytes()
    ws = _workspace(nb, x.device)
    _check(_L().pats_conv1x1_f32(_ptr(w_t), _ptr(bs), _ptr(xp), b, xp.shape[1], M, n, _ptr(sc), _ptr(sh), _ptr(res), _ptr(y),
                                 _ptr(ws), nb, _stream()), "conv1d")
    return y


def bn_fold(h, gamma, beta, eps=1e-5):
    """BatchNorm1d in train mode, folded: (scale, shift) [C] each with scale = gamma / sqrt(var + eps) and
    shift = beta - mean * scale over the batch statistics of h [b,C,n] (biased variance, what F.batch_norm
    normalises with).  Feed them to the next conv1d as in_scale / in_shift."""
    h, gamma, beta = _dev(h, "h"), _dev(gamma, "gamma").reshape(-1), _dev(beta, "beta").reshape(-1)
    b, C, n = h.shape
    if gamma.numel() != C or beta.numel() != C:
        raise RuntimeError("bn_fold: gamma / beta must have %d entries" % C)
    scale = torch.empty((C,), dtype=torch.float32, device=h.device)
    shift = torch.empty_like(scale)
    nb = _L().pats_bn_fold_workspace_bytes(C)
    ws = _workspace(nb, h.device)
    _check(_L().pats_bn_fold_f32(_ptr(h), b, C, n, _ptr(gamma), _ptr(beta), float(eps), _ptr(scale), _ptr(shift), _ptr(ws), nb,
                                 _stream()), "bn_fold")
    return scale, shift


class MLPParams:
    """The parameters of one `MLP(channels)` (modules.py:57-69: Conv1d [BatchNorm1d ReLU] ... Conv1d) from the
    nn.Sequential's own state_dict ("0.weight", "0.bias", "1.weight", "1.bias", "1.running_mean", "1.running_var",
    "3.weight", ...), device-resident.  `prefix` selects a sub-module, e.g. "encoder." for a KeypointEncoder."""

    def __init__(self, state, device="cuda", eps=1e-5, prefix=""):
        def get(name):
            t = state[prefix + name]
            t = torch.from_numpy(np.asarray(t)) if not isinstance(t, torch.Tensor) else t
            return t.detach().to(device=device, dtype=torch.float32).contiguous()
        self.eps = float(eps)
        idx = sorted({int(k[len(prefix):].split(".")[0]) for k in state if k.startswith(prefix) and k[len(prefix):].split(".")[0].isdigit()})
        convs = [i for i in idx if get("%d.weight" % i).dim() == 3]
        self.layers = []
        for li, i in enumerate(convs):
            wgt = get("%d.weight" % i)
            layer = {"weight": wgt, "bias": get("%d.bias" % i), "bn": None,
                     "weight_t": _pad8(wgt.reshape(wgt.shape[0], -1).t().contiguous(), 0)}       # once, not per call
            if li + 1 < len(convs) and (prefix + "%d.running_var" % (i + 1)) in state:        # do_bn, not after the last Conv1d
                g, bta = get("%d.weight" % (i + 1)), get("%d.bias" % (i + 1))
                rm, rv = get("%d.running_mean" % (i + 1)), get("%d.running_var" % (i + 1))
                sc = g / torch.sqrt(rv + self.eps)
                layer["bn"] = {"gamma": g, "beta": bta, "eval": (sc.contiguous(), (bta - rm * sc).contiguous())}
            self.layers.append(layer)


def mlp(x, params, bn_train=False):
    """MLP.forward (modules.py:57-69): Conv1d -> BatchNorm1d -> ReLU -> ... -> Conv1d.  One GEMM launch per Conv1d; each
    BatchNorm + ReLU rides on the next layer's operand staging (eval: folded running statistics; bn_train: batch
    statistics, pats_bn_fold_f32).  An MLP built with do_bn=False would need a ReLU-only fold: scale 1, shift 0."""
    sc = sh = None
    h = _dev(x, "x")
    for i, layer in enumerate(params.layers):
        h = conv1d(h, layer["weight"], layer["bias"], sc, sh, weight_t=layer.get("weight_t"))
        if i + 1 < len(params.layers):
            bn = layer["bn"]
            if bn is None:
                sc = torch.ones((h.shape[1],), dtype=torch.float32, device=h.device)
                sh = torch.zeros_like(sc)
            elif bn_train:
                sc, sh = bn_fold(h, bn["gamma"], bn["beta"], params.eps)
            else:
                sc, sh = bn["eval"]
    return h


def keypoint_encoder(kpts, params, bn_train=False):
    """KeypointEncoder.forward (modules.py:77-82): kpts [n,2] -> [1, feature_dim, n] (first_layer.py:81,99 adds it to the
    coarse descriptors, third_layer.py:139-140 to the 8x8 windows).  `params` = MLPParams(kenc.state_dict(), prefix="encoder.")."""
    kpts = _dev(kpts, "kpts")
    inputs = kpts.transpose(0, 1).reshape(1, 2, -1).contiguous()
    return mlp(inputs, params, bn_train)


def scale_head(desc1, height, width, weights, biases, return_heads=False):
    """The scale head of a layer -> `ns` [b,1,h*w] of its OT problems:
        exp(sigmoid(proj(desc1[:, :, :h*w].reshape(b, C, h, w))) * log(256) - log(256) / 2),  proj = nn.Conv2d(C, 1, 3, padding=1)
    first_layer.py:106-107 (weights = [scalex_proj.weight]); second_layer.py:92-98 (weights = [scalex_proj.weight,
    scaley_proj.weight]: the product scale_x * scale_y); third_layer.py:151-152 ([scale_proj.weight]).  desc1 [b,C,ld]
    with ld = h*w or h*w + 1 (the dustbin feature column is skipped like `[:, :, :-1]`).  return_heads: also the list of
    the heads on their own ([b,1,h*w] each: scale_x, scale_y of second_layer.py:92-97, which est_position takes separately)."""
    desc1 = _dev(desc1, "desc1")
    b, C, ld = desc1.shape
    if not isinstance(weights, (list, tuple)):
        weights, biases = [weights], [biases]
    heads = len(weights)
    wt = torch.cat([_dev(wg, "weight").reshape(1, C, 3, 3) for wg in weights], dim=0).contiguous()
    bs = torch.cat([_dev(bi, "bias").reshape(1) for bi in biases]).contiguous()
    out = torch.empty((b, 1, height * width), dtype=torch.float32, device=desc1.device)
    per = torch.empty((b, heads, height * width), dtype=torch.float32, device=desc1.device) if return_heads else None
    _check(_L().pats_scale_head_f32(_ptr(desc1), b, C, ld, int(height), int(width), _ptr(wt), _ptr(bs), heads, _ptr(out),
                                    _ptr(per), _stream()), "scale_head")
    if return_heads:
        return out, [per[:, i:i + 1, :] for i in range(heads)]
    return out
