"""ctypes binding of oracle/libpats_oracle.so (numpy in, numpy out).

TEST INFRASTRUCTURE, NOT PRODUCT: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  pats_amd/ never imports it.  Function names mirror the reference's
(/root/reference/models/modules.py, utils/utils.py, models/third_layer.py, setup/library.cpp).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpats_oracle.so")

c_f = ctypes.POINTER(ctypes.c_float)
c_i64 = ctypes.POINTER(ctypes.c_int64)
c_i32 = ctypes.POINTER(ctypes.c_int32)
c_u8 = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    src = os.path.join(_HERE, "pats_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libpats_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_split_patches.restype = ctypes.c_int
        _lib.oracle_compute_imgs_bounds.restype = ctypes.c_int64
        _lib.oracle_left_crops.restype = ctypes.c_int64
        _lib.oracle_tensor_resize.restype = ctypes.c_int
        _lib.oracle_num_threads.restype = ctypes.c_int
        _lib.oracle_third_descriptors.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_f)


def _p(a, t):
    return a.ctypes.data_as(t)


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(ctypes.c_int(int(n)))


def cost(d0, d1):
    d0, p0 = _f(d0)
    d1, p1 = _f(d1)
    b, D, n = d0.shape
    m = d1.shape[2]
    out = np.empty((b, n, m), np.float32)
    lib().oracle_cost(p0, p1, ctypes.c_int64(b), D, n, m, _p(out, c_f))
    return out


def log_sinkhorn_iterations(Z, log_mu, log_nu, iters):
    Z, pz = _f(Z)
    log_mu, pm = _f(log_mu)
    log_nu, pn = _f(log_nu)
    b, M, N = Z.shape
    out = np.empty_like(Z)
    lib().oracle_sinkhorn(pz, ctypes.c_int64(b), M, N, pm, pn, int(iters), _p(out, c_f))
    return out


def log_optimal_transport(scores, alpha, ns, iters):
    scores, ps = _f(scores)
    ns, pn = _f(ns)
    b, m, n = scores.shape
    out = np.empty((b, m + 1, n + 1), np.float32)
    lib().oracle_log_optimal_transport(ps, ctypes.c_int64(b), m, n, ctypes.c_float(float(alpha)),
                                       pn, int(iters), _p(out, c_f))
    return out


def log_optimal_transport2(scores, one, ns, iters):
    scores, ps = _f(scores)
    ns, pn = _f(ns)
    b, m, n = scores.shape
    out = np.empty((b, m, n), np.float32)
    lib().oracle_log_optimal_transport2(ps, ctypes.c_int64(b), m, n, ctypes.c_float(float(one)),
                                        pn, int(iters), _p(out, c_f))
    return out


def colmass_sqrt(Z):
    Z, pz = _f(Z)
    b, M, N = Z.shape
    out = np.empty((b, N - 1), np.float32)
    lib().oracle_colmass_sqrt(pz, ctypes.c_int64(b), M, N, _p(out, c_f))
    return out


def dustbin_bias(Z, k):
    Z = np.array(Z, dtype=np.float32, order="C", copy=True)
    b, M, N = Z.shape
    lib().oracle_dustbin_bias(_p(Z, c_f), ctypes.c_int64(b), M, N, ctypes.c_float(float(k)))
    return Z


def argmax(Z):
    Z, pz = _f(Z)
    b, M, N = Z.shape
    r = np.empty((b, M), np.int64)
    c = np.empty((b, N), np.int64)
    lib().oracle_argmax(pz, ctypes.c_int64(b), M, N, _p(r, c_i64), _p(c, c_i64))
    return r, c


def compute_positions_and_ranges(h, w):
    """utils/utils.py:1527-1537: (positions [h*w,2], ranges [max(h,w), max(h,w)]) float32."""
    n = max(h, w)
    pos, rng = np.empty((h * w, 2), np.float32), np.empty((n, n), np.float32)
    lib().oracle_positions(int(h), int(w), _p(pos, c_f))
    lib().oracle_ranges(int(n), _p(rng, c_f))
    return pos, rng


def iterative_expand(P, scalex, scaley, lim3, h, w, lower_bound, iter_num, with_margin=False):
    """with_margin: a seventh output [b,m,2], the smallest relative distance by which (0) a rectangle decision and
    (1) a per-element `> lower_bound` test (whole_cost only) could have gone the other way - see pats_oracle.c."""
    P, pp = _f(P)
    b, M, N = P.shape
    scalex, px = _f(np.asarray(scalex).reshape(b, N - 1))
    scaley, py = _f(np.asarray(scaley).reshape(b, N - 1))
    m = M - 1
    whole = np.empty((b, m), np.float32)
    core = np.empty((b, m), np.float32)
    avg = np.empty((b, m, 2), np.float32)
    xs = np.empty((b, m), np.float32)
    ys = np.empty((b, m), np.float32)
    bound = np.empty((b, m, 4), np.int64)
    if with_margin:
        margin = np.empty((b, m, 2), np.float32)
        lib().oracle_iterative_expand_margin(pp, ctypes.c_int64(b), M, N, px, py, int(lim3), int(h), int(w),
                                             ctypes.c_float(lower_bound), int(iter_num), _p(whole, c_f),
                                             _p(core, c_f), _p(avg, c_f), _p(xs, c_f), _p(ys, c_f),
                                             _p(bound, c_i64), _p(margin, c_f))
        return whole, core, avg, xs, ys, bound, margin
    lib().oracle_iterative_expand(pp, ctypes.c_int64(b), M, N, px, py, int(lim3), int(h), int(w),
                                  ctypes.c_float(lower_bound), int(iter_num), _p(whole, c_f),
                                  _p(core, c_f), _p(avg, c_f), _p(xs, c_f), _p(ys, c_f),
                                  _p(bound, c_i64))
    return whole, core, avg, xs, ys, bound


def split_patches(sum_cycle, h, w, cap):
    sc = np.ascontiguousarray(sum_cycle, dtype=np.int32)
    second = np.zeros((h + 1, 2), np.int64)
    third = np.zeros((h + 1, 2), np.int64)
    n = lib().oracle_split_patches(_p(sc, c_i32), int(h), int(w), int(cap), _p(second, c_i64),
                                   _p(third, c_i64))
    return n, second[:n].copy(), third[:n].copy()


def compute_imgs_bounds(x_scale, y_scale, average_point, if_nomatching, height, width, img=0):
    xs, px = _f(np.asarray(x_scale).reshape(-1))
    ys, py = _f(np.asarray(y_scale).reshape(-1))
    ap, pa = _f(np.asarray(average_point).reshape(-1, 2))
    ifn = np.ascontiguousarray(np.asarray(if_nomatching).reshape(-1), dtype=np.uint8)
    Np = xs.shape[0]
    bound5 = np.zeros((Np, 5), np.int64)
    xsn = np.empty((Np, 2), np.float32)
    ysn = np.empty((Np, 2), np.float32)
    avn = np.empty((Np, 2), np.float32)
    K = lib().oracle_compute_imgs_bounds(px, py, pa, _p(ifn, c_u8), Np, int(height), int(width),
                                         int(img), _p(bound5, c_i64), _p(xsn, c_f), _p(ysn, c_f),
                                         _p(avn, c_f))
    return bound5[:K].copy(), xsn, ysn, avn


def left_crops(left_hwc, if_nomatching, h, w):
    left, pl = _f(left_hwc)
    H, W = left.shape[0], left.shape[1]
    ifn = np.ascontiguousarray(np.asarray(if_nomatching).reshape(-1), dtype=np.uint8)
    K = int((ifn == 0).sum())
    out = np.empty((K, 96, 96, 3), np.float32)
    lib().oracle_left_crops(pl, H, W, _p(ifn, c_u8), int(h), int(w), _p(out, c_f))
    return out


def tensor_resize(inp, bound):
    inp, pi = _f(inp)
    bound = np.ascontiguousarray(bound, dtype=np.int64).reshape(-1, 5)
    n_img, C, Hp, Wp = inp.shape
    K = bound.shape[0]
    out = np.zeros((K, C, 96, 96), np.float32)
    if K:
        rc = lib().oracle_tensor_resize(pi, n_img, C, Hp, Wp, _p(bound, c_i64), ctypes.c_int64(K),
                                        _p(out, c_f))
        if rc != 0:
            raise RuntimeError("tensor_resize: empty or out-of-range crop")
    return out


def compute_result(scores, scale_x, scale_y, p_s, p_t, outdoor):
    scores, ps = _f(scores)
    P = scores.shape[0]
    sx, px = _f(np.asarray(scale_x).reshape(P, 64))
    sy, py = _f(np.asarray(scale_y).reshape(P, 64))
    p_s = np.ascontiguousarray(p_s, dtype=np.int64).reshape(P, 2)
    p_t = np.ascontiguousarray(p_t, dtype=np.int64).reshape(P, 2)
    m0 = np.empty((P, 16, 2), np.float32)
    m1 = np.empty((P, 16, 2), np.float32)
    wl = np.empty((P, 16), np.float32)
    label = np.empty((P * 16, 2), np.float32)
    ifm = np.empty((P, 16), np.uint8)
    lib().oracle_compute_result(ps, ctypes.c_int64(P), px, py, _p(p_s, c_i64), _p(p_t, c_i64),
                                int(bool(outdoor)), _p(m0, c_f), _p(m1, c_f), _p(wl, c_f),
                                _p(label, c_f), _p(ifm, c_u8))
    return m0, m1, wl, label, ifm.astype(bool)


def fine_descriptors(f0, f1, f2, title, rubbish):
    f0, p0 = _f(f0)
    f1, p1 = _f(f1)
    f2, p2 = _f(f2)
    B = f0.shape[0] // 2
    ti, pt_ = _f(np.asarray(title).reshape(B, 8))
    ru, pr = _f(np.asarray(rubbish).reshape(B, 264))
    desc = np.empty((2, B, 264, 145), np.float32)
    lib().oracle_fine_descriptors(p0, p1, p2, pt_, pr, ctypes.c_int64(B), _p(desc, c_f))
    return desc


def third_descriptors(ff0, ff1, mk0, mk1, b_ids, kenc, rubbish):
    ff0, p0 = _f(ff0)
    ff1, p1 = _f(ff1)
    B = ff0.shape[0]
    mk0, pm0 = _f(np.asarray(mk0).reshape(-1, 2))
    mk1, pm1 = _f(np.asarray(mk1).reshape(-1, 2))
    P = mk0.shape[0]
    bi = np.ascontiguousarray(b_ids, dtype=np.int64).reshape(P)
    ke, pk = _f(np.asarray(kenc).reshape(128, 64))
    ru, pr = _f(np.asarray(rubbish).reshape(B, 128, 144))
    o0 = np.zeros((P, 128, 65), np.float32)
    o1 = np.zeros((P, 128, 65), np.float32)
    ps = np.zeros((P, 2), np.int64)
    pt = np.zeros((P, 2), np.int64)
    rc = lib().oracle_third_descriptors(p0, p1, pm0, pm1, _p(bi, c_i64), pk, pr, ctypes.c_int64(P),
                                        ctypes.c_int64(B), _p(o0, c_f), _p(o1, c_f), _p(ps, c_i64), _p(pt, c_i64))
    if rc != 0:
        raise IndexError("third_descriptors: gather index out of range")
    return o0, o1, ps, pt


# ---- SURVEY.md section 8(f) rows ------------------------------------------------------------------
def merge_patches(merge_new, trust, image_shape, ifn_L1, ifn_L2, scores_back):
    """second_layer.py:137-238.  Returns (if_nomatching [B,144] bool, trust', ifn_L2', scores_back');
    like the reference the 'old' variant hands back a zeroed scores_back to the caller, the array
    written during the call is returned as the 4th item for inspection."""
    t = np.array(trust, dtype=np.float32, order="C", copy=True)
    B = t.shape[0]
    l1 = np.ascontiguousarray(np.asarray(ifn_L1), dtype=np.uint8)
    bt = l1.shape[0]
    l2 = np.array(np.asarray(ifn_L2), dtype=np.uint8, order="C", copy=True).reshape(B, 144)
    sb = np.array(scores_back, dtype=np.float64, order="C", copy=True)
    out = np.ones((B, 144), np.uint8)
    lib().oracle_merge_patches.restype = ctypes.c_int
    rc = lib().oracle_merge_patches(int(bool(merge_new)), ctypes.c_int64(B), _p(t, c_f), int(image_shape[0]),
                                    int(image_shape[1]), int(bt), _p(l1, c_u8), _p(l2, c_u8),
                                    sb.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _p(out, c_u8))
    if rc == -1:
        raise IndexError("merge_patches: number of unmasked coarse patches != rows of trust_score")
    if rc != 0:
        raise RuntimeError("merge_patches: scatter index out of range")
    return out.astype(bool), t, l2.astype(bool), sb


def third_inputs(ifn2, pts):
    f = np.ascontiguousarray(np.asarray(ifn2), dtype=np.uint8)
    B = f.shape[0]
    pts, pp = _f(np.asarray(pts).reshape(B, 144, 2))
    n = int((f == 0).sum())
    mk0 = np.empty((n, 2), np.float32)
    mk1 = np.empty((n, 2), np.float32)
    b_ids = np.empty((n,), np.int64)
    lib().oracle_third_inputs.restype = ctypes.c_int64
    lib().oracle_third_inputs(_p(f, c_u8), pp, ctypes.c_int64(B), _p(mk0, c_f), _p(mk1, c_f), _p(b_ids, c_i64))
    return mk0, mk1, b_ids


def refine_scatter(ifn2, pts, mkpts1, label0):
    f = np.ascontiguousarray(np.asarray(ifn2), dtype=np.uint8)
    B = f.shape[0]
    pts, pp = _f(np.asarray(pts).reshape(B, 144, 2))
    mk, pm = _f(np.asarray(mkpts1).reshape(-1, 16, 2))
    lb, pl = _f(np.asarray(label0).reshape(-1))
    ifn16 = np.empty((B, 2304), np.uint8)
    pts16 = np.empty((B, 2304, 2), np.float32)
    lib().oracle_refine_scatter(_p(f, c_u8), pp, pm, pl, ctypes.c_int64(B), _p(ifn16, c_u8), _p(pts16, c_f))
    return ifn16.astype(bool), pts16


def get_result(batch_size, if_nomatching, average_point, scale, patch_size, left_choice):
    f0 = np.ascontiguousarray(np.asarray(if_nomatching[0]), dtype=np.uint8)
    f1 = np.ascontiguousarray(np.asarray(if_nomatching[1]), dtype=np.uint8)
    a0, pa0 = _f(average_point[0])
    a1, pa1 = _f(average_point[1])
    s0, ps0 = _f(scale[0])
    s1, ps1 = _f(scale[1])
    z0 = np.asarray(patch_size[0], dtype=np.int32)
    z1 = np.asarray(patch_size[1], dtype=np.int32)
    c0 = np.ascontiguousarray(np.asarray(left_choice[0]), dtype=np.uint8)
    c1 = np.ascontiguousarray(np.asarray(left_choice[1]), dtype=np.uint8)
    cap = int((f1 == 0).sum())
    ml = np.empty((cap, 2), np.float32)
    mr = np.empty((cap, 2), np.float32)
    lib().oracle_get_result.restype = ctypes.c_int64
    M = lib().oracle_get_result(int(batch_size), _p(f0, c_u8), _p(f1, c_u8), pa0, pa1, ps0, ps1, _p(z0, c_i32),
                                _p(z1, c_i32), _p(c0, c_u8), _p(c1, c_u8), _p(ml, c_f), _p(mr, c_f))
    return ml[:M].copy(), mr[:M].copy()


def attention(query, key, value, with_prob=True):
    """modules.py:84-88: (out [b,dim,heads,n], prob [b,heads,n,m])."""
    q, pq = _f(query)
    k, pk = _f(key)
    v, pv = _f(value)
    b, dim, heads, n = q.shape
    m = k.shape[3]
    out = np.empty((b, dim, heads, n), np.float32)
    prob = np.empty((b, heads, n, m), np.float32) if with_prob else None
    lib().oracle_attention(pq, pk, pv, ctypes.c_int64(b), dim, heads, n, m, _p(out, c_f),
                           _p(prob, c_f) if with_prob else None)
    return out, prob


def attentional_propagation(x, source, params, heads=4, bn_train=False, eps=1e-5, residual=None):
    """modules.py:107-117 (+ the residual of :133 when given).  params: the reference's state_dict names
    (attn.proj.{0,1,2}.weight/bias, attn.merge.*, mlp.0.*, mlp.1.weight/bias/running_mean/running_var, mlp.3.*)."""
    x, px = _f(x)
    s, ps = _f(source)
    b, C, n = x.shape
    m = s.shape[2]
    keep = []

    def w(name):
        a, p = _f(np.asarray(params[name]).reshape(np.asarray(params[name]).shape[0], -1))
        keep.append(a)
        return p
    out = np.empty((b, C, n), np.float32)
    res = None
    if residual is not None:
        r, res = _f(residual)
        keep.append(r)
    lib().oracle_attentional_propagation(
        px, ps, ctypes.c_int64(b), C, int(heads), n, m, w("attn.proj.0.weight"), w("attn.proj.0.bias"),
        w("attn.proj.1.weight"), w("attn.proj.1.bias"), w("attn.proj.2.weight"), w("attn.proj.2.bias"),
        w("attn.merge.weight"), w("attn.merge.bias"), w("mlp.0.weight"), w("mlp.0.bias"), w("mlp.1.weight"),
        w("mlp.1.bias"), w("mlp.1.running_mean"), w("mlp.1.running_var"), ctypes.c_float(eps), int(bool(bn_train)),
        w("mlp.3.weight"), w("mlp.3.bias"), res, _p(out, c_f))
    return out


def conv1d(x, weight, bias=None):
    """nn.Conv1d(kernel_size=1): x [b,K,n], weight [M,K(,1)], bias [M] or None -> [b,M,n]."""
    x, px = _f(x)
    w, pw = _f(np.asarray(weight).reshape(np.asarray(weight).shape[0], -1))
    M, K = w.shape
    assert x.shape[1] == K
    bs, pb = _f(np.zeros((M,), np.float32) if bias is None else np.asarray(bias).reshape(-1))
    out = np.empty((x.shape[0], M, x.shape[2]), np.float32)
    lib().oracle_conv1d(pw, pb, px, ctypes.c_int64(x.shape[0]), K, M, x.shape[2], _p(out, c_f))
    return out


def mlp(x, state, prefix="", bn_train=False, eps=1e-5):
    """MLP.forward (modules.py:57-69) from the nn.Sequential's state_dict names ("0.weight", "1.running_mean", ...)."""
    names = sorted({int(k[len(prefix):].split(".")[0]) for k in state if k.startswith(prefix) and k[len(prefix):].split(".")[0].isdigit()})
    convs = [i for i in names if np.asarray(state[prefix + "%d.weight" % i]).ndim == 3]
    h = np.ascontiguousarray(x, np.float32)
    for li, i in enumerate(convs):
        h = conv1d(h, state[prefix + "%d.weight" % i], state[prefix + "%d.bias" % i])
        if li + 1 < len(convs):
            keep = [_f(np.asarray(state[prefix + "%d.%s" % (i + 1, nm)])) for nm in ("weight", "bias", "running_mean", "running_var")]
            h = np.ascontiguousarray(h, np.float32)
            lib().oracle_bn_relu(_p(h, c_f), ctypes.c_int64(h.shape[0]), h.shape[1], h.shape[2], keep[0][1], keep[1][1],
                                 keep[2][1], keep[3][1], ctypes.c_float(eps), int(bool(bn_train)))
    return h


def keypoint_encoder(kpts, state, prefix="encoder.", bn_train=False, eps=1e-5):
    """KeypointEncoder.forward (modules.py:77-82): kpts [n,2] -> [1, feature_dim, n]."""
    k = np.ascontiguousarray(np.asarray(kpts, np.float32).T.reshape(1, 2, -1))
    return mlp(k, state, prefix, bn_train, eps)


def scale_head(desc1, h, w, weights, biases):
    """exp(sigmoid(Conv2d(C, 1, 3, padding=1)(desc1[:, :, :h*w] as [b,C,h,w])) * ln256 - ln256 / 2), heads multiplied
    (first_layer.py:106-107, second_layer.py:92-98, third_layer.py:151-152) -> [b,1,h*w]."""
    x, px = _f(desc1)
    b, C, ld = x.shape
    wt, pw = _f(np.concatenate([np.asarray(v, np.float32).reshape(1, C, 3, 3) for v in weights], 0))
    bs, pb = _f(np.concatenate([np.asarray(v, np.float32).reshape(1) for v in biases]))
    out = np.empty((b, 1, h * w), np.float32)
    lib().oracle_scale_head(px, ctypes.c_int64(b), C, ld, int(h), int(w), pw, pb, len(weights), _p(out, c_f))
    return out
