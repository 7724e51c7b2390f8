"""pats_amd.dropin.install(): the reference's names are rebound at run time so that its own drivers
(evaluate.py:106-108) run unchanged.  CPU test: no kernel is launched, only namespaces are inspected.
Stand-in modules carry the reference's module / function / class names and signatures (typed here from
the signatures, no reference source); when /root/reference is present the real modules are checked too."""
import inspect
import sys
import types

import pytest

from conftest import REPO  # noqa: F401


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# the reference's signatures (models/modules.py:84,137,145,165; utils/utils.py:152,189,1179,1343,1527;
# models/second_layer.py:137,193; models/third_layer.py:184)
def log_sinkhorn_iterations(Z, log_mu, log_nu, iters: int): return "ref"
def log_optimal_transport(scores, alpha, ns, iters: int): return "ref"
def log_optimal_transport2(scores, one, ns, iters: int): return "ref"
def attention(query, key, value): return "ref"
def Iterative_expand_matrix(scores_in, scalex, scaley, limitation, ranges, positions, lower_bound=1e-3, upper_bound=1e7,
                            iter_num=15, width=20, height=15, type='distance'): return "ref"
def Compute_positions_and_ranges(height, width, device): return "ref"
def split_patches(sum_cycle, height, width, max_once_used=350): return "ref"
def Compute_imgs(x_scale, y_scale, average_point, if_nomatching, left, right, sequence_num=0, output_path=None,
                 if_view=False, margin=128, width=20, height=15, patch_scale=32): return "ref"
def get_result(batch_size, if_nomatching, average_point, scale, patch_size, left_choice, layer_num=2): return "ref"


class SecondLayer:
    def merge_patches_old(self, patch_num, trust_score, original_image_shape, if_nomatching1_L1, if_nomatching1_L2,
                          scores_back): return "ref"
    def merge_patches_new(self, patch_num, trust_score, original_image_shape, if_nomatching1_L1, if_nomatching1_L2,
                          scores_back): return "ref"


class ThirdLayer:
    def Compute_result(self, scores, W, T, scale_x, scale_y, p_s, p_t, device): return "ref"


class AttentionalPropagation:
    def forward(self, x, source): return "ref"


class AttentionalGNN:
    def forward(self, desc0, desc1): return "ref"


class KeypointEncoder:
    def forward(self, kpts): return "ref"


REF_NAMES = ("models", "models.modules", "models.first_layer", "models.second_layer", "models.third_layer", "utils",
             "utils.utils", "tensor_resize")


@pytest.fixture
def standins():
    saved = {n: sys.modules.get(n) for n in REF_NAMES}
    native = _mod("tensor_resize", tensor_resize=lambda a, b: "ref")
    _mod("models")
    _mod("utils")
    mm = _mod("models.modules", log_sinkhorn_iterations=log_sinkhorn_iterations, log_optimal_transport=log_optimal_transport,
              log_optimal_transport2=log_optimal_transport2, attention=attention, AttentionalPropagation=AttentionalPropagation,
              AttentionalGNN=AttentionalGNN, KeypointEncoder=KeypointEncoder)
    uu = _mod("utils.utils", Iterative_expand_matrix=Iterative_expand_matrix, split_patches=split_patches,
              Compute_positions_and_ranges=Compute_positions_and_ranges, Compute_imgs=Compute_imgs, get_result=get_result,
              tensor_resize=native)
    # the layer files import the names into their own namespaces (first_layer.py:7-8 etc.)
    l1 = _mod("models.first_layer", log_optimal_transport=log_optimal_transport, Iterative_expand_matrix=Iterative_expand_matrix,
              split_patches=split_patches, Compute_imgs=Compute_imgs, Compute_positions_and_ranges=Compute_positions_and_ranges,
              unrelated=lambda: "keep")
    l2 = _mod("models.second_layer", log_optimal_transport2=log_optimal_transport2, SecondLayer=SecondLayer,
              Iterative_expand_matrix=Iterative_expand_matrix)
    l3 = _mod("models.third_layer", log_optimal_transport2=log_optimal_transport2, ThirdLayer=ThirdLayer)
    yield dict(mm=mm, uu=uu, l1=l1, l2=l2, l3=l3, native=native)
    from pats_amd import dropin
    dropin.uninstall()
    for n, m in saved.items():
        if m is None:
            sys.modules.pop(n, None)
        else:
            sys.modules[n] = m


def _prefix_compatible(ref_fn, new_fn, drop_self=False):
    """every parameter of the reference's function exists, in the same position, in the replacement"""
    rp = list(inspect.signature(ref_fn).parameters)
    npar = list(inspect.signature(new_fn).parameters)
    if drop_self:
        rp = rp[1:]
        npar = npar[1:] if npar and npar[0] == "self" else npar
    return npar[:len(rp)] == rp


def test_install_rebinds_every_namespace_and_restores(standins):
    from pats_amd import dropin, ops
    import tensor_resize as mine_before  # noqa: F401  (the stand-in at this point)
    touched = dropin.install()
    s = standins
    assert s["mm"].log_optimal_transport is ops.log_optimal_transport
    assert s["l1"].log_optimal_transport is ops.log_optimal_transport          # `from .modules import ...` copies too
    assert s["l2"].log_optimal_transport2 is ops.log_optimal_transport2 and s["l3"].log_optimal_transport2 is ops.log_optimal_transport2
    assert s["mm"].log_sinkhorn_iterations is ops.log_sinkhorn_iterations and s["mm"].attention is ops.attention
    for n in ("Iterative_expand_matrix", "Compute_positions_and_ranges", "split_patches", "Compute_imgs", "get_result"):
        assert getattr(s["uu"], n) is getattr(ops, n)
    assert s["l1"].Compute_imgs is ops.Compute_imgs and s["l2"].Iterative_expand_matrix is ops.Iterative_expand_matrix
    assert s["l1"].unrelated() == "keep"
    import importlib
    native = importlib.import_module("tensor_resize")
    # the native module is the COMPILED extension of this repository (csrc/binding/tensor_resize_ext.cpp), not a .py
    assert native.__file__.endswith(".so") and "pats_amd" in native.__doc__ and s["uu"].tensor_resize is native
    assert type(native.tensor_resize).__name__ == "builtin_function_or_method"
    assert SecondLayer.merge_patches_new is not None and SecondLayer().merge_patches_new.__func__.__module__ == "pats_amd.dropin"
    assert ThirdLayer().Compute_result.__func__.__module__ == "pats_amd.dropin"
    assert AttentionalPropagation.forward.__module__ == "pats_amd.dropin" and AttentionalGNN.forward.__module__ == "pats_amd.dropin"
    assert list(inspect.signature(AttentionalPropagation.forward).parameters) == ["self", "x", "source"]
    assert list(inspect.signature(AttentionalGNN.forward).parameters) == ["self", "desc0", "desc1"]
    assert KeypointEncoder.forward.__module__ == "pats_amd.dropin" and list(inspect.signature(KeypointEncoder.forward).parameters) == ["self", "kpts"]
    assert "models.first_layer.log_optimal_transport" in touched and "models.third_layer.ThirdLayer.Compute_result" in touched
    # signatures: same positional parameters as the reference's
    for ref, new in ((log_sinkhorn_iterations, ops.log_sinkhorn_iterations), (log_optimal_transport, ops.log_optimal_transport),
                     (log_optimal_transport2, ops.log_optimal_transport2), (attention, ops.attention),
                     (Iterative_expand_matrix, ops.Iterative_expand_matrix), (split_patches, ops.split_patches),
                     (Compute_positions_and_ranges, ops.Compute_positions_and_ranges), (Compute_imgs, ops.Compute_imgs),
                     (get_result, ops.get_result)):
        assert _prefix_compatible(ref, new), ref.__name__
    dropin.uninstall()
    assert s["l1"].log_optimal_transport is log_optimal_transport and s["uu"].get_result is get_result
    assert SecondLayer().merge_patches_new(1, 2, 3, 4, 5, 6) == "ref" and ThirdLayer().Compute_result(*range(8)) == "ref"
    assert AttentionalPropagation().forward(1, 2) == "ref" and AttentionalGNN().forward(1, 2) == "ref"
    assert KeypointEncoder().forward(1) == "ref"
    assert sys.modules["tensor_resize"] is s["native"]


def test_install_against_the_real_reference_when_present():
    """In the build container the reference itself is importable (tools/ref_import.py): the same rebinding on the real
    modules, and the real signatures against the replacements."""
    sys.path.insert(0, REPO + "/tools")
    import ref_import
    if not ref_import.available():
        pytest.skip("reference tree not present")
    path_before, native_before = list(sys.path), sys.modules.get("tensor_resize")
    R = ref_import.load()
    from pats_amd import dropin, ops
    originals = {n: getattr(R.M, n) for n in ("log_sinkhorn_iterations", "log_optimal_transport", "log_optimal_transport2", "attention")}
    originals.update({n: getattr(R.U, n) for n in ("Iterative_expand_matrix", "Compute_positions_and_ranges", "split_patches",
                                                   "Compute_imgs", "get_result")})
    merge_new, comp_res = R.L2.SecondLayer.merge_patches_new, R.L3.ThirdLayer.Compute_result
    prop_fwd, gnn_fwd = R.M.AttentionalPropagation.forward, R.M.AttentionalGNN.forward
    kenc_fwd = R.M.KeypointEncoder.forward
    try:
        touched = dropin.install()
        for n, ref in originals.items():
            assert _prefix_compatible(ref, getattr(ops, n)), n
        assert R.L1.log_optimal_transport is ops.log_optimal_transport
        assert R.L2.log_optimal_transport2 is ops.log_optimal_transport2 and R.L3.log_optimal_transport2 is ops.log_optimal_transport2
        assert R.U.Compute_imgs is ops.Compute_imgs and "pats_amd" in R.U.tensor_resize.__doc__ \
            and R.U.tensor_resize.__file__.endswith(".so")
        assert _prefix_compatible(merge_new, R.L2.SecondLayer.merge_patches_new, drop_self=True)
        assert _prefix_compatible(comp_res, R.L3.ThirdLayer.Compute_result, drop_self=True)
        assert len(touched) >= 17
        assert _prefix_compatible(prop_fwd, R.M.AttentionalPropagation.forward) and R.M.AttentionalPropagation.forward is not prop_fwd
        assert _prefix_compatible(gnn_fwd, R.M.AttentionalGNN.forward)
        assert _prefix_compatible(kenc_fwd, R.M.KeypointEncoder.forward) and R.M.KeypointEncoder.forward is not kenc_fwd
    finally:
        dropin.uninstall()
        # ref_import put the reference's compiled extension first on sys.path / into sys.modules: later tests
        # must find this repository's tensor_resize.py again
        sys.path[:] = path_before
        if native_before is None:
            sys.modules.pop("tensor_resize", None)
        else:
            sys.modules["tensor_resize"] = native_before
    assert R.L1.log_optimal_transport is originals["log_optimal_transport"] and R.L2.SecondLayer.merge_patches_new is merge_new
    assert R.M.AttentionalPropagation.forward is prop_fwd and R.M.AttentionalGNN.forward is gnn_fwd
    assert R.M.KeypointEncoder.forward is kenc_fwd
