"""Drop-in for the reference's native extension module `tensor_resize`
(setup/library.cpp:92-93 `PYBIND11_MODULE(tensor_resize, m)`, built by setup/setup.py:114-115,
imported at utils/utils.py:17, called at utils/utils.py:1385):

    import tensor_resize
    crops = tensor_resize.tensor_resize(resize_source, bound_new)   # [K,C,96,96] float32

Same module name, function name, argument order, dtypes and result layout; backed by the batched
HIP gather kernel (pats_amd/csrc/resize.hip) through the C-ABI `pats_tensor_resize_f32`.
"""
from pats_amd.ops import tensor_resize  # noqa: F401

__all__ = ["tensor_resize"]
