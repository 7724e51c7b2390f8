"""The compiled `tensor_resize` extension module (pats_amd/csrc/binding/tensor_resize_ext.cpp -> tensor_resize.cpython-*.so
at the repo root): SURVEY 8(b1), the reference's one native boundary (setup/library.cpp:47-66,92-93, built by
setup/setup.py:107-118).  Loaded BY PATH here, never through a .py shim.  CPU tests: it builds, loads and refuses host
tensors; GPU tests: the reference's compiled library.cpp (oracle/_ref) on the same crops, K = 0, strided inputs, the
error cases."""
import importlib.machinery
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, golden, reference_tensor_resize


def _load_by_path(path, name="tensor_resize"):
    loader = importlib.machinery.ExtensionFileLoader(name, path)
    spec = importlib.util.spec_from_file_location(name, path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def ext():
    from pats_amd import build
    build.build()
    path = build.build_tensor_resize_ext()
    assert os.path.dirname(path) == REPO and path.endswith(".so")
    return _load_by_path(path)


def test_extension_is_compiled_and_links_the_c_abi(ext):
    assert type(ext.tensor_resize).__name__ == "builtin_function_or_method"
    assert "input_tensor: torch.Tensor, bound: torch.Tensor" in ext.tensor_resize.__doc__
    needed = subprocess.run(["readelf", "-d", ext.__file__], capture_output=True, text=True).stdout
    assert "libpats_amd.so" in needed and "libtorch_python.so" in needed
    assert "$ORIGIN/pats_amd" in needed          # finds the product library in-tree, wherever the tree is mounted


def test_plain_import_resolves_to_the_compiled_module(ext):
    # an extension module beside tensor_resize.py wins the import; the .py is only a by-path loader of the same .so
    code = ("import sys; sys.path.insert(0, %r); import tensor_resize as t; print(t.__file__); "
            "print(type(t.tensor_resize).__name__)" % REPO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[0] == ext.__file__ and "builtin" in out.stdout
    shim = importlib.util.spec_from_file_location("tensor_resize_shim_probe", os.path.join(REPO, "tensor_resize.py"))
    before = sys.modules.get("tensor_resize")
    try:
        m = importlib.util.module_from_spec(shim)
        shim.loader.exec_module(m)
        assert sys.modules["tensor_resize"].__file__ == ext.__file__ and m.tensor_resize.__doc__ == ext.tensor_resize.__doc__
    finally:
        if before is None:
            sys.modules.pop("tensor_resize", None)
        else:
            sys.modules["tensor_resize"] = before


def test_host_tensors_and_bad_arguments_raise(ext):
    src = torch.zeros(1, 3, 16, 16)
    b = torch.tensor([[0, 4, 0, 4, 0]], dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ext.tensor_resize(src, b)
    with pytest.raises(TypeError):
        ext.tensor_resize(src.numpy(), b)


# ---- on the MI355X ---------------------------------------------------------------------------------------------------
def _cases():
    rng = np.random.default_rng(7)
    src = torch.from_numpy(rng.uniform(0, 255, (2, 3, 200, 240)).astype(np.float32))
    y0 = rng.integers(0, 150, 64)
    x0 = rng.integers(0, 180, 64)
    bound = np.stack([y0, y0 + rng.integers(1, 50, 64), x0, x0 + rng.integers(0, 59, 64),
                      rng.integers(0, 2, 64) * 10000 + np.arange(64)], 1).astype(np.int64)
    return src, torch.from_numpy(bound)


@pytest.mark.gpu
def test_against_the_compiled_reference_on_64_crops(ext, oracle):
    src, bound = _cases()
    got = ext.tensor_resize(src.cuda(), bound.cuda())
    assert got.shape == (64, 3, 96, 96) and got.dtype == torch.float32 and got.is_cuda
    want = reference_tensor_resize(src.numpy(), bound.numpy())     # library.cpp itself, compiled unmodified (oracle/build_ref.sh)
    if want is None:
        want = oracle.tensor_resize(src.numpy(), bound.numpy())
    else:
        np.testing.assert_allclose(oracle.tensor_resize(src.numpy(), bound.numpy()), want, atol=1e-4)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=1e-4)
    maps = open("/proc/self/maps").read()
    assert os.path.basename(ext.__file__) in maps and "libpats_amd.so" in maps


@pytest.mark.gpu
def test_empty_bound_strided_input_and_borrowed_arguments(ext, oracle):
    src, bound = _cases()
    empty = ext.tensor_resize(src.cuda(), torch.zeros(0, 5, dtype=torch.int64).cuda())
    assert empty.shape == (0, 3, 96, 96) and empty.is_cuda
    # a channels-last view (what `.permute(0,3,1,2)` of an HWC image is) and a strided bound: packed, not refused
    hwc = src.permute(0, 2, 3, 1).contiguous().cuda()
    strided = hwc.permute(0, 3, 1, 2)
    assert not strided.is_contiguous()
    wide = torch.zeros(64, 10, dtype=torch.int64)
    wide[:, ::2] = bound
    b_strided = wide.cuda()[:, ::2]
    assert not b_strided.is_contiguous()
    keep_src, keep_b = strided.clone(), b_strided.clone()
    got = ext.tensor_resize(strided, b_strided)
    np.testing.assert_allclose(got.cpu().numpy(), oracle.tensor_resize(src.numpy(), bound.numpy()), atol=1e-4)
    assert torch.equal(strided, keep_src) and torch.equal(b_strided, keep_b)        # inputs are never written
    assert torch.equal(got, ext.tensor_resize(src.cuda(), bound.cuda()))
    # the caller's current stream is the one used
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        on_s = ext.tensor_resize(src.cuda(), bound.cuda())
    s.synchronize()
    assert torch.equal(on_s, got)


@pytest.mark.gpu
def test_error_cases_raise_runtime_error(ext):
    """library.cpp:56-60: narrow() outside the tensor or an empty crop into upsample_bilinear2d is a c10::Error there."""
    g = golden("resize_small.npz")
    src = torch.from_numpy(g["src"]).cuda()
    Hp, Wp = src.shape[2], src.shape[3]
    ok = torch.from_numpy(g["bound"]).cuda()[:1]
    for bad in ([0, Hp + 1, 0, 10, 0], [5, 5, 0, 10, 0], [0, 10, 4, Wp, 0], [0, 10, 0, 10, 10000 * src.shape[0]],
                [-1, 10, 0, 10, 0]):
        b = torch.cat([ok, torch.tensor([bad], dtype=torch.int64).cuda()])
        with pytest.raises(RuntimeError, match="tensor_resize"):
            ext.tensor_resize(src, b)
    with pytest.raises(RuntimeError, match="float32"):
        ext.tensor_resize(src.double(), ok)
    with pytest.raises(RuntimeError, match="int64"):
        ext.tensor_resize(src, ok.int())
    with pytest.raises(RuntimeError, match=r"\[K,5\]"):
        ext.tensor_resize(src, ok[:, :4])
    with pytest.raises(RuntimeError, match="bound is on"):
        ext.tensor_resize(src, ok.cpu())
    np.testing.assert_allclose(ext.tensor_resize(src, torch.from_numpy(g["bound"]).cuda()).cpu().numpy(), g["out"], atol=1e-4)
