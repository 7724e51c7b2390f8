import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/libpats_oracle.so) - the CHECKER, never the thing under test in
    the -m gpu parity tests."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import pats_oracle
    pats_oracle.lib()
    return pats_oracle


def reference_tensor_resize(src, bound):
    """The reference's own library.cpp, compiled unmodified into oracle/_ref/ (oracle/build_ref.sh), run on CPU tensors
    in a SUBPROCESS: it is a pybind11 module named `tensor_resize` like the product's compiled extension, and pybind11
    keeps one module object per name and interpreter - two same-named extension modules in one process hand each other's
    functions out.  Returns None when oracle/_ref was not built (the reference tree was absent at build time)."""
    import subprocess
    import tempfile
    ref_dir = os.path.join(REPO, "oracle", "_ref")
    so = [f for f in os.listdir(ref_dir) if f.startswith("tensor_resize") and f.endswith(".so")] if os.path.isdir(ref_dir) else []
    if not so:
        return None
    code = (
        "import sys, importlib.util, importlib.machinery, numpy as np, torch\n"
        "path, io = sys.argv[1], sys.argv[2]\n"
        "ld = importlib.machinery.ExtensionFileLoader('tensor_resize', path)\n"
        "m = importlib.util.module_from_spec(importlib.util.spec_from_file_location('tensor_resize', path, loader=ld))\n"
        "ld.exec_module(m)\n"
        "d = np.load(io)\n"
        "out = m.tensor_resize(torch.from_numpy(d['src']), torch.from_numpy(d['bound']))\n"
        "np.save(io + '.out.npy', out.numpy())\n")
    with tempfile.TemporaryDirectory() as tmp:
        io = os.path.join(tmp, "io.npz")
        np.savez(io, src=np.ascontiguousarray(src, dtype=np.float32), bound=np.ascontiguousarray(bound, dtype=np.int64))
        p = subprocess.run([sys.executable, "-c", code, os.path.join(ref_dir, so[0]), io], capture_output=True, text=True,
                           cwd=tmp)
        assert p.returncode == 0, p.stderr[-2000:]
        return np.load(io + ".out.npy")
