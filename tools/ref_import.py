"""Import the reference's hot-path Python (this container only; never on the GPU box).

TEST INFRASTRUCTURE.  Used only by tools/make_golden.py (fixture generation) and by the
`-m "not gpu"` tests that cross-check the oracle against the live reference when
/root/reference happens to be present.  The product path (pats_amd/) never imports this.

Recipe = SURVEY.md section 8c:
  * /root/reference/models/modules.py imports with torch only;
  * /root/reference/utils/utils.py needs cv2/h5py/imagesize/pydegensac/open3d (absent) and
    `numpy.lib.function_base.average` (removed in numpy 2) -> empty stand-in modules, none of
    which is reached by the hot-path functions;
  * the layer modules need torchvision / kornia (absent) at import time only; their hot-path
    methods are called as unbound functions (no constructor runs, no weights exist);
  * `tensor_resize` is the reference's own library.cpp compiled by oracle/build_ref.sh.
The only stand-in with behaviour is kornia.utils.grid.create_meshgrid (kornia==0.5.5 pinned at
/root/reference/environment.yaml:23,36; un-vendored): [1,H,W,2] grid, [...,0]=x, [...,1]=y,
un-normalised when normalized_coordinates=False.  Parity at that call is "unpinned" (index grid
only, no arithmetic) - see DESIGN.md.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("PATS_REFERENCE_ROOT", "/root/reference")
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_REF_SO_DIR = os.path.join(_REPO, "oracle", "_ref")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "models", "modules.py"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """Returns a namespace with the reference modules (M=modules, U=utils, L1/L2/L3 layers)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    import numpy as np
    import torch

    sys.dont_write_bytecode = True
    for p in (_REF_SO_DIR, REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    for name in ("cv2", "h5py", "imagesize", "pydegensac", "open3d"):
        _stub(name)
    _stub("numpy.lib.function_base", average=np.average)

    class _Normalize:  # never instantiated on the hot path
        def __init__(self, *a, **k):
            pass

    tv = _stub("torchvision")
    tvm = _stub("torchvision.models")
    tvt = _stub("torchvision.transforms", Normalize=_Normalize)
    tvf = _stub("torchvision.transforms.functional")
    tv.models, tv.transforms = tvm, tvt
    tvt.functional = tvf

    def create_meshgrid(height, width, normalized_coordinates=True, device=None):
        assert not normalized_coordinates
        xs = torch.linspace(0, width - 1, width, device=device)
        ys = torch.linspace(0, height - 1, height, device=device)
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        return torch.stack([gx, gy], dim=-1).unsqueeze(0)

    k = _stub("kornia")
    ku = _stub("kornia.utils")
    kg = _stub("kornia.utils.grid", create_meshgrid=create_meshgrid)
    k.utils, ku.grid = ku, kg

    import tensor_resize as ref_tensor_resize  # oracle/_ref build of library.cpp
    import models.modules as M
    import utils.utils as U
    import models.first_layer as L1
    import models.second_layer as L2
    import models.third_layer as L3

    _loaded = types.SimpleNamespace(M=M, U=U, L1=L1, L2=L2, L3=L3,
                                    tensor_resize=ref_tensor_resize)
    return _loaded
