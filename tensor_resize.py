"""Fallback loader for the reference's native extension module `tensor_resize`
(setup/library.cpp:92-93 `PYBIND11_MODULE(tensor_resize, m)`, built by setup/setup.py:107-118,
imported at utils/utils.py:17, called at utils/utils.py:1385).

The module itself is COMPILED: pats_amd/csrc/binding/tensor_resize_ext.cpp (pybind11 + libtorch over the
C-ABI `pats_tensor_resize_f32`), built by `python -m pats_amd.build` into tensor_resize.cpython-*.so next to
this file - and an extension module in the same directory wins the import, so this file normally never runs.
It only runs when the .so sits elsewhere (e.g. this file was copied alone onto PYTHONPATH): it then loads the
compiled module by path and puts IT into sys.modules.  There is no Python or CPU implementation behind it.
"""
import importlib.machinery
import importlib.util
import os
import sys
import sysconfig

_here = os.path.dirname(os.path.abspath(__file__))
_name = "tensor_resize" + sysconfig.get_config_var("EXT_SUFFIX")
_candidates = [os.path.join(_here, _name), os.path.join(_here, "pats_amd", _name)]
try:
    import pats_amd as _pkg
    _candidates += [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(_pkg.__file__))), _name),
                    os.path.join(os.path.dirname(os.path.abspath(_pkg.__file__)), _name)]
except ImportError:
    pass
for _path in _candidates:
    if os.path.exists(_path):
        _loader = importlib.machinery.ExtensionFileLoader("tensor_resize", _path)
        _spec = importlib.util.spec_from_file_location("tensor_resize", _path, loader=_loader)
        _mod = importlib.util.module_from_spec(_spec)
        _loader.exec_module(_mod)
        sys.modules["tensor_resize"] = _mod
        tensor_resize = _mod.tensor_resize
        break
else:
    raise ImportError("tensor_resize: the compiled extension %s was not found (looked in %s) - build it with "
                      "`python -m pats_amd.build` / __graft_entry__.build(); there is no Python fallback"
                      % (_name, ", ".join(sorted(set(os.path.dirname(c) for c in _candidates)))))
