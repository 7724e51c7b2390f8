"""Runtime drop-in: rebind the reference's hot-path functions to pats_amd.ops so that the reference's own
drivers (`evaluate.py:106-108`, `demo.py`) run unchanged.

    import pats_amd.dropin
    pats_amd.dropin.install()        # before or after `from models.pats import PATS`
    model = PATS(config) ...         # the reference's code from here on

What is rebound (reference path:line -> replacement), in every already-imported module of the reference
that holds the name (the layer files do `from .modules import ...` / `from utils.utils import ...`, so the
functions live in several namespaces), and in `sys.modules` for the native extension:

    tensor_resize (module; setup/library.cpp:92-93, imported at utils/utils.py:17)   -> the repo's compiled tensor_resize extension
    models/modules.py:137,145,165  log_sinkhorn_iterations / log_optimal_transport / log_optimal_transport2
    models/modules.py:84           attention
    utils/utils.py:1179,1527       Iterative_expand_matrix / Compute_positions_and_ranges
    utils/utils.py:152,1343,189    split_patches / Compute_imgs / get_result
    models/second_layer.py:137,193 SecondLayer.merge_patches_old / merge_patches_new   (methods: `self` dropped)
    models/third_layer.py:184      ThirdLayer.Compute_result                          (method: 3-tuple as the reference)
    models/modules.py:77           KeypointEncoder.forward (six Conv1d GEMMs, BatchNorm + ReLU folded into the next
                                   layer's staging; follows `self.training` like the reference).  `final_proj` is an
                                   nn.Conv1d INSTANCE and is not rebound: ops.conv1d(x, m.weight, m.bias) replaces a call
    models/modules.py:114,127      AttentionalPropagation.forward / AttentionalGNN.forward (the layer's own parameters,
                                   read from its state_dict and cached on the instance under a fingerprint of every
                                   parameter / buffer (address, version, device) so load_state_dict / .to() / an optimizer
                                   step rebuild them; BatchNorm follows `self.training` like the reference - the third
                                   layer's stays in train mode, pats.py:112-120; with autograd enabled on parameters that
                                   require grad the reference's own forward runs instead: the HIP path is inference only)

Nothing of the reference is copied or imported here unless the caller has imported it already (or
`import_reference=True` and the reference is on sys.path).  `uninstall()` restores the originals.
"""
import importlib
import sys

_FUNCTIONS = {
    # name -> (defining reference module, attribute of pats_amd.ops)
    "log_sinkhorn_iterations": ("models.modules", "log_sinkhorn_iterations"),
    "log_optimal_transport": ("models.modules", "log_optimal_transport"),
    "log_optimal_transport2": ("models.modules", "log_optimal_transport2"),
    "attention": ("models.modules", "attention"),
    "Iterative_expand_matrix": ("utils.utils", "Iterative_expand_matrix"),
    "Compute_positions_and_ranges": ("utils.utils", "Compute_positions_and_ranges"),
    "split_patches": ("utils.utils", "split_patches"),
    "Compute_imgs": ("utils.utils", "Compute_imgs"),
    "get_result": ("utils.utils", "get_result"),
}
_REFERENCE_MODULES = ("models.modules", "utils.utils", "models.first_layer", "models.second_layer",
                      "models.third_layer", "models.pats")
_saved = []          # (object, attribute name, original value) in installation order
_MISSING = object()
_original = {}       # (class name, method) -> the reference's own method, for callers that need autograd
_cached_on = set()   # modules carrying a `_pats_params` cache (cleared by uninstall)


def _set(obj, name, value):
    _saved.append((obj, name, getattr(obj, name, _MISSING)))
    setattr(obj, name, value)


def _methods(ops):
    def merge_patches_new(self, patch_num, trust_score, original_image_shape, if_nomatching1_L1, if_nomatching1_L2,
                          scores_back):
        return ops.merge_patches_new(patch_num, trust_score, original_image_shape, if_nomatching1_L1,
                                     if_nomatching1_L2, scores_back)

    def merge_patches_old(self, patch_num, trust_score, original_image_shape, if_nomatching1_L1, if_nomatching1_L2,
                          scores_back):
        return ops.merge_patches_old(patch_num, trust_score, original_image_shape, if_nomatching1_L1,
                                     if_nomatching1_L2, scores_back)

    def Compute_result(self, scores, W, T, scale_x, scale_y, p_s, p_t, device):
        # third_layer.py:184-217 returns (mkpts0_f, mkpts1_f, whole_loss); the label of :161-170 stays with the caller
        return ops.Compute_result(scores, W, T, scale_x, scale_y, p_s, p_t, device)[:3]

    def _fingerprint(module):
        # (storage address, in-place version counter, device) of every parameter and buffer: load_state_dict(), .to(),
        # an optimizer step or a BatchNorm running-statistics update all change it
        return tuple((t.data_ptr(), t._version, str(t.device))
                     for t in list(module.parameters()) + list(module.buffers()))

    def _cached(module, make):
        fp = _fingerprint(module)
        hit = getattr(module, "_pats_params", None)
        if hit is None or hit[0] != fp:
            hit = (fp, make(next(module.parameters()).device))
            object.__setattr__(module, "_pats_params", hit)
            _cached_on.add(module)
        return hit[1]

    def _params(layer):
        return _cached(layer, lambda dev: ops.PropagationParams(layer.state_dict(), device=dev, eps=layer.mlp[1].eps))

    def _needs_autograd(module):
        # the HIP path is inference only (evaluate.py:20 `@torch.no_grad()`): it returns tensors outside the autograd
        # graph and does not update BatchNorm running statistics.  A caller that is TRAINING gets the reference's own
        # forward back instead of silently wrong gradients.
        import torch
        return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())

    def propagation_forward(self, x, source):
        if _needs_autograd(self):
            return _original[("AttentionalPropagation", "forward")](self, x, source)
        return ops.attentional_propagation(x, source, _params(self), heads=self.attn.num_heads, bn_train=self.training)

    def gnn_forward(self, desc0, desc1):
        if _needs_autograd(self):
            return _original[("AttentionalGNN", "forward")](self, desc0, desc1)
        layers = [_params(layer) for layer in self.layers]
        heads = self.layers[0].attn.num_heads if len(self.layers) else 4
        train = bool(len(self.layers) and self.layers[0].training)
        return ops.attentional_gnn(desc0, desc1, layers, self.names, heads=heads, bn_train=train)

    def kenc_forward(self, kpts):
        if _needs_autograd(self):
            return _original[("KeypointEncoder", "forward")](self, kpts)
        p = _cached(self, lambda dev: ops.MLPParams(self.state_dict(), device=dev, eps=self.encoder[1].eps,
                                                    prefix="encoder."))
        return ops.keypoint_encoder(kpts, p, bn_train=self.training)

    return {"models.modules#kenc": ("KeypointEncoder", {"forward": kenc_forward}),
            "models.second_layer": ("SecondLayer", {"merge_patches_new": merge_patches_new,
                                                    "merge_patches_old": merge_patches_old}),
            "models.third_layer": ("ThirdLayer", {"Compute_result": Compute_result}),
            "models.modules": ("AttentionalPropagation", {"forward": propagation_forward}),
            "models.modules#gnn": ("AttentionalGNN", {"forward": gnn_forward})}


def prepare_backbones(model, names=("descriptor_extract", "backbone", "compress", "compress_1", "compress_2")):
    """Optional, beside install(): switch the convolution stacks whose OUTPUTS the two descriptor gathers read -
    `SecondLayer.descriptor_extract` (second_layer.py:69: the three ResNet2.forward2 maps of a15) and
    `ThirdLayer.descriptor_extract` / `.backbone` (third_layer.py:113-117: the half-resolution maps of a16) - to
    torch.channels_last parameters.  PyTorch then picks channels-last for the convolutions' outputs as well (with MIOpen:
    PYTORCH_MIOPEN_SUGGEST_NHWC=1 in the environment), the logical shapes stay [B,C,H,W], and ops.fine_descriptors /
    ops.third_descriptors take their channels-last kernels (same bits; 4.0 + 3.0 instead of 6.4 + 8.0 ms per 48-pair step:
    a pixel's channels are one contiguous run).  The unchanged reference emits NCHW maps, which is what bench.py's headline
    runs on; this is the one-line opt-in behind `value_nhwc`.  Returns the sub-modules it converted."""
    import torch
    done = []
    for mod in model.modules():
        for n in names:
            sub = getattr(mod, n, None)
            if isinstance(sub, torch.nn.Module) and sub not in done:
                sub.to(memory_format=torch.channels_last)
                done.append(sub)
    return done


def install(import_reference=False, model=None):
    """Rebinds the names listed in the module docstring; returns the list of "module.name" it touched.
    Idempotent (a second call first undoes the first).  Needs the HIP library: pats_amd.ops raises if
    libpats_amd.so is missing - there is no CPU fallback to fall back to.
    model: the instantiated PATS (or any module holding the layers) - its gather-feeding backbones are switched to
    channels_last parameters as well (prepare_backbones: the maps then arrive in the order the gathers read fastest,
    profiles/r05_backbone_layout.txt); without it the maps stay NCHW and everything still works."""
    if _saved:
        uninstall()
    if model is not None:
        prepare_backbones(model)
    from . import ops
    native = _load_native()
    touched = []
    _set_module("tensor_resize", native)
    touched.append("sys.modules[tensor_resize]")
    if import_reference:
        for name in _REFERENCE_MODULES:
            try:
                importlib.import_module(name)
            except Exception:                        # the reference's optional dependencies may be absent
                pass
    originals = {}
    for fname, (home, _) in _FUNCTIONS.items():
        mod = sys.modules.get(home)
        if mod is not None and hasattr(mod, fname):
            originals[fname] = getattr(mod, fname)
    for mname in _REFERENCE_MODULES:
        mod = sys.modules.get(mname)
        if mod is None:
            continue
        if getattr(mod, "tensor_resize", None) is not None and mname == "utils.utils":
            _set(mod, "tensor_resize", native)       # `import tensor_resize` at utils/utils.py:17 bound the module object
            touched.append(mname + ".tensor_resize")
        for fname, (home, attr) in _FUNCTIONS.items():
            if not hasattr(mod, fname):
                continue
            cur = getattr(mod, fname)
            if mname == home or cur is originals.get(fname):
                _set(mod, fname, getattr(ops, attr))
                touched.append(mname + "." + fname)
    for mkey, (cls_name, methods) in _methods(ops).items():
        mname = mkey.split("#")[0]
        mod = sys.modules.get(mname)
        cls = getattr(mod, cls_name, None) if mod is not None else None
        if cls is None:
            continue
        for meth, fn in methods.items():
            if hasattr(cls, meth):
                _original[(cls_name, meth)] = getattr(cls, meth)
                _set(cls, meth, fn)
                touched.append("%s.%s.%s" % (mname, cls_name, meth))
    return touched


def _load_native():
    """This repository's COMPILED tensor_resize extension (csrc/binding/tensor_resize_ext.cpp, built by
    pats_amd.build), loaded by path: `import tensor_resize` could hand back the reference's own compiled extension if
    that is already in sys.modules or ahead on sys.path."""
    import importlib.machinery
    import importlib.util
    import os
    from . import build as _build
    path = _build.ext_path()
    if not os.path.exists(path):
        raise ImportError("pats_amd.dropin: %s is missing - build it with `python -m pats_amd.build`; there is no "
                          "Python fallback for the native module" % path)
    import torch  # noqa: F401  (the extension's Tensor casters need torch's Python side)
    loader = importlib.machinery.ExtensionFileLoader("tensor_resize", path)
    spec = importlib.util.spec_from_file_location("tensor_resize", path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def _set_module(name, module):
    _saved.append((sys.modules, name, sys.modules.get(name, _MISSING)))
    sys.modules[name] = module


def uninstall():
    """Restores everything install() replaced and drops the parameter caches it left on the reference's modules."""
    for module in list(_cached_on):
        try:
            object.__delattr__(module, "_pats_params")
        except AttributeError:
            pass
    _cached_on.clear()
    _original.clear()
    while _saved:
        obj, name, old = _saved.pop()
        if obj is sys.modules:
            if old is _MISSING:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = old
        elif old is _MISSING:
            try:
                delattr(obj, name)
            except AttributeError:
                pass
        else:
            setattr(obj, name, old)
