"""Builds pats_amd/libpats_amd.so (hipcc, gfx950 only) in-tree.  Called by __graft_entry__.build()."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpats_amd.so")
# diagnostic twin: the same objects, third_fused3.hip compiled with -DPATS_DIAG (every sweep-loop variant and the timing
# ablations whose results are wrong by design).  Built on request only (`--diag`); never loaded unless PATS_AMD_DIAG_LIB=1.
LIB_DIAG = os.path.join(HERE, "libpats_amd_diag.so")
DIAG_SOURCES = None      # every source, each with -DPATS_DIAG=1 (set below SOURCES): diag_env() (csrc/common.hpp) is live in that library only
SOURCES = ["host.cpp", "sinkhorn.hip", "sinkhorn_stream.hip", "sinkhorn_blk.hip", "sinkhorn_blk2w.hip", "cost.hip", "post.hip", "expand.hip", "resize.hip", "third.hip", "third_fused.hip", "third_fused3.hip", "gather.hip", "merge.hip", "attention.hip", "attention145.hip", "gnn.hip", "gnn_fused.hip", "gnn_fine.hip", "conv_pk.hip",
           "fused.hip", "scale_head.hip", "batch.hip", "chunk_walk.cpp"]
DIAG_SOURCES = {src: ["-DPATS_DIAG=1"] for src in SOURCES}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-x", "hip"]
# third_fused.hip runs at the 168-VGPR edge (three waves per SIMD).  The SLP vectoriser pairs the
# per-lane and dustbin scaling updates into v_pk_mul_f32, which pins {nu, nu64} / {mu, mu64} in two
# extra VGPRs and puts a spill reload + vmcnt(0) into every Sinkhorn sweep (+12 % kernel time,
# measured).  The explicit float2 FMAs of the sweeps are not SLP products and stay packed.
EXTRA_FLAGS = {"third_fused.hip": ["-fno-slp-vectorize"], "third_fused3.hip": ["-fno-slp-vectorize"],
               "cost.hip": ["-fno-slp-vectorize"], "gnn.hip": ["-fno-slp-vectorize"]}


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found - libpats_amd.so cannot be built")
    return exe


def compile_tu(spath, obj, extra=(), verbose=False):
    """One translation unit -> host object with the gfx950 code object embedded: plain `hipcc -c`.  (Round 3 split hipcc's
    steps to patch `s_waitcnt lgkmcnt(0)` into the device assembly in front of every barrier; since round 4 the wait is in the
    source - `wg_barrier()` of csrc/common.hpp - and tools/check_code_objects.py checks the built library for it.)"""
    cmd = [hipcc()] + FLAGS[:-2] + list(extra) + FLAGS[-2:] + ["-c", spath, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if p.returncode != 0:
        sys.stderr.write(p.stdout.decode())
        raise RuntimeError("hipcc failed on %s" % os.path.basename(spath))
    if verbose and p.stdout:
        print(p.stdout.decode())


def compile_many(jobs, verbose=False):
    """jobs: [(source path, object path, extra flags)] compiled concurrently."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        futs = [ex.submit(compile_tu, s, o, e, verbose) for s, o, e in jobs]
        return [f.result() for f in futs]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "pats_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_diag(force=False, verbose=False, suffix="", defines=()):
    """libpats_amd_diag<suffix>.so = the production objects with the DIAG_SOURCES recompiled under their extra defines
    (+ `defines`, e.g. an experiment hook like -DPATS_EXP_SWAP_NOPS; _lib.py loads it with PATS_AMD_DIAG_LIB=<suffix>)."""
    build(force=force, verbose=verbose)
    lib_out = os.path.join(HERE, "libpats_amd_diag%s.so" % suffix)
    objdir = os.path.join(HERE, "build")
    objs = []
    for src in SOURCES:
        stem = os.path.splitext(src)[0]
        if src not in DIAG_SOURCES:
            objs.append(os.path.join(objdir, stem + ".o"))
            continue
        obj = os.path.join(objdir, stem + "_diag%s.o" % suffix)
        compile_tu(os.path.join(CSRC, src), obj, EXTRA_FLAGS.get(src, []) + DIAG_SOURCES[src] + list(defines), verbose)
        objs.append(obj)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib_out])
    return lib_out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        spath = os.path.join(CSRC, src)
        # every object depends on every header (a struct shared through a .hpp must never be seen in two layouts)
        headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + \
                  [os.path.join(HERE, "..", "include", "pats_amd.h")]
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(spath)
                and all(os.path.getmtime(obj) > os.path.getmtime(h) for h in headers)):
            continue
        jobs.append((spath, obj, EXTRA_FLAGS.get(src, [])))
    compile_many(jobs, verbose)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    subprocess.check_call(cmd)
    return LIB


# ---- the reference's native module, compiled: tensor_resize.cpython-*.so at the repo root (setup/setup.py:107-118
#      builds `CppExtension('tensor_resize', ['library.cpp'])`; this is the same module name over the C-ABI) ----------
EXT_SRC = os.path.join(CSRC, "binding", "tensor_resize_ext.cpp")


def ext_path():
    import sysconfig
    return os.path.join(os.path.dirname(HERE), "tensor_resize" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_tensor_resize_ext(force=False, verbose=False):
    """g++ on the pybind11 / libtorch binding, linked against libpats_amd.so (rpath $ORIGIN/pats_amd) and torch's own
    libraries.  Plain compiler call: the binding holds no device code, so neither hipcc nor ninja is needed."""
    import sysconfig
    out = ext_path()
    deps = [EXT_SRC, os.path.join(HERE, "..", "include", "pats_amd.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) > os.path.getmtime(d) for d in deps):
        return out
    import torch
    from torch.utils import cpp_extension as ce
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found - the tensor_resize extension cannot be built")
    tlib = ce.library_paths()[0]
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=tensor_resize",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for inc in ce.include_paths() + [sysconfig.get_paths()["include"], os.path.join(rocm, "include"),
                                     os.path.join(HERE, "..", "include")]:
        cmd += ["-isystem" if "torch" in inc or "rocm" in inc else "-I", inc]
    # torch's libraries FIRST in the DT_NEEDED order: they bring torch's own libamdhip64 (soname libamdhip64.so.7), which
    # libpats_amd.so's dependency of that soname then resolves to - one HIP runtime in the process even when this module
    # is imported before torch (the other order loads /opt/rocm's copy, then torch's beside it: see _lib.py)
    cmd += [EXT_SRC, "-o", out, "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
            "-ltorch_python", "-L" + HERE, "-l:libpats_amd.so",
            "-Wl,-rpath," + tlib, "-Wl,-rpath,$ORIGIN/pats_amd"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--diag" in sys.argv:
        print(build_diag(verbose=True))
    print(build_tensor_resize_ext(force="--force" in sys.argv, verbose=True))
