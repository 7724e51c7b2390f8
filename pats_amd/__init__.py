"""pats_amd - MI355X (gfx950) implementation of the PATS patch-area optimal-transport hot path.

`pats_amd.ops` mirrors the reference's operator surface over the C-ABI of include/pats_amd.h
(libpats_amd.so, hand-written HIP).  Importing `pats_amd.ops` loads the library and raises if it
was not built - there is no CPU fallback.  `pats_amd.synth` holds the deterministic synthetic
workloads shared by tests and bench.py.
"""
__version__ = "0.1.0"
