#!/usr/bin/env bash
# Builds the reference's ONLY native component, setup/library.cpp (the `tensor_resize`
# torch extension, /root/reference/setup/library.cpp:47-66,92-93), unmodified, from where it
# lies, into oracle/_ref/.  Test infrastructure only: used to validate oracle/pats_oracle.c,
# to generate tests/golden/*, and as the `cpu_baseline.kind == "reference"` leg for the
# subdivision gather.  Never imported by the product path (pats_amd/).
#
# The reference's own build (setup/setup.py:107-118) is a torch CppExtension; this recipe is
# the same compile issued directly with g++ (no reference build system is run).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="${PATS_REFERENCE_ROOT:-/root/reference}/setup/library.cpp"
OUT="$HERE/_ref"
if [ ! -f "$SRC" ]; then
  echo "[oracle/_ref] reference source $SRC not present (GPU box?) - keeping prebuilt files" >&2
  exit 0
fi
mkdir -p "$OUT"
PY=${PYTHON:-python3}
read -r TORCH_INC TORCH_LIB PY_INC EXT_SUFFIX CXX11_ABI < <($PY - <<'PYEOF'
import sysconfig, torch, os
from torch.utils import cpp_extension as ce
inc = " ".join("-I" + p for p in ce.include_paths())
print(inc.replace(" ", ","), os.path.join(os.path.dirname(torch.__file__), "lib"),
      sysconfig.get_paths()["include"], sysconfig.get_config_var("EXT_SUFFIX"),
      int(torch._C._GLIBCXX_USE_CXX11_ABI))
PYEOF
)
TORCH_INC="${TORCH_INC//,/ }"
TARGET="$OUT/tensor_resize$EXT_SUFFIX"
if [ "$TARGET" -nt "$SRC" ]; then echo "[oracle/_ref] up to date: $TARGET"; exit 0; fi
g++ -O2 -std=c++17 -fPIC -shared -w $TORCH_INC -I"$PY_INC" \
    -DTORCH_EXTENSION_NAME=tensor_resize -DTORCH_API_INCLUDE_EXTENSION_H \
    -D_GLIBCXX_USE_CXX11_ABI=$CXX11_ABI \
    "$SRC" -o "$TARGET" \
    -L"$TORCH_LIB" -Wl,-rpath,"$TORCH_LIB" -ltorch_python -ltorch -ltorch_cpu -lc10
echo "[oracle/_ref] built $TARGET"
