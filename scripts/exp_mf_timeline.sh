#!/bin/bash
# Build the library with the MF timeline stamps (run where hipcc is; the .so travels with gpurun);
# afterwards: python -m neurec_amd.build rebuilds the product library.
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
from neurec_amd import build
build.FLAGS.append("-DNR_MF_TIMELINE")
build.build_extension(force=True, verbose=False)
PY
echo "debug library built"
