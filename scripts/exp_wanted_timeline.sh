#!/bin/bash
# Build the library with the timeline stamps, run the experiment, rebuild the product library.
# (Run where hipcc is: the build container; the .so travels to the GPU box with gpurun.)
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
from neurec_amd import build
build.FLAGS.append("-DNR_WW_TIMELINE")
build.build_extension(force=True, verbose=False)
PY
echo "debug library built"
