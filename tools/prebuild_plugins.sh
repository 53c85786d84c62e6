#!/bin/bash
# Build, on the CPU box, the library and every generated plug-in the GPU tests and the bench ask for
# (they are cached by digest under exmc_amd/lib/gen/ and travel with gpurun), so that no GPU-box
# minute is spent in hipcc. Tests are skipped where they would open the device.
set -o pipefail
cd "$(dirname "$0")/.."
python __graft_entry__.py || exit 1
EXMC_PREBUILD_PLUGINS=1 python -m pytest tests -m gpu -q -n ${1:-6} -p no:cacheprovider 2>&1 | tail -3
python - <<'PY'
import bench
for m in ("gen_eight_schools", "gen_sv", "gen_radon", "gen_logistic"):
    try:
        bench.make_spec(m)
    except SystemExit:
        pass
PY
