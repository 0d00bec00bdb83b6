#!/bin/bash
# Experiment: cost_volume_bwd_kernel register budget (2 waves/SIMD with spills vs 1 wave/SIMD without), see DESIGN.md section 4.5
set -e
cd "$(dirname "$0")/.."
for flags in "" "-DCVB_MIN_WAVES=1"; do
  RTK_EXTRA_FLAGS="$flags" python -m ratrack_amd.build --force > /dev/null 2>&1
  python - <<PY
import torch
from ratrack_amd import train_ops as T
ms, fl = T.time_cost_volume_bwd(64, 256, "cuda", 20)
print("flags '%s': %.4f ms  %.1f TFLOP/s" % ("$flags", ms, fl / ms / 1e9))
PY
done
python -m ratrack_amd.build --force > /dev/null 2>&1
