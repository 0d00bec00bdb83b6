#!/bin/bash
# Ablation of cost_volume_bwd_kernel on the GPU box: build variants, time with bench.py's live kernel timer.
for flags in "" "-DCVB_NOSTORE"; do
  RTK_EXTRA_FLAGS="$flags" python ratrack_amd/build.py --force > /dev/null 2>&1
  echo "flags='$flags'"
  python - <<'PY'
import torch, bench
ms, fl = bench.time_cost_volume_bwd(64, 256, torch.device("cuda"))
print("  %.3f ms  %.1f TFLOP/s" % (ms, fl / ms / 1e9))
PY
done
python ratrack_amd/build.py --force > /dev/null 2>&1
