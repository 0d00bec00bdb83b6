#!/bin/bash
# Experiment: LDS footprint of the per-point chain kernel's weight stream (PW_F) vs end-to-end forward throughput (two batches in flight)
cd "$(dirname "$0")/.."
for flags in "-DPW_F=32" "-DPW_F=16" "-DPW_F=8"; do
  RTK_EXTRA_FLAGS="$flags" python -m ratrack_amd.build --force > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-train --no-irregular 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$flags', r['ms_per_step'], r['value'], 'cv in situ', r['roofline']['kernel_ms'])"
done
python -m ratrack_amd.build --force > /dev/null 2>&1
