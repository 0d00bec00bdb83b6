#!/bin/bash
# Experiment: LDS footprint of the cost-volume weight stream (CV_F fragments per half of the double buffer) vs end-to-end forward
# throughput with two batches in flight -- does a smaller footprint let the other batch's kernels co-reside?
cd "$(dirname "$0")/.."
for flags in "-DCV_F=32" "-DCV_F=16" "-DCV_F=8"; do
  RTK_EXTRA_FLAGS="$flags" python -m ratrack_amd.build --force > /dev/null 2>&1
  python bench.py --no-cpu-baseline --no-train --no-irregular 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$flags', r['ms_per_step'], r['value'], 'cv in situ', r['roofline']['kernel_ms'])"
done
python -m ratrack_amd.build --force > /dev/null 2>&1
