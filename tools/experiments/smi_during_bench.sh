#!/bin/bash
# Round 5: board power and shader clock while the pipelined forward runs (is the launch power-limited?).
# usage (GPU box): bash tools/experiments/smi_during_bench.sh > gpurun_out/r05/smi_during_bench.txt
python bench.py --no-cpu-baseline --no-train --no-configs --traffic off --seconds 15 > /tmp/smi_bench.json 2>/tmp/smi_bench.err &
BP=$!
sleep 2
for i in $(seq 1 80); do
    if ! kill -0 $BP 2>/dev/null; then break; fi
    rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|GPU use" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 0.5
done
wait $BP
python - <<'PY'
import json
d = json.loads(open("/tmp/smi_bench.json").read().strip().splitlines()[-1])
print("bench:", d["value"], "pairs/s", d["ms_per_step"], "ms/step")
PY
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
