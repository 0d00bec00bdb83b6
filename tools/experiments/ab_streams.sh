# Round 5: train step (B = 64) A/B of stream options -- one box, alternating.   bash tools/experiments/ab_streams.sh
for rep in 1 2 3; do for m in 1 0; do RTK_OPT=$m python - <<PY 2>/dev/null
import os, json, sys, io, contextlib, runpy
from ratrack_amd import train_path, train_ops
train_ops.WGRAD_TWO_STREAMS = os.environ["RTK_OPT"] == "1"
sys.argv = ["bench.py", "--mode", "train", "--no-cpu-baseline", "--traffic", "off"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("WGRAD_TWO_STREAMS", os.environ["RTK_OPT"], d["value"], "pairs/s", d["ms_per_step"], "ms/step")
PY
done; done
