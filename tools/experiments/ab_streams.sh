# Round 5: train step (B = 64) with / without the MSG scales on their own stream -- one box, alternating.  (The class head on a stream
# of its own and the queued weight gradients issued on a stream of their own during the backward were measured with the same script
# and removed: HISTORY.md.)   bash tools/experiments/ab_streams.sh
for rep in 1 2 3; do for m in 1 0; do RTK_MSG=$m python - <<PY 2>/dev/null
import os, json, sys, io, contextlib, runpy
from ratrack_amd import train_path
train_path.MSG_STREAMS = os.environ["RTK_MSG"] == "1"
sys.argv = ["bench.py", "--mode", "train", "--no-cpu-baseline", "--traffic", "off"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("MSG_STREAMS", os.environ["RTK_MSG"], d["value"], "pairs/s", d["ms_per_step"], "ms/step")
PY
done; done
