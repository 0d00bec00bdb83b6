"""Which B = 64, N = 256 synthetic batch has the fewest flipped ReLU / max-pool decisions between this path, the fp32 oracle and float64?
Per seed: the gradient-parity summary of tests/test_fullsize_oracle_gpu.py (one GPU train step + two CPU oracle steps, ~1-2 min of host
time each).  python tools/experiments/scan_fullsize_seed.py 2031 2032 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_fullsize_oracle_gpu as T

# --small B: scan batches of B distinct pairs (the flip-free ones, eight times over, make the B = 64 case of the test)
small = 0
args = sys.argv[1:]
if args and args[0] == "--small":
    small, args = int(args[1]), args[2:]
for seed in [int(x) for x in args] or list(range(2031, 2037)):
    try:
        rows, e_arb, floor = T._train_step_parity(seed, B=small or 64)
        print("RESULT seed %d  this path vs float64: median %.2e p90 %.2e max %.2e | fp32 oracle vs float64: median %.2e p90 %.2e max %.2e"
              % (seed, np.median(e_arb), np.quantile(e_arb, 0.9), e_arb.max(), np.median(floor), np.quantile(floor, 0.9), floor.max()), flush=True)
    except AssertionError as e:
        print("RESULT seed %d  assertion: %r" % (seed, str(e)[:200]), flush=True)
