#!/usr/bin/env python3
"""Round 5: the B = 64 train step's gradient distance from float64 (tests/test_fullsize_oracle_gpu.py) with the cost volume on the fp16
split path and on the fp32-input MFMA kernels, same batch: are the differences flipped decisions or the split?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fullsize_oracle_gpu as T
from ratrack_amd import train_ops
for split in (True, False):
    train_ops.CV_SPLIT = split
    print("==== cost volume on the %s ====" % ("fp16 split path" if split else "fp32-input MFMA kernels"), flush=True)
    try:
        T.test_train_step_matches_oracle_at_full_size()
        print("passed")
    except AssertionError as e:
        print("assertion:", str(e)[:300])
