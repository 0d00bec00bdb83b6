#!/bin/bash
# usage: tools/exp_cv.sh "<flags>"  -> builds a variant into ratrack_amd/lib and prints the cost-volume kernel time
set -e
RTK_EXTRA_FLAGS="$1" python ratrack_amd/build.py --force > /dev/null 2>&1
tools/kres.sh ratrack_amd/csrc/fused_group.hip 2>/dev/null | head -1
