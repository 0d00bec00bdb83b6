#!/bin/bash
# What keeps a split layer's group step above its 12 MFMAs (384 clocks)?  Builds instrumented copies of the forward cost volume
# (tools/experiments/cv_ticks.py --totals-only) against patched copies of split_mfma.h, each with ONE ingredient of the group step
# removed (results are wrong; only the clock counts matter).  Run on CPU (hipcc); then cv_ticks.py --so <lib> on the GPU.
set -e
cd "$(dirname "$0")/../.."
V=ratrack_amd/lib/variants
mkdir -p $V
abl() {   # name, python expression patching the header text s
  mkdir -p $V/abl_$1
  python - "$1" "$2" <<'PY'
import sys
name, expr = sys.argv[1], sys.argv[2]
s = open("ratrack_amd/csrc/split_mfma.h").read()
n = len(s)
s = eval(expr)
assert len(s) != n or name == "none", "patch did not apply"
open("ratrack_amd/lib/variants/abl_%s/split_mfma.h" % name, "w").write(s)
PY
  cp ratrack_amd/csrc/fused_split.hip $V/abl_$1/fused_split.hip
  python tools/experiments/cv_ticks.py --build --totals-only --src $V/abl_$1/fused_split.hip --out $V/librtk_abl_$1.so > /tmp/abl_$1.log 2>&1 || { tail -5 /tmp/abl_$1.log; exit 1; }
  echo built $1
}
if [ -z "$ONLY" ]; then
abl none 's'
abl nosplit 's.replace("if constexpr (s + 1 < SPLIT_KS) split3_word<GI % 4>(h[2 * (s + 1)], h[2 * (s + 1) + 1], b[(s + 1) & 1]);", "")'
abl noside 's.replace("        side.template at<GI>(h);\n", "")'
abl nodma 's.replace("if constexpr ((f0 % F) / 6 < split_issue_groups(F)) ws.template issue_part<(f0 % F) / 6, split_issue_groups(F)>();", "")'
abl nosync 's.replace("if constexpr (f0 % F == 0 && f0 != 0) ws.sync();", "").replace("if constexpr ((f0 % F) / 6 < split_issue_groups(F)) ws.template issue_part<(f0 % F) / 6, split_issue_groups(F)>();", "")'
abl nolds 's.replace("asm volatile(\"ds_read_b128 %0, %1 offset:%2\" : \"=v\"(r) : \"v\"(rd), \"n\"(FI * 1024));", "asm volatile(\"\" : \"=v\"(r) : \"v\"(rd), \"n\"(FI * 1024));")'
abl nogroupbar 's.replace("            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);\n            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);\n", "")'
fi
# spreading the weight-stream requests over more group steps of a chunk
abl ig3 's.replace("return F / 6 > 4 ? 3 : 2;", "return  F / 6 > 4 ? 3 : 3;")'
abl ig4 's.replace("return F / 6 > 4 ? 3 : 2;", "return  F / 6 > 4 ? 3 : 4;")'
