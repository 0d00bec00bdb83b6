"""CPU: a lint on the generated gfx950 code of the geometry kernels (hipcc cross-compiles without a GPU).

Round 4 traced run-to-run differences of the furthest-point selection to ONE instruction form: a packed fp32 operation that
broadcasts the odd half of a register pair through its op_sel modifier (`v_pk_add_f32 v[a:b], v[c:d], v[14:15] op_sel:[0,1]`) read a
wrong value about once in 10^4 executions whenever waves of another batch's split-bf16 matrix kernels shared the SIMD (DESIGN
section 8; tools/hazard_fps.py is the stress that shows it on a GPU, tests/test_hazard_gpu.py the regression test).  The source now
pins its broadcasts in register pairs of their own; this test keeps the compiler from quietly bringing the form back into
csrc/ops_pointnet2.hip, whose kernels decide INDICES -- where one wrong read changes the result instead of its last bit."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_reads_half_of_a_pair_through_op_sel_in_packed_fp32_arithmetic(tmp_path):
    """Every csrc/*.hip file, compiled for gfx950 with the flags it is BUILT with (ratrack_amd/build.flags_for: the SLP vectoriser,
    which produced 490 such instructions in the training kernels, is off for the files that need it; the selection kernels'
    hand-written float2 arithmetic pins its broadcasts): no `v_pk_*_f32` instruction with an op_sel half-selection anywhere."""
    from ratrack_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    procs = []
    for src in B.sources():
        out = str(tmp_path / (os.path.basename(src)[:-4] + ".s"))
        cmd = [hipcc] + [f for f in B.flags_for(src) if f != "-fPIC"] + ["-I", os.path.join(ROOT, "include"), "-I", B.CSRC, "-S", "--cuda-device-only", "-o", out, src]
        procs.append((src, out, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    n_packed = {}
    for src, out, p in procs:
        assert p.wait() == 0, "hipcc -S failed on %s" % src
        packed = re.findall(r"^\s*(v_pk_\w+_f32 .*)$", open(out).read(), flags=re.M)
        n_packed[os.path.basename(src)] = len(packed)
        bad = [ins for ins in packed if re.search(r"op_sel:\[(0,1|1,0|1,1)", ins)]
        assert not bad, "%s: packed fp32 instructions that read half of a pair through op_sel:\n  %s" % (src, "\n  ".join(bad[:8]))
    # the selection kernels' packed distance arithmetic is still there (the lint is not vacuous)
    assert n_packed["ops_pointnet2.hip"] >= 12, n_packed
