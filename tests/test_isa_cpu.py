"""CPU: a lint on the generated gfx950 code of every kernel file (hipcc cross-compiles without a GPU).

Round 4 traced run-to-run differences of the furthest-point selection to ONE instruction form; round 5 reproduced it outside the
library in two seconds of GPU time (tools/micro/opsel_hazard.hip, profiles/r05_opsel_hazard_standalone.txt): a two-operand packed
fp32 operation whose SECOND source takes the odd half of its register pair for the low lane -- `v_pk_add_f32 / v_pk_mul_f32 v[a:b],
v[c:d], v[e:f] op_sel:[0,1]` -- reads a wrong value in EVERY workgroup as soon as another wave on its SIMD runs a matrix
instruction whose result a vector instruction reads (any layer followed by its epilogue).  Wait states do not help, the register
bank does not matter, a VALU copy of the pair fails the same.  Measured SAFE under the same noise: the low-half broadcast
`op_sel_hi:[1,0]`, the pair in the FIRST source slot (`op_sel:[1,0]`), `v_pk_fma_f32 ... op_sel:[0,0,1]`, and the 16-bit half
selections of `v_fma_mix_f32 / v_fma_mixlo_f16 / v_fma_mixhi_f16` (what the fp16 split layers use).  The lint is deliberately wider
than the measured fault: NO packed fp32 instruction in the library selects a half through op_sel, in any source slot."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def op_sel_half_selections(asm_text):
    """The `v_pk_*_f32` instructions of an assembly listing whose op_sel list has a 1 in ANY source slot (two- and three-operand forms)."""
    packed = re.findall(r"^\s*(v_pk_\w+_f32 .*)$", asm_text, flags=re.M)
    bad = []
    for ins in packed:
        m = re.search(r"op_sel:\[([01,]+)\]", ins)
        if m and "1" in m.group(1):
            bad.append(ins)
    return packed, bad


def test_the_lint_fires_on_every_spelling_of_the_form():
    listing = """
	v_pk_add_f32 v[32:33], v[20:21], v[14:15] op_sel:[0,1]
	v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0] op_sel_hi:[1,1]
	v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1] neg_lo:[0,0,1]
	v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]
	v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[1,0]
	v_pk_mul_f32 v[0:1], v[2:3], v[4:5]
	v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]
	v_fma_mixhi_f16 v3, v1, s2, -v2 op_sel:[0,0,1] op_sel_hi:[0,0,1]
"""
    packed, bad = op_sel_half_selections(listing)
    assert len(packed) == 7                                 # (the fma_mix line is not a packed fp32 instruction)
    assert len(bad) == 4 and all("op_sel:[" in b for b in bad)
    assert not any("op_sel_hi:[1,0]" in b and "op_sel:[" not in b for b in bad)


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
        packed, bad = op_sel_half_selections(open(out).read())
        n_packed[os.path.basename(src)] = len(packed)
        assert not bad, "%s: packed fp32 instructions that read half of a pair through op_sel:\n  %s" % (src, "\n  ".join(bad[:8]))
    # the selection kernels' packed distance arithmetic is still there (the lint is not vacuous)
    assert n_packed["ops_pointnet2.hip"] >= 12, n_packed
