"""Static side of the inline-assembly hazard class: the compiler's device assembly of the
translation units that mix hand-written assembly with compiler-emitted MFMAs is linted for dependencies across an asm
boundary that the hazard recognizer cannot see (tools/check_asm_hazards.py).  CPU test: hipcc cross-compiles gfx950
without a GPU; ~20 s.  Round 6 found 36 sites here (K-tail MFMAs -> relu_bits' v_cmp on the accumulator after 0 - 3 of
18 wait states); they must stay at zero."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pointsecguard_amd", "csrc")


def _asm_of(unit, out_dir):
    cmd = subprocess.run(["make", "-n", "-B", unit + ".o"], cwd=CSRC, capture_output=True, text=True, check=True).stdout
    line = next(l for l in cmd.splitlines() if "hipcc" in l and " -c " in l)
    out = os.path.join(out_dir, unit + ".s")
    line = line.replace(" -c ", " -S --cuda-device-only -c ").replace("-o %s.o" % unit, "-o " + out)
    subprocess.run(line, shell=True, cwd=CSRC, check=True, capture_output=True)
    return out


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_unseen_wait_states_at_asm_boundaries(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_hazards
    files = [_asm_of(u, str(tmp_path)) for u in ("psg_pn2", "psg_ops")]
    n_asm = sum(len(re.findall("ASMSTART", open(f).read())) for f in files)
    assert n_asm > 1000            # the k-loops and the per-register ReLU blocks are really in there
    bad = [b for f in files for b in check_asm_hazards.scan(f)]
    assert not bad, bad[:5]


def test_the_lint_sees_a_planted_hazard(tmp_path):
    """the lint itself: a compiler MFMA followed by an asm consumer of its result without wait states is reported, the same with
    18 wait states between is not"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_hazards
    body = "_Z1kv:\n\tv_mfma_f32_32x32x2_f32 v[0:15], v16, v17, v[0:15]\n%s\t;;#ASMSTART\n\tv_cmp_lt_f32 vcc, 0, v15\n\t;;#ASMEND\n\ts_endpgm\n"
    p = tmp_path / "a.s"
    p.write_text(body % "")
    assert len(check_asm_hazards.scan(str(p))) == 1
    p.write_text(body % "\ts_nop 15\n\ts_nop 1\n")
    assert check_asm_hazards.scan(str(p)) == []
    p.write_text("_Z1kv:\n\tv_add_u32 v1, v2, v3\n\t;;#ASMSTART\n\tv_readfirstlane_b32 s0, v1\n\t;;#ASMEND\n\ts_endpgm\n")
    assert len(check_asm_hazards.scan(str(p))) == 1
