"""Lint of hipcc's device assembly (-S) for the wait states the hazard recognizer does NOT insert because one side of the
dependency sits inside an inline-assembly block (the compiler treats INLINEASM as opaque: round 5's stale v_readfirstlane
count, DESIGN section 4 "inline assembly and wait states").  Checked on gfx950, with the wait states the compiler itself
uses between its own instructions (tools: a probe compiled with -S, round 6):

  A  MFMA result (compiler-emitted v_mfma, vDst in VGPRs)  ->  first read / write of a vDst register INSIDE an asm block:
     18 wait states for the 16-pass shapes (32x32x2 f32, 32x32x16 bf16 ...), 10 for 8-pass, 6 for 4-pass;
  B  MFMA inside an asm block  ->  compiler-emitted VALU / VMEM / LDS instruction touching its vDst afterwards: same counts
     (the k-loops end in s_nop 15 + s_nop 3 for this);
  C  VALU write of a VGPR (compiler-emitted)  ->  v_mfma INSIDE an asm block reading it as SrcA / SrcB / SrcC: 2;
  D  VALU write of a VGPR  ->  v_readfirstlane / v_readlane of it where either side is inside an asm block: 1;
  E  VALU write of a VGPR  ->  DPP read of it where either side is inside an asm block: 2.

A straight-line scan per function (labels and branches end a window conservatively: a window that reaches a branch or a
label before the requirement is met is reported as "unresolved" only when the consumer is found before the label).
usage: check_asm_hazards.py file.s [file2.s ...]      exit code 1 if a violation is found"""
import re
import sys

PASSES = {"4x4": 2, "16x16x4_f32": 8, "16x16x1": 8, "16x16": 8, "32x32": 16}


def mfma_wait(op):
    m = re.match(r"v_mfma_\w+?_(\d+x\d+x\d+)", op)
    shape = m.group(1) if m else ""
    if shape.startswith("32x32"):
        return 18
    if shape.startswith("16x16"):
        return 10
    return 6


def regs_of(tok):
    """VGPR numbers named by an operand token: v12, v[4:7]"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def parse(line):
    code = line.split(";")[0].strip()
    if not code or code.endswith(":") or code.startswith("."):
        return None
    parts = code.replace(",", " ").split()
    op, toks = parts[0], parts[1:]
    return op, toks


def is_valu(op):
    return op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_smfmac")


def wait_states(op, toks):
    if op == "s_nop":
        return int(toks[0], 0) + 1
    return 1


def scan(path):
    bad = []
    func, inasm = None, False
    insts = []          # (func, inasm, op, toks, lineno, is_label_or_branch)
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\S+|[A-Za-z_]\w*):\s*(;.*)?$", line)
        if m and not line.startswith(".L") and not line.startswith("L_"):
            func = m.group(1)
        if "ASMSTART" in line:
            inasm = True
            continue
        if "ASMEND" in line:
            inasm = False
            continue
        code = line.split(";")[0].strip()
        if re.match(r"^(\.L\w+|L_\w+):", code):
            insts.append((func, inasm, "<label>", [], ln))
            continue
        p = parse(line)
        if p:
            insts.append((func, inasm, p[0], p[1], ln))
    n = len(insts)
    for i, (f, ia, op, toks, ln) in enumerate(insts):
        # ---- A / B: MFMA -> consumer across the asm boundary
        if op.startswith("v_mfma") and toks:
            dst = regs_of(toks[0])
            need = mfma_wait(op)
            ws = 0
            for j in range(i + 1, min(n, i + 64)):
                f2, ia2, op2, toks2, ln2 = insts[j]
                if f2 != f or op2 == "<label>" or op2.startswith("s_cbranch") or op2 == "s_branch" or op2 == "s_endpgm":
                    break
                if ws >= need:
                    break
                touched = set().union(*[regs_of(t) for t in toks2]) if toks2 else set()
                if touched & dst and not op2.startswith("v_mfma") and ia2 != ia:
                    bad.append((path, f, ln, ln2, "MFMA (%s, line %d, %s) -> %s (%s) after %d of %d wait states" % (
                        op, ln, "asm" if ia else "compiler", op2, "asm" if ia2 else "compiler", ws, need)))
                    break
                if touched & dst and not op2.startswith("v_mfma"):
                    break            # same side of the boundary: the compiler's / the author's own business
                ws += wait_states(op2, toks2)
        # ---- C / D / E: VALU write -> MFMA / readlane / DPP read across the boundary
        if is_valu(op) and toks:
            dst = regs_of(toks[0])
            if not dst:
                continue
            ws = 0
            for j in range(i + 1, min(n, i + 4)):
                f2, ia2, op2, toks2, ln2 = insts[j]
                if f2 != f or op2 == "<label>":
                    break
                srcs = set().union(*[regs_of(t) for t in toks2[1:]]) if len(toks2) > 1 else set()
                need = 0
                if op2.startswith("v_mfma"):
                    need = 2
                elif op2.startswith("v_readfirstlane") or op2.startswith("v_readlane"):
                    need = 1
                elif "dpp" in op2 or any("quad_perm" in t or "row_" in t for t in toks2):
                    need = 2
                if need and (srcs & dst) and ws < need and (ia or ia2):
                    bad.append((path, f, ln, ln2, "VALU write (%s, line %d, %s) -> %s (%s) after %d of %d wait states" % (
                        op, ln, "asm" if ia else "compiler", op2, "asm" if ia2 else "compiler", ws, need)))
                ws += wait_states(op2, toks2)
                if ws >= 2:
                    break
    return bad


if __name__ == "__main__":
    allbad = []
    for p in sys.argv[1:]:
        allbad += scan(p)
    for path, f, a, b, msg in allbad:
        print("%s: %s: lines %d -> %d: %s" % (path, (f or "?")[:60], a, b, msg))
    print("violations:", len(allbad))
    sys.exit(1 if allbad else 0)
