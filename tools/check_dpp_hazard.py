#!/usr/bin/env python3
"""Hazard check for hand-written DPP instructions (csrc/fit_quad.hip, row16_solve_kernel): on gfx9 a VGPR written by a VALU instruction may be
read through DPP two wait states later at the earliest, and inline assembly is invisible to the compiler's hazard recogniser (it inserts
the s_nop for its own DPP instructions only).  Compiles the translation unit to ISA and checks, for every `*_dpp` instruction of the
named kernel, that none of the instructions within two wait states in front of it (an instruction = one wait state, `s_nop N` = N + 1)
is a VALU instruction writing its DPP source (src0).
usage: python tools/check_dpp_hazard.py [csrc/fit_quad.hip] [kernel-name-substring]      exit 1 on a violation"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "python-wlsqm_amd", "csrc", "fit_quad.hip")
want = sys.argv[2] if len(sys.argv) > 2 else "row16_solve_kernel"
out = tempfile.mktemp(suffix=".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                       "-I" + os.path.join(ROOT, "python-wlsqm_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, src],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
os.unlink(out)
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l.split(":")[0])
end = start
while not lines[end].startswith(".Lfunc_end"):
    end += 1


def regs(op):
    m = re.match(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", op)
    return {int(m.group(1))} if m else set()


insts = []
for l in lines[start:end]:
    l = l.split(";")[0].strip()
    if not l or l.startswith(".") or l.endswith(":"):
        continue
    parts = l.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    insts.append((parts[0], ops, l))
bad = ndpp = 0
for k, (op, ops, text) in enumerate(insts):
    if "_dpp" not in op:
        continue
    ndpp += 1
    srcs = regs(ops[1].split()[0])
    states, b = 0, k - 1
    while b >= 0 and states < 2:
        pop, pops, ptext = insts[b]
        if pop == "s_nop":
            states += int(pops[0]) + 1
        else:
            if pop.startswith("v_") and pops and regs(pops[0].split()[0]) & srcs:
                bad += 1
                print("HAZARD: %s   <-   %s  (%d wait states apart)" % (text, ptext, states))
            states += 1
        b -= 1
print("%s: %d DPP instructions, %d hazards" % (want, ndpp, bad))
sys.exit(1 if bad else 0)
