#!/usr/bin/env python
"""Static instruction account of the headline kernel by phase.  Input: the assembly of scripts/ubench/headline_isa.hip compiled with -gline-tables-only
(.loc directives).  Every instruction is attributed to the source line it came from (innermost inlined location as the assembler sees it) and the lines
to phases; prints VALU / FP64 / SALU / LDS / VMEM per phase.   python scripts/isa_account.py /tmp/headline.s"""
import collections
import re
import sys

path = sys.argv[1]
files = {}
cur_file, cur_line = None, None
counts = collections.defaultdict(lambda: collections.Counter())

PH = [  # (file substring, first line, last line, phase) — first match wins; line numbers of the files in diffsol_amd/csrc at this commit
]

def classify(fname, line):
    f = fname.rsplit("/", 1)[-1]
    if f == "diffsol_detpow.h": return "pow (dsh_det_pow)"
    if f == "dsh_lu_dev.hpp": return "LU factor / solve (lane)"
    if f in ("dsh_models.hpp", "dsh_models_lane.hpp"): return "model rhs / Jacobian"
    if f == "dsh_device.hpp": return "wavefront reductions (DPP)"
    if f == "dsh_resident.hpp": return "dsh_resident.hpp:" + str(line)
    if f == "dsh_adaptive_kernel.hpp": return "dsh_adaptive_kernel.hpp:" + str(line)
    return f

def kind(op):
    if op.startswith("v_"):
        if "f64" in op: return "valu_f64"
        return "valu_other"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "scratch_", "flat_", "buffer_")): return "vmem"
    return "other"

# phases of the kernel's own source: '// @phase name' markers in dsh_adaptive_kernel.hpp open a phase that lasts until the next marker
import os
PHASES = []
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffsol_amd", "csrc", "dsh_adaptive_kernel.hpp")
marks = [(i + 1, l.split("@phase", 1)[1].strip()) for i, l in enumerate(open(src)) if "// @phase" in l]
for (a, name), nxt in zip(marks, marks[1:] + [(10 ** 9, "")]): PHASES.append((a, nxt[0] - 1, name))
in_kernel = False
for raw in open(path):
    line = raw.rstrip("\n")
    m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", line)
    if m:
        files[int(m.group(1))] = (m.group(2) + "/" + m.group(3)) if m.group(3) else m.group(2)
        continue
    if re.match(r"^_ZN3dsh14k_bdf_adaptive.*:", line): in_kernel = True; continue
    if in_kernel and "s_endpgm" in line: in_kernel = False
    if not in_kernel: continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur_file, cur_line = files.get(int(m.group(1)), "?"), int(m.group(2))
        continue
    m = re.match(r"\s+([a-z_0-9]+)\s", line + " ")
    if not m or line.lstrip().startswith((";", ".")): continue
    counts[classify(cur_file or "?", cur_line)][kind(m.group(1))] += 1

# fold the kernel's own lines into named phases
def phase_of(key):
    if not key.startswith(("dsh_adaptive_kernel.hpp:", "dsh_resident.hpp:")): return key
    f, l = key.split(":"); l = int(l)
    if f == "dsh_resident.hpp":
        return "resident helpers (norms, convergence, pi controller, consistent init, root finder)"
    table = PHASES
    for a, b, name in table:
        if a <= l <= b: return name
    return key
tot = collections.defaultdict(lambda: collections.Counter())
for k, c in counts.items():
    for kk, v in c.items(): tot[phase_of(k)][kk] += v
print(f"{'phase':90s} {'VALU':>6s} {'f64':>6s} {'other':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s}")
for name, c in sorted(tot.items(), key=lambda kv: -(kv[1]['valu_f64'] + kv[1]['valu_other'])):
    v = c['valu_f64'] + c['valu_other']
    print(f"{name[:90]:90s} {v:6d} {c['valu_f64']:6d} {c['valu_other']:6d} {c['salu']:6d} {c['lds']:5d} {c['vmem']:5d}")
allv = sum(c['valu_f64'] + c['valu_other'] for c in tot.values())
print(f"{'total (static)':90s} {allv:6d}")
