#!/usr/bin/env python3
"""Static instruction budget of one kernel by source region ("where do the 4 100 lane-instructions per particle-substep go").
Input: `llvm-objdump -d -l` of the gfx950 code object built with -gline-tables-only (see the usage line).  Every instruction is attributed to the
source line llvm gives it (the innermost inlined frame); lines are grouped into the regions below by file and line range.  The count is STATIC (each
instruction once): multiply a region by its trip count yourself -- the table prints the static count, the share, and the split VALU / SALU / LDS /
VMEM, which is what tells arithmetic from address and control overhead.
usage: python tools/isa_budget.py build/dbg/mpm_large.dis '_ZN2ud10lg_g2p_p2gILi1EEEvNS_9LargeArgsEPf' """
import collections
import re
import sys

dis, sym = sys.argv[1], sys.argv[2]
lines = open(dis).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <" + re.escape(sym) + r">:", l))
cur = ("?", 0)
per_line = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
for l in lines[start + 1:]:
    if re.match(r"^[0-9a-f]+ <", l):
        break
    m = re.match(r"^; (\S+):(\d+)", l)
    if m:
        cur = (m.group(1).split("/")[-1], int(m.group(2)))
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", l)
    if not m:
        continue
    op = m.group(1)
    kind = "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
    per_line[cur] += 1
    kinds[cur][kind] += 1
tot = sum(per_line.values())
by_file = collections.defaultdict(list)
for (f, ln), n in per_line.items():
    by_file[f].append((ln, n))
print(f"{sym}: {tot} instructions (static)")
for f in sorted(by_file, key=lambda f: -sum(n for _, n in by_file[f])):
    fl = sorted(by_file[f])
    print(f"  {f}: {sum(n for _, n in fl)} ({100 * sum(n for _, n in fl) / tot:.0f} %)")
    # contiguous line clusters (gap > 12 lines starts a new one)
    cl = []
    for ln, n in fl:
        if cl and ln - cl[-1][1] <= 12:
            cl[-1][1] = ln; cl[-1][2] += n
            for k, v in kinds[(f, ln)].items():
                cl[-1][3][k] += v
        else:
            cl.append([ln, ln, n, collections.Counter(kinds[(f, ln)])])
    for a, b, n, k in sorted(cl, key=lambda c: -c[2])[:14]:
        print(f"     lines {a:5d}-{b:<5d} {n:6d} ({100 * n / tot:4.1f} %)   " + "  ".join(f"{kk} {vv}" for kk, vv in k.most_common()))
