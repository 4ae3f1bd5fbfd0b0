"""Instruction statistics of the largest innermost loop of a kernel in a hipcc -save-temps .s file.
usage: python tools/asm_loop_stats.py file.s kernel_name_substring [--dump]"""
import re, sys
from collections import Counter

path, kname = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kname in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
blocks, cur = [], None
for l in body:
    t = l.strip()
    if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", t):
        m = re.search(r"Header=(BB\d+_\d+) Depth=(\d)", t)
        cur = {"head": t, "loop": (m.group(1), int(m.group(2))) if m else None, "ins": []}
        lab = re.match(r"^\.L(BB\d+_\d+):", t)
        cur["label"] = lab.group(1) if lab else None
        blocks.append(cur)
        continue
    if cur is None:
        continue
    if t.startswith(";"):
        if not cur["ins"]:
            m = re.search(r"Header=(BB\d+_\d+) Depth=(\d)", t)
            if m:
                cur["loop"] = (m.group(1), int(m.group(2)))
            m = re.search(r"This (Inner )?Loop Header: Depth=(\d)", t)
            if m and cur["label"]:
                cur["loop"] = (cur["label"], int(m.group(2)))
        continue
    if not t or t.startswith("."):
        continue
    cur["ins"].append(t)
sizes = Counter()
for b in blocks:
    if b["loop"]:
        sizes[b["loop"]] += len(b["ins"])
maxdepth = max(d for (_, d) in sizes)
best = max((k for k in sizes if k[1] == maxdepth), key=lambda k: sizes[k])
print("loops:", dict(sizes), "-> reporting", best)
tot = Counter(); n = 0
for b in blocks:
    if b["loop"] == best:
        n += len(b["ins"])
        tot.update(i.split()[0] for i in b["ins"])
        if "--dump" in sys.argv:
            print(b["head"])
            for i in b["ins"]:
                print("\t" + i)
print("inner-loop instructions:", n)
cls = Counter()
for k, v in tot.items():
    c = ("pk" if k.startswith("v_pk_") else "mov" if k.startswith(("v_mov", "v_accvgpr")) else "dpp/lane" if ("dpp" in k or "lane" in k) else
         "cndmask" if "cndmask" in k else "cmp" if k.startswith("v_cmp") else "trans" if k.startswith(("v_rsq", "v_rcp", "v_sqrt")) else
         "nop" if k == "s_nop" else "wait" if k == "s_waitcnt" else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else
         "vmem" if k.startswith(("global_", "buffer_")) else "valu")
    cls[c] += v
print(dict(cls.most_common()))
print(tot.most_common(14))
