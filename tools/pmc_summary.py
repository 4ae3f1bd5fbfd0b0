"""Summarise rocprofv3 --pmc passes into profiles/pmc_traffic.json (HBM bytes per launch of the ud:: kernels).

usage: python tools/pmc_summary.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> [out.json] [SQ_INSTS counter_collection.csv]
The optional fourth file (a pass with SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU) adds `insts`: mean wave-instructions per launch (bench.py: roofline.issue).
FETCH_SIZE / WRITE_SIZE are reported in KB; per MI355X_MICROARCH.md (HBM / rocprofv3 section) gfx950 tallies 128-B read
requests at 64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 as is.  Each counter comes from its own pass.
"""
import csv, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import src_hash


def load(path, counter):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or not r["Kernel_Name"].startswith("ud::"):
            continue
        per[r["Kernel_Name"].split("(")[0].replace("ud::", "")].append(float(r["Counter_Value"]))
    return per


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
insts = {c: load(sys.argv[4], c) for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU")} if len(sys.argv) > 4 else {}
dst = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
out = json.load(open(dst)) if os.path.exists(dst) else {}      # entries of kernels not in these passes are kept
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
    if k.startswith("void "):
        k = k[5:]
    out[k] = {"FETCH_SIZE": {"n": len(f), "mean_kb": sum(f) / len(f), "min_kb": min(f), "max_kb": max(f)},
              "WRITE_SIZE": {"n": len(w), "mean_kb": sum(w) / len(w), "min_kb": min(w), "max_kb": max(w)},
              "hbm_bytes_per_launch": (2 * max(f) + max(w)) * 1024,
              "src_sha16": src_hash.sha16(k),   # bench.py reports traffic only while the kernel sources still hash to this
              "note": "2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE, KB x 1024, separate --pmc passes; "
                      "max over launches (forward launches under no_grad write no checkpoints)"}
for c, per in insts.items():
    for k, vals in per.items():
        k = k[5:] if k.startswith("void ") else k
        if k in out and vals:
            out[k].setdefault("insts", {})[c] = max(vals)      # max over launches: the launches that write checkpoints (as for the bytes)
json.dump(out, open(dst, "w"), indent=1)
for k, v in out.items():
    if k not in fetch and k not in write:
        continue
    print(k, "%.1f MB/launch" % (v["hbm_bytes_per_launch"] / 1e6), "fetch n=%d" % v["FETCH_SIZE"]["n"])
