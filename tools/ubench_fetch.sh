#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (tools/ubench_fetch.hip) -> gpurun_out/ubench_fetch.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ubench_fetch; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace -d $O/$C -o p -f csv -- tools/ubench_fetch > $O/$C.log 2>&1 || echo "pass $C failed"
done
# the raw request counters behind FETCH_SIZE (counter_defs.yaml, gfx950: TCC_BUBBLE * 128 + (RDREQ - BUBBLE - RDREQ_32B) * 64 + RDREQ_32B * 32) and the
# DRAM-side 32-byte tally, which counts a 64-B request twice and a 128-B request four times -- an exact byte count if it works on this part
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace -d $O/RAW -o p -f csv -- tools/ubench_fetch > $O/RAW.log 2>&1 || echo "pass RAW failed"
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace -d $O/DRAM -o p -f csv -- tools/ubench_fetch > $O/DRAM.log 2>&1 || echo "pass DRAM failed"
python3 - <<PY
import csv
vals = {}
import os
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open("$O/%s/p_counter_collection.csv" % c)):
        k = r["Kernel_Name"].split("(")[0]
        if r["Counter_Name"] == c:
            if k.startswith("args"): vals.setdefault(k, {}).setdefault(c + "_list", []).append(float(r["Counter_Value"]))
            else: vals.setdefault(k, {})[c] = float(r["Counter_Value"])
for d in ("RAW", "DRAM"):
    f = "$O/%s/p_counter_collection.csv" % d
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            vals.setdefault(r["Kernel_Name"].split("(")[0], {})[r["Counter_Name"]] = float(r["Counter_Value"])
GiB = 1 << 30
useful = {"read16": GiB, "read4": GiB, "read4q": GiB, "gather16": GiB // 4, "write16": GiB, "write4q": GiB, "atomic4": GiB, "scatter16": GiB // 8}
print("# tools/ubench_fetch.sh on 1x MI355X: counter KB x 1024 / useful bytes (1 GiB per kernel; gather16: 256 MiB useful, one 16-B cell per 64 B; scatter16: 128 MiB useful, one 16-B cell per 128 B)")
print("%-10s %14s %14s %10s %10s" % ("kernel", "FETCH_SIZE KB", "WRITE_SIZE KB", "fetch/use", "write/use"))
for k, u in useful.items():
    v = vals.get(k, {})
    f, w = v.get("FETCH_SIZE", float("nan")), v.get("WRITE_SIZE", float("nan"))
    print("%-10s %14.0f %14.0f %10.3f %10.3f" % (k, f, w, f * 1024 / u, w * 1024 / u))
for k in ("args1536", "args96"):
    v = vals.get(k, {})
    f, w = v.get("FETCH_SIZE_list", [0]), v.get("WRITE_SIZE_list", [0])
    nb = int(k[4:])
    print("%-10s empty kernel with a 1 KB by-value argument, %4d blocks x 256 threads, mean of %d launches: FETCH_SIZE %.1f KB (x 2 = %.0f B per block), WRITE_SIZE %.1f KB (%.0f B per block)"
          % (k, nb, len(f), sum(f) / len(f), 2 * 1024 * sum(f) / len(f) / nb, sum(w) / len(w), 1024 * sum(w) / len(w) / nb))
print("# raw read-request counters per kernel (requests; DRAM_32B x 32 B / useful bytes in the last column)")
names = ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_DRAM_32B_sum"]
print("%-10s " % "kernel" + " ".join("%26s" % n for n in names) + "  dram32x32/use")
for k, u in useful.items():
    v = vals.get(k, {})
    print("%-10s " % k + " ".join("%26.0f" % v.get(n, float("nan")) for n in names) + "  %10.3f" % (v.get("TCC_EA0_RDREQ_DRAM_32B_sum", float("nan")) * 32 / u))
PY
