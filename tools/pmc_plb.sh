#!/bin/bash
# HBM traffic of the PlasticineLab path per bench step (all plb_* kernels of one timed step: 10 env.steps x 19 substeps, with --plb-grad also
# the loss and the adjoint): FETCH_SIZE and WRITE_SIZE in a pass of their own each, gfx950 read correction (MI355X_MICROARCH.md).
# usage (GPU box): bash tools/pmc_plb.sh   -> gpurun_out/pmc_plb/pmc_traffic.json entries plb:{fwd,grad}:ngrid{64,128}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_plb; rm -rf $O; mkdir -p $O
cp profiles/pmc_traffic.json $O/pmc_traffic.json
for NG in 64 128; do for MODE in fwd grad; do
  FL=""; [ $MODE = grad ] && FL="--plb-grad"
  for C in FETCH_SIZE WRITE_SIZE INSTS; do
    CN=$C; [ $C = INSTS ] && CN="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU"
    timeout -k 10 300 rocprofv3 --pmc $CN --kernel-trace -d $O/${MODE}_${NG}_$C -o p -f csv -- python3 bench.py --workload torus --n-grid $NG $FL --steps 2 --warmup 1 --no-cpu-baseline > $O/${MODE}_${NG}_$C.log 2>&1 || echo "pass $MODE $NG $C failed"
  done
  python3 - <<PY
import csv, json, sys
sys.path.insert(0, "tools")
import src_hash
tot, per_k = {}, {}
for c, f in (("FETCH_SIZE", "FETCH_SIZE"), ("WRITE_SIZE", "WRITE_SIZE"), ("SQ_WAVES", "INSTS"), ("SQ_INSTS_VALU", "INSTS"), ("SQ_INSTS_SALU", "INSTS")):
    s = 0.0
    for r in csv.DictReader(open("$O/${MODE}_${NG}_%s/p_counter_collection.csv" % f)):
        if r["Counter_Name"] == c and ("plb" in r["Kernel_Name"] or "pcl_" in r["Kernel_Name"]):
            s += float(r["Counter_Value"])
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ud::", "")
            per_k.setdefault(k, {}).setdefault(c, []).append(float(r["Counter_Value"]))
    tot[c] = s
steps = 3.0   # --warmup 1 --steps 2: a warm-up step is a whole bench step in both modes (bench.py::bench_torus)
per_step = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps
d = json.load(open("$O/pmc_traffic.json"))
d["plb:$MODE:ngrid$NG"] = {"hbm_bytes_per_launch": per_step, "src_sha16": src_hash.sha16("plb"),
    "insts": {k: tot[k] / steps for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU")},
    "note": "all plb_* / pcl_* kernels of one bench step (8 envs x 10 env.steps x 19 substeps" + (", + loss + adjoint" if "$MODE" == "grad" else "") +
            "): (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 summed over the pass / its 3 bench steps; separate --pmc passes (tools/pmc_plb.sh)"}
for k in ("pcl_fwd_kernel", "pcl_bwd_kernel"):       # the persistent kernels, per launch (= per step call): what the bench line's roofline.traffic quotes
    if k in per_k and ("$MODE" == "grad" or k == "pcl_fwd_kernel"):
        v = per_k[k]
        mean = lambda c: sum(v[c]) / len(v[c]) if v.get(c) else 0.0
        d["%s:$MODE:ngrid$NG" % k] = {"hbm_bytes_per_launch": (2 * mean("FETCH_SIZE") + mean("WRITE_SIZE")) * 1024, "src_sha16": src_hash.sha16("pcl_"),
            "launches_in_the_pass": len(v.get("FETCH_SIZE", [])),
            "insts": {c: mean(c) for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU")},
            "note": "mean per launch (one step call of 8 envs); 2 x FETCH_SIZE + WRITE_SIZE, KB x 1024, separate --pmc passes (tools/pmc_plb.sh)"}
        print("%s:$MODE:ngrid$NG  %.1f MB per launch" % (k, d["%s:$MODE:ngrid$NG" % k]["hbm_bytes_per_launch"] / 1e6))
json.dump(d, open("$O/pmc_traffic.json", "w"), indent=1)
print("plb:$MODE:ngrid$NG  %.3f GB per bench step" % (per_step / 1e9))
PY
  rm -rf $O/${MODE}_${NG}_FETCH_SIZE $O/${MODE}_${NG}_WRITE_SIZE $O/${MODE}_${NG}_INSTS
done; done
