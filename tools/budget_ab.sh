#!/bin/bash
# tools/pmc_budget.sh on the default library and on a timing-only ablation build (tools/build_abl.sh <bits> -> gpurun_in/lib_abl<bits>.so), side by side for the
# three one-lane particle kernels of the rope at n_grid 256.   usage (GPU box): ABL=1 bash tools/budget_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ABL=${ABL:-1}
W=whip_rope ARGS="--n-grid 256" NAME=whip_rope_ngrid256 bash tools/pmc_budget.sh > gpurun_out/budget_default.txt 2>&1
cp gpurun_out/pmc_budget_whip_rope_ngrid256.csv gpurun_out/pmc_budget_default.csv
UNIDOM_HIP_SO=$PWD/gpurun_in/lib_abl$ABL.so W=whip_rope ARGS="--n-grid 256" NAME=whip_rope_ngrid256_abl$ABL bash tools/pmc_budget.sh > gpurun_out/budget_abl$ABL.txt 2>&1
python3 - <<PY
import csv
for f in ("gpurun_out/pmc_budget_default.csv", "gpurun_out/pmc_budget_whip_rope_ngrid256_abl$ABL.csv"):
    print(f)
    for r in csv.DictReader(open(f)):
        if r["kernel"] in ("lg_p2g_adj<1>", "lg_g2p_p2g<1>", "lg_g2p_adj<1>"):
            g = lambda k: float(r[k]) if r.get(k) else float("nan")
            w = g("SQ_WAVES")
            print("  %-16s waves %5.0f valu/wave %6.0f salu/wave %6.0f lds/wave %5.0f vmem_rd/wave %5.0f | wave_cycles/wave %8.0f busy_cycles %9.0f | active_any %4.2f wait_inst %4.2f wait_any %4.2f active_valu %4.2f" % (
                r["kernel"], w, g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_LDS") / w, g("SQ_INSTS_VMEM_RD") / w, g("SQ_WAVE_CYCLES") / w, g("SQ_BUSY_CYCLES"),
                g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
PY
