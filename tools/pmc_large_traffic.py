"""HBM bytes of ONE ud_mpm_step_fwd / ud_mpm_step_bwd call of the many-workgroup MPM path, from the per-kernel summary of
tools/pmc_large.sh (mean FETCH_SIZE / WRITE_SIZE per launch in KB, separate --pmc passes): sum over the kernels of a direction of
launches-per-call x (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md).
usage: python tools/pmc_large_traffic.py <summary.csv> <name> [out.json]  -> entries large_path:<name>:fwd / :bwd"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import src_hash

FWD = ("lg_clear_fk", "lg_p2g<", "lg_grid", "lg_g2p<", "lg_g2p_p2g", "lg_sort", "lg_pack", "lg_unpack", "lg_fwd_out", "lg_prim_in", "lg_fk_all", "clm_fwd_kernel")
BWD = ("lg_restore", "lg_g2p_adj", "lg_grid_adj", "lg_p2g_adj", "lg_gadj_restore", "lg_padj_gadj", "lg_bwd_in", "lg_bwd_norm", "lg_bwd_out", "clm_bwd_kernel")
rows = list(csv.DictReader(open(sys.argv[1])))
name = sys.argv[2]
dst = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
by = {r["kernel"]: r for r in rows}
n_fwd = int(by["lg_fwd_out"]["launches"])
n_bwd = int(by["lg_bwd_out"]["launches"])
tot = {"fwd": 0.0, "bwd": 0.0}
insts = {"fwd": {"SQ_INSTS_VALU": 0.0, "SQ_INSTS_SALU": 0.0, "SQ_WAVES": 0.0}, "bwd": {"SQ_INSTS_VALU": 0.0, "SQ_INSTS_SALU": 0.0, "SQ_WAVES": 0.0}}
for r in rows:
    k = r["kernel"]
    if k.startswith("lg_grid_adj") or any(k.startswith(p) for p in BWD):
        d, n = "bwd", n_bwd
    elif any(k.startswith(p) for p in FWD):
        # lg_pack also packs the cotangents (one launch per backward call), lg_clear_fk also runs in a recomputing backward: billed to the forward
        d, n = "fwd", n_fwd
    else:
        continue
    for c in insts[d]:          # wave-instructions per step call: mean per launch x launches per call
        if r.get(c):
            insts[d][c] += float(r[c]) * int(r["launches"]) / n
    if not r["FETCH_SIZE"] or not r["WRITE_SIZE"]:
        continue
    tot[d] += (2 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024 * int(r["launches"]) / n
out = json.load(open(dst)) if os.path.exists(dst) else {}
for d in ("fwd", "bwd"):
    out[f"large_path:{name}:{d}"] = {
        "hbm_bytes_per_launch": tot[d],
        "note": f"sum over the kernels of one ud_mpm_step_{d} call of 2 x FETCH_SIZE + WRITE_SIZE (KB x 1024, mean per launch x launches per call), "
                f"from {os.path.basename(sys.argv[1])} (tools/pmc_large.sh: one counter group per pass, gfx950 read correction)",
        "insts": insts[d],     # summed over the kernels of the call (bench.py: roofline.issue)
        "src_sha16": src_hash.sha16("large_path"), "step_calls_in_the_passes": {"fwd": n_fwd, "bwd": n_bwd}}
    print(f"large_path:{name}:{d}: {tot[d] / 1e9:.3f} GB per step call")
json.dump(out, open(dst, "w"), indent=1)
