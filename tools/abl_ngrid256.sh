cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05q
for A in 0 1 64 128 256; do
  if [ $A = 0 ]; then unset UNIDOM_HIP_SO; else export UNIDOM_HIP_SO=$GRAFT_REPO_ROOT/gpurun_in/lib_abl$A.so; fi
  rm -rf gpurun_out/r05q/abl_$A; mkdir -p gpurun_out/r05q/abl_$A
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r05q/abl_$A -o p -f csv -- python3 bench.py --workload whip_rope --n-grid 256 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r05q/abl_$A/log 2>&1
  echo "ABLATE=$A $(python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/r05q/abl_$A/p_kernel_stats.csv")):
    n = r["Name"].split("(")[0].replace("void ud::","").replace("ud::","")
    if n.startswith("lg_") and float(r["AverageNs"]) > 4000 and int(r["Calls"]) > 50:
        print(n, "%.1f" % (float(r["AverageNs"])/1e3), end="  ")
PY
)"
done
