#!/bin/bash
# quick GPU loop for the soft-contact / many-workgroup MPM work: exactness tool, the MPM + env test files, then the named bench lines
# usage (on the GPU box): TAG=r05b bash tools/quick_mpm.sh pour_water pour_soup shape_rope ...   (NOTEST=1 skips pytest)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-quick}; O=gpurun_out/$TAG; mkdir -p $O
if [ -z "$NOTEST" ]; then
  [ -x tools/check_exact_div ] && { timeout -k 10 300 tools/check_exact_div ${DIVLG:-33} > $O/check_exact_div.txt 2>&1; echo "check_exact_div rc=$?"; cat $O/check_exact_div.txt; }
  timeout -k 10 1000 python -m pytest ${TESTS:-tests/test_mpm_gpu.py tests/test_mpm_det.py tests/test_envs_gpu.py} -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
  tail -2 $O/tests.log
fi
for name in "$@"; do
  case $name in
    whip_rope_ngrid128) args="--workload whip_rope --n-grid 128";;
    whip_rope_ngrid256) args="--workload whip_rope --n-grid 256";;
    *) args="--workload $name";;
  esac
  timeout -k 10 300 python bench.py $args --no-cpu-baseline $EXTRA 2>$O/$name.err | tail -n 1 > $O/bench_line_$name.json
  python - <<PY
import json
d = json.load(open("$O/bench_line_$name.json"))
print("$name", round(d["value"]), "%.2f ms" % d["ms_per_step"], "frac %.4f" % d["roofline"]["frac"], d["roofline"].get("kernel_ms"), "traffic", d["roofline"].get("traffic"))
PY
  if [ -n "$KS" ]; then
    rm -rf $O/ks_$name; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ks_$name -o p -f csv -- python3 bench.py $args --no-cpu-baseline $EXTRA > $O/ks_$name.log 2>&1 \
      && cp $O/ks_$name/p_kernel_stats.csv $O/kernel_stats_$name.csv && rm -rf $O/ks_$name && head -12 $O/kernel_stats_$name.csv | cut -d, -f1-4
  fi
done
