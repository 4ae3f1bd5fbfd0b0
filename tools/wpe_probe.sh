cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for so in "" gpurun_in/lib_wpe1.so; do
  if [ -z "$so" ]; then unset UNIDOM_HIP_SO; else export UNIDOM_HIP_SO=$GRAFT_REPO_ROOT/$so; fi
  for gc in 2 0; do
  echo "so=$so grid_ckpt=$gc envs16 $(UD_MPM_CLUSTER=1 timeout -k 10 120 python bench.py --workload whip_rope --n-grid 128 --envs 16 --grid-ckpt $gc --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -n 1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])')"
  done
done
