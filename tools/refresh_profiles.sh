set -x
mkdir -p gpurun_out/r01
timeout 300 python bench.py > gpurun_out/r01/sd15_bench.json 2> gpurun_out/r01/sd15_bench.err
timeout 400 python bench.py --model sdxl > gpurun_out/r01/sdxl_bench.json 2> gpurun_out/r01/sdxl_bench.err
cd /tmp && export TMPDIR=/tmp
for m in sd15 sdxl; do
  rm -rf /tmp/prof_$m
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -- python $GRAFT_REPO_ROOT/bench.py --model $m --steps 8 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r01/${m}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r01/${m}_rocprof.err
  db=$(find /tmp/prof_$m -name '*.db' | head -1)
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db $GRAFT_REPO_ROOT/gpurun_out/r01/${m}_kernel_stats.txt > /dev/null
done
cat $GRAFT_REPO_ROOT/gpurun_out/r01/*bench*.json
