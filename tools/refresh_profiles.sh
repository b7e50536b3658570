# Re-create the round's measurement set on the GPU box (run from the repo root under gpurun); results land in gpurun_out/r06/
# and are copied to profiles/r06_* after review:  bench lines, rocprofv3 kernel-trace summaries of the SAME commands, the PMC
# collection (tools/pmc_collect.py), the per-launch breakdown of both stacks, the projection microbenchmark, the attention
# A/B table of the knobs that decide the variants, the sublayer modes.
set -x
R=gpurun_out/r06; mkdir -p $R
timeout 900 python bench.py --steps 20 --warmup 5 > $R/bench_default.json 2> $R/bench_default.err
timeout 400 python bench.py --workload ip --steps 20 --warmup 5 > $R/bench_ip.json 2> $R/bench_ip.err
timeout 400 python bench.py --workload seq16 --steps 20 --warmup 5 --no-cpu-baseline > $R/bench_seq16.json 2> $R/bench_seq16.err
cd /tmp && export TMPDIR=/tmp
for w in sdxl sd15; do
  rm -rf /tmp/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-also --min-seconds 0 $([ $w = sd15 ] && echo --passes serial) > $GRAFT_REPO_ROOT/$R/${w}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$R/${w}_rocprof.err
  db=$(ls -S $(find /tmp/prof_$w -name '*.db') | head -1)      # the bench's own database, not the one of the mfma_ceiling subprocess it starts
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db $GRAFT_REPO_ROOT/$R/${w}_kernel_stats.txt > /dev/null
done
cd $GRAFT_REPO_ROOT
for w in sdxl sd15; do python tools/stack_breakdown.py $w > $R/breakdown_$w.txt 2>/dev/null; done
python tools/kbench_proj.py > $R/kbench_proj.txt 2>/dev/null
python tools/kbench_attn_ab.py "ATTN_V2=-1" "ATTN_V2=0" --rounds 5 --iters 6 --shapes sdxl --inner > $R/kbench_attn.txt 2>/dev/null
python tools/kbench_attn_ab.py "ATTN_TX=0" "ATTN_TX=-1" --rounds 5 --iters 6 --shapes sdxl --only x77 > $R/kbench_attn_tx.txt 2>/dev/null
python tools/gemm_ab.py "GEMM_TRI=0,GEMM_PP=0" "GEMM_TRI=0,GEMM_PP=1" "GEMM_TRI=-1,GEMM_PP=1" --rounds 5 --iters 10 --torch > $R/gemm_ab.txt 2>/dev/null
python tools/gemm_ab.py "GEMM_RS=0" "GEMM_RS=-1" "GEMM_RS=1" --short --rounds 5 --iters 10 > $R/gemm_rs_ab.txt 2>/dev/null
for u in mfma_cadence store_bw valu_rate barrier_cost wave_simd head_stride_copy mfma_ceiling gemm_w4_model; do
  [ -x tools/ubench/$u ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/$u tools/ubench/$u.hip
  ./tools/ubench/$u > $R/ubench_$u.txt 2>&1
done
( for wl in sdxl sd15; do for mode in "off 1" "fused 0" "fused 1"; do set -- $mode
    AID_LN_FOLD=$2 python bench.py --workload $wl --sublayers $1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-also 2>/dev/null | tail -1 |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], 'sublayers=' + sys.argv[2], 'AID_LN_FOLD=' + sys.argv[3], 'frames/s', round(d['value'], 3), 'ms/step', round(d['ms_per_step'], 3))" $wl $1 $2
  done; done ) > $R/sublayers.txt
cp profiles/r05_depth_parity.json gpurun_out/depth_parity.json 2>/dev/null    # (keeps the *_fullwidth entries of the AID_E2E_FULL_WIDTH=1 run)
AID_WRITE_MEASUREMENTS=1 python -m pytest tests/test_hip_depth_and_pipelines.py -m gpu -q > $R/depth_pytest.log 2>&1; cp gpurun_out/depth_parity.json $R/ 2>/dev/null
timeout 1500 python tools/pmc_collect.py $R/pmc.json > $R/pmc.log 2>&1
tail -5 $R/pmc.log
