#!/bin/bash
# Developer aid (GPU box): the round's closing sequence -- smoke, GPU tests, bench, rocprofv3 summaries, PMC traffic passes, timelines.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/t_all.log 2>&1; tail -2 gpurun_out/t_all.log
timeout -k 10 600 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; echo "bench rc=$?"
bash tools/dev/dev_rocprof.sh > gpurun_out/rocprof.log 2>&1; tail -1 gpurun_out/rocprof.log
python tools/dev/dev_wide_timeline.py 256 -1 > gpurun_out/${ROUND:-r04}_timeline_b256_eval.txt 2>&1
python tools/dev/dev_wide_timeline.py 64 -1 > gpurun_out/${ROUND:-r04}_timeline_b64_eval.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${ROUND:-r04}_pmc_bench/$c -o p -- python3 $R/bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > $R/gpurun_out/${ROUND:-r04}_pmc_bench_$c.log 2>&1 || echo "pmc $c failed"
done
echo "final run done"
