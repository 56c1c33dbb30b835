#!/bin/bash
# Developer aid (GPU box): the round's closing sequence -- smoke, GPU tests, bench, rocprofv3 summaries, timelines, PMC passes.
# Steps are joined so that a failed GPU step ends the sequence (no further GPU work behind a fault).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export ROUND=${ROUND:-r04}
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
grep -q "smoke ok" gpurun_out/smoke.log || exit 1
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/t_all.log 2>&1; rc=$?; tail -2 gpurun_out/t_all.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py > gpurun_out/bench_${ROUND}.json 2> gpurun_out/bench_${ROUND}.err; rc=$?; echo "bench rc=$rc"; [ $rc -eq 0 ] || exit 1
bash tools/dev/dev_rocprof.sh > gpurun_out/rocprof.log 2>&1 || { tail -3 gpurun_out/rocprof.log; exit 1; }
tail -1 gpurun_out/rocprof.log
bash tools/dev/dev_rocprof_train.sh > gpurun_out/rocprof_train.log 2>&1 || exit 1
timeout -k 10 200 python tools/dev/dev_wide2_timeline.py 256 > gpurun_out/${ROUND}_timeline_b256_eval.txt 2>&1 || exit 1
timeout -k 10 200 python tools/dev/dev_wide2_timeline.py 64 > gpurun_out/${ROUND}_timeline_b64_eval.txt 2>&1 || exit 1
timeout -k 10 200 python tools/dev/dev_wide2_timeline.py 1024 train > gpurun_out/${ROUND}_timeline_b1024_train.txt 2>&1 || exit 1
echo "timelines done"
( cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${ROUND}_pmc_bench/$c -o p -- python3 $R/bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > $R/gpurun_out/${ROUND}_pmc_bench_$c.log 2>&1 || { echo "pmc $c failed"; exit 1; }
done ) || exit 1
echo "bench pmc done (part 1 complete; parts 2 and 3: bash tools/dev/dev_pmc_fwd.sh, bash tools/dev/dev_pmc_train.sh -- gpurun allows 20 minutes per call)"
