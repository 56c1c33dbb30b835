#!/bin/bash
# Developer aid (GPU box): rocprofv3 kernel-trace summaries of the round's three reference runs, copied to gpurun_out/${ROUND:-r04}_*.
# Usage (from the repo root, through gpurun):  bash tools/dev/dev_rocprof.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() {   # name, program args...
  local name=$1; shift
  echo "== $name: $*"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python3 "$@" > $R/gpurun_out/${ROUND:-r04}_${name}.out 2> $R/gpurun_out/${ROUND:-r04}_${name}.err || { echo "rocprofv3 failed for $name"; tail -5 $R/gpurun_out/${ROUND:-r04}_${name}.err; return 1; }
  local f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $R/gpurun_out/${ROUND:-r04}_${name}_kernel_stats.csv && head -12 "$f"
  local t=$(find /tmp/prof_$name -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && cp "$t" $R/gpurun_out/${ROUND:-r04}_${name}_kernel_trace.csv
  echo "done $name"
}
run bench_b16 $R/bench.py --steps 400 --warmup 20 --no-extras --no-cpu-baseline &&
run train_b64 $R/tools/dev/dev_fwd_loop.py 64 -1 train 300 &&
run fwd_b256 $R/tools/dev/dev_fwd_loop.py 256 -1 eval 300
