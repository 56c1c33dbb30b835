#!/bin/bash
# Developer aid (GPU box): one rocprofv3 --pmc pass per counter over a short run; CSVs land in gpurun_out/${ROUND:-r04}_pmc/<name>/<COUNTER>/.
# Usage: bash tools/dev/dev_pmc.sh <name> "<COUNTER> <COUNTER> ..." <python program and args>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; counters=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for c in $counters; do
  out=$R/gpurun_out/${ROUND:-r04}_pmc/$name/$c
  mkdir -p $out
  echo "== $name $c"
  timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d $out -o p -- python3 "$@" > $out/run.log 2>&1 || { echo "failed: $c"; tail -3 $out/run.log; }
  find $out -name '*counter_collection.csv' | head -1
done
