#!/bin/bash
# Developer aid (build container): copies the closing sequence's records from gpurun_out/ (scratch) into profiles/ under the round's name
# and regenerates the derived tables.   ROUND=r04 bash tools/dev/collect_profiles.sh
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
ROUND=${ROUND:-r04}
G=gpurun_out
cp $G/bench_${ROUND}.json profiles/${ROUND}_bench_bf16.json
grep -h '^{"metric"' $G/${ROUND}_bench_b16.out | tail -1 > profiles/${ROUND}_bench_bf16_under_rocprof.json
for n in bench_b16 train_b64 fwd_b256 train_b256 train_b1024; do cp $G/${ROUND}_${n}_kernel_stats.csv profiles/${ROUND}_${n}_kernel_stats.csv; done
python profiles/dispatch_table.py profiles/${ROUND}_dispatch_table.csv B16=$G/${ROUND}_bench_b16_kernel_trace.csv B64=$G/${ROUND}_train_b64_kernel_trace.csv > /dev/null
for t in b256_eval b64_eval b1024_train; do cp $G/${ROUND}_timeline_${t}.txt profiles/${ROUND}_timeline_${t}.txt; done
python profiles/summarize_pmc.py $G/${ROUND}_pmc_bench/FETCH_SIZE $G/${ROUND}_pmc_bench/WRITE_SIZE > /dev/null
mkdir -p profiles/${ROUND}_pmc_bench
for c in FETCH_SIZE WRITE_SIZE; do f=$(find $G/${ROUND}_pmc_bench/$c -name '*counter_collection.csv' | head -1); python profiles/compact_pmc.py $f > /dev/null; cp ${f%counter_collection.csv}per_kernel.csv profiles/${ROUND}_pmc_bench/${c}_per_kernel.csv; done
for name in fwd_b256 train_b1024; do
  for d in $G/${ROUND}_pmc/$name/*/; do c=$(basename $d); f=$(find $d -name '*per_kernel.csv' | head -1); [ -n "$f" ] && mkdir -p profiles/${ROUND}_pmc/$name && cp $f profiles/${ROUND}_pmc/$name/${c}_per_kernel.csv; done
done
ls profiles | grep ${ROUND}
