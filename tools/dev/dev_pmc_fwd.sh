# Developer aid (GPU box): PMC counter passes (one per counter) + a kernel trace of 200 eval forwards at B = 256 -- the figures DESIGN.md quotes
# for the 64-row forward kernels.  Raw per-dispatch dumps are reduced to one row per kernel (profiles/compact_pmc.py).   bash tools/dev/dev_pmc_fwd.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export ROUND=r04
bash tools/dev/dev_pmc.sh fwd_b256 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS TA_BUSY_avr FETCH_SIZE WRITE_SIZE" $R/tools/dev/dev_fwd_loop.py 256 -1 eval 60 > gpurun_out/r04_pmc_fwd.log 2>&1
for f in $(find gpurun_out/r04_pmc/fwd_b256 -name '*counter_collection.csv'); do python profiles/compact_pmc.py $f > /dev/null; rm -f $f; done
find gpurun_out/r04_pmc -name '*.csv' | head -20
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fwd256 -o p -- python3 $R/tools/dev/dev_fwd_loop.py 256 -1 eval 200 > $R/gpurun_out/r04_fwd_b256.out 2>&1; cp $(find /tmp/prof_fwd256 -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r04_a_fwd_b256_kernel_stats.csv; head -8 $R/gpurun_out/r04_a_fwd_b256_kernel_stats.csv | cut -c1-200
