# Developer aid (GPU box): PMC counter passes (one per counter) over 20 training steps at B = 1024 -- the figures DESIGN.md quotes for the
# 64-row training kernels (rgfwd2 saving variant, bwd1w).   bash tools/dev/dev_pmc_train.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export ROUND=r04
bash tools/dev/dev_pmc.sh train_b1024 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM TCP_PENDING_STALL_CYCLES TA_BUSY_avr FETCH_SIZE WRITE_SIZE" $R/tools/dev/dev_fwd_loop.py 1024 -1 train 20 > gpurun_out/r04_pmc_train.log 2>&1
for f in $(find gpurun_out/r04_pmc/train_b1024 -name '*counter_collection.csv'); do python profiles/compact_pmc.py $f > /dev/null; rm -f $f; done
find gpurun_out/r04_pmc/train_b1024 -name '*.csv' | head -20
