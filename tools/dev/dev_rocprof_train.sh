R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for B in 256 1024; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr$B -o p -- python3 $R/tools/dev/dev_fwd_loop.py $B -1 train 60 > $R/gpurun_out/r04_train_b$B.out 2>&1
cp $(find /tmp/prof_tr$B -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${ROUND:-r04}_train_b${B}_kernel_stats.csv
done
