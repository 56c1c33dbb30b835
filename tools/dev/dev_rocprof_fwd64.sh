R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for B in 64; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f$B -o p -- python3 $R/tools/dev/dev_fwd_loop.py $B -1 eval 300 > $R/gpurun_out/r04_fwd_b$B.out 2>&1
cp $(find /tmp/prof_f$B -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r04_k_fwd_b${B}_kernel_stats.csv
done
