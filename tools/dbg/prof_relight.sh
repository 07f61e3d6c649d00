export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prl -o rl -- python3 $R/tools/bench_relight.py --repeat 1 > /dev/null 2>&1
cp $(find /tmp/prl -name '*kernel_stats*' | head -1) $R/gpurun_out/relight_kernel_stats.csv
head -30 $R/gpurun_out/relight_kernel_stats.csv | cut -c1-200
