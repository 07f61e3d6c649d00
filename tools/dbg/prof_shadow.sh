export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/bench_shadow.py > $R/gpurun_out/shadow_r03a.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psh -o sh -- python3 $R/tools/bench_shadow.py > /dev/null 2>&1
cp $(find /tmp/psh -name '*kernel_stats*' | head -1) $R/gpurun_out/shadow_r03a_kernel_stats.csv
head -25 $R/gpurun_out/shadow_r03a_kernel_stats.csv | cut -c1-200
cat $R/gpurun_out/shadow_r03a.json
