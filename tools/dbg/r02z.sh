#!/bin/bash
# end-of-round measurement set (round 2): everything quoted in DESIGN.md / profiles/README.md
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02z}
O=$R/gpurun_out/$TAG
mkdir -p $O
bash $R/tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
bash $R/tools/dbg/pmc_chains.sh $TAG > $O/pmc_chains.log 2>&1
cd $R
python tools/bench_composite.py > $O/composite.json 2> /dev/null
python tools/bench_relight.py > $O/relight.json 2> /dev/null
python tools/bench_shadow.py > $O/shadow.json 2> /dev/null
python tools/run_e2e.py > $O/e2e.json 2> /dev/null
python bench.py --gpus 2 --steps 10 --warmup 3 --backend gloo --single-device --no-stage1 2> /dev/null | grep '^{' > $O/bench_2rank_gloo_single_device.json
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o c -- python3 $R/tools/bench_composite.py > /dev/null 2>&1
cp $(find /tmp/pc -name '*kernel_stats*' | head -1) $O/composite_kernel_stats.csv
ls -la $O
tail -c 600 $O/bench_stage2.json; echo; cat $O/bench_stage1.json | cut -c1-200; cat $O/composite.json | cut -c1-600; echo; cat $O/relight.json | cut -c1-700; echo; cat $O/shadow.json; cat $O/e2e.json | cut -c1-300
