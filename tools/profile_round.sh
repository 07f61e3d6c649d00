#!/bin/bash
# Collects the measurement evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01h
# writes gpurun_out/<tag>/{bench_stage2.json, bench_stage1.json, *_kernel_stats.csv, pmc_*.csv};
# tools/collect_profiles.py then copies / reduces them into profiles/ (tracked).
# rocprofv3 rules of this pool: the program itself follows `--` (no env / bash hops), counters are collected in
# their own passes with --kernel-trace only (never with sys / hip / hsa traces).
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py > $O/bench_stage2.json 2> $O/bench_stage2.err
python3 $R/tools/bench_stage1.py --steps 20 --warmup 5 > $O/bench_stage1.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps2 -o s2 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stage1 --no-extra > /dev/null 2>&1
cp $(find /tmp/ps2 -name '*kernel_stats*' | head -1) $O/bench_stage2_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps1 -o s1 -- python3 $R/tools/bench_stage1.py --steps 3 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/ps1 -name '*kernel_stats*' | head -1) $O/bench_stage1_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    N=$(echo $C | cut -d' ' -f1)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$N -o c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage1 --no-extra > /dev/null 2>&1
    F=$(find /tmp/pmc_$N -name '*counter_collection*' | head -1)
    # keep the header and the fused-MLP dispatches only (the full table is large)
    (head -1 $F; grep mlp_infer_kernel $F) > $O/pmc_$N.csv
done
# round 3 additions: per-experiment evidence (each a JSON / text file of its own)
python3 $R/tools/dbg/bench_march.py 2>/dev/null | tail -1 > $O/march_sweep.json
python3 $R/tools/dbg/bench_x3.py 2>/dev/null | tail -1 > $O/bf16x6_kernel.json
ZERO=1 python3 $R/tools/dbg/bench_x3.py 2>/dev/null | tail -1 > $O/bf16x6_kernel_zero_operands.json
python3 $R/tools/dbg/ab_tn256.py $R/psnerf_amd/libpsnerf_hip.so 2>/dev/null | grep -v amdgpu > $O/tn256.txt
python3 $R/tools/bench_shadow.py 2>/dev/null | tail -1 > $O/shadow_visibility.json
python3 $R/tools/bench_composite.py 2>/dev/null | tail -1 > $O/composite.json
# matrix-pipe occupancy of the split-bf16 engine (own PMC pass, --kernel-trace only)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_x3 -o c -- python3 $R/tools/dbg/bench_x3.py > /dev/null 2>&1
F=$(find /tmp/pmc_x3 -name '*counter_collection*' | head -1)
(head -1 $F; grep "mlp_infer_x3_kernel\|mlp_infer_bf16_kernel" $F | head -40) > $O/pmc_x3.csv
# round 4 additions: the rank shards of BASELINE cfg 4 on one GPU (data-parallel path on, eager / HIP graph, host vs GPU time),
# the split-bf16 occupancy engine, the kernel timeline of a replayed 4096-px step
python3 $R/tools/strong_projection.py --graph --queue-ahead --steps 40 --out $O/strong_projection.json > $O/strong_projection.log 2>&1
python3 $R/tools/dbg/x3occ_probe.py 2>/dev/null | grep -v amdgpu > $O/x3occ.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps4 -o s4 -- python3 $R/tools/strong_projection.py --pixels 4096 --steps 30 --graph --no-profiler --out /tmp/ps4/p.json > /dev/null 2>&1
cp $(find /tmp/ps4 -name '*kernel_stats*' | head -1) $O/strong4096_kernel_stats.csv
# kernel timelines (start offset, duration, gap, queue per launch of one step): the replayed 4096-px rank shard and the eager 32768-px step
for CASE in "4096 graph" "32768"; do
    N=$(echo $CASE | tr ' ' '_')
    rm -rf /tmp/tl_$N
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$N -o t -- python3 $R/tools/dbg/trace_step.py $CASE > /dev/null 2>&1
    python3 $R/tools/dbg/trace_step_analyse.py $(find /tmp/tl_$N -name '*kernel_trace.csv' | head -1) > $O/timeline_${N}.txt 2>&1
done
# stage 1: the rank shards of the ray-parallel step (4096 training rays over 1 / 2 / 4 / 8 ranks) on this one GPU
for RAYS in 4096 2048 1024 512; do
    python3 $R/tools/bench_stage1.py --rays $RAYS --steps 30 --warmup 5 2>/dev/null | tail -1
done > $O/stage1_rank_shards.jsonl
# round 5 additions: configs[4] at its stated size with the device sampler in the loop, the split-bf16 weight-gradient kernel, the
# single-dump chains A/B
python3 $R/tools/run_e2e.py --full --json-out $O/e2e_full.json > $O/e2e_full.log 2>&1
python3 $R/tools/run_e2e.py --full --precision bf16x3 --json-out $O/e2e_full_bf16x3.json > $O/e2e_full_bf16x3.log 2>&1
python3 $R/tools/dbg/bench_tn256_x3.py 2>/dev/null | tail -1 > $O/tn256_x3.json
python3 $R/tools/dbg/ab_single_dump.py 2>/dev/null | tail -1 > $O/ab_single_dump.json
# split-bf16 weight stages (PSN_W_BF16X2): the four geometry chains and the shading-row launch, fp32 vs three partial products
python3 $R/tools/dbg/ab_chain_x3.py 2>/dev/null | tail -1 > $O/ab_chain_x3.json
python3 $R/tools/dbg/bench_lrow_x3.py 2>/dev/null | tail -1 > $O/lrow_x3.json
# round 6 additions: the visibility launch per workgroup order (times + PMC traffic), the composite kernels under rocprofv3 at a
# size where they are HBM-bound (kernel stats + FETCH_SIZE / WRITE_SIZE), the geometry chains' counters
python3 $R/tools/ab_block_order.py 2>/dev/null | tail -1 > $O/ab_block_order.json
bash $R/tools/pmc_block_order.sh $TAG > /dev/null 2>&1
bash $R/tools/prof_composite.sh $TAG > /dev/null 2>&1
bash $R/tools/dbg/pmc_chains.sh $TAG single > /dev/null 2>&1
ls -la $O
