#!/bin/bash
# Regenerates the whole evidence set of a round on the GPU box:  bash profiles/refresh_all.sh <out-dir under gpurun_out/>
# (kernel-trace runs and PMC passes are separate rocprofv3 invocations; see run_profiles.sh)
set -o pipefail
OUT=${1:-gpurun_out/evidence}
mkdir -p $OUT
bash profiles/run_profiles.sh $OUT/b16 pf pmc > $OUT/b16.log 2>&1 || { echo "b16 failed"; exit 1; }
echo "b16 done"
EXTRA="--batch 32" bash profiles/run_profiles.sh $OUT/b32 pf pmc > $OUT/b32.log 2>&1 || { echo "b32 failed"; exit 1; }
echo "b32 done"
EXTRA="--mfma f32" PMC_MATCH='igemm_conv_ws_kernel<3' bash profiles/run_profiles.sh $OUT/f32 pmc > $OUT/f32.log 2>&1 || { echo "f32 pmc failed"; exit 1; }
echo "f32 pmc done"
bash profiles/run_profiles.sh $OUT/other so rn tn > $OUT/other.log 2>&1 || { echo "so/rn/tn failed"; exit 1; }
echo "so rn tn done"
# ResNet-101 (BASELINE.json configs[4] shape on one GPU): PMC traffic of its dominant kernel (the bf16x3 multi-tap conv)
EXTRA="--backbone resnet --workload source_only --batch 8" bash profiles/run_profiles.sh $OUT/rn pmc > $OUT/rn_pmc.log 2>&1 || { echo "resnet pmc failed"; exit 1; }
echo "resnet pmc done"
python3 bench.py --backbone resnet --workload source_only --batch 8 --steps 10 > $OUT/bench_rn_b8.log 2>&1 || { echo "bench rn failed"; exit 1; }
grep '^{' $OUT/bench_rn_b8.log | tail -1 > $OUT/bench_rn_b8.json
python3 bench.py > $OUT/bench_b16.log 2>&1 || { echo "bench b16 failed"; exit 1; }
grep '^{' $OUT/bench_b16.log | tail -1 > $OUT/bench_b16.json
python3 bench.py --batch 32 --steps 5 > $OUT/bench_b32.log 2>&1 || { echo "bench b32 failed"; exit 1; }
grep '^{' $OUT/bench_b32.log | tail -1 > $OUT/bench_b32.json
echo "bench lines done"
python3 tests/bench_x3.py > $OUT/bench_x3.txt 2>&1 || { echo "bench_x3 failed"; exit 1; }
echo "all done"
