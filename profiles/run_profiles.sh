#!/bin/bash
# Regenerates the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   bash profiles/run_profiles.sh <out-dir under gpurun_out/> [pf|so|rn|tn|pmc ...]
# kernel-trace/stats runs and PMC runs are separate rocprofv3 invocations (never combined), program directly after `--`.
set -o pipefail
# EXTRA="--batch 32" adds bench flags to every run; PMC_MATCH selects the kernel of the traffic reduction
# (default: the bf16x3 dominant kernel; 'igemm_conv_ws_kernel<3' with EXTRA="--mfma f32").
OUT=${1:-gpurun_out/prof}; shift
WHAT=${@:-pf so rn pmc}
EXTRA="${EXTRA:-} --no-other-mfma --no-host-input"
PMC_MATCH=${PMC_MATCH:-igemm_conv_x3_kernel<3,x3_tail_reduce_kernel}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for w in $WHAT; do
  case $w in
    pf) ARGS="--steps 5 --warmup 2 --no-cpu-baseline"; PER=2 ;;
    so) ARGS="--workload source_only --steps 5 --warmup 2 --no-cpu-baseline"; PER=1 ;;
    rn) ARGS="--backbone resnet --workload source_only --batch 8 --steps 5 --warmup 2 --no-cpu-baseline"; PER=1 ;;
    tn) ARGS="--use-tn --steps 5 --warmup 2 --no-cpu-baseline"; PER=5 ;;       # 2 halves x (2 grad-mode forwards) + the MC prefix forward
    pmc)
      rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/f.log 2>&1 || exit 1
      rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/w.log 2>&1 || exit 1
      # conv calls of the run = bench.py's roofline.launches (1 timed step) x 2 (the warm-up step ran the same launches)
      CALLS=$(grep '^{' $OUT/f.log | tail -1 | python3 -c "import sys, json; print(2 * json.loads(sys.stdin.read())['roofline']['launches'])")
      python3 profiles/pmc_traffic.py $(find $OUT/f -name '*counter_collection.csv') $(find $OUT/w -name '*counter_collection.csv') --match "$PMC_MATCH" --calls $CALLS > $OUT/traffic.json || exit 1
      rm -rf $OUT/f $OUT/w
      continue ;;
  esac
  MARK=stem_fwd; [ $w = rn ] && MARK=stem7_fwd_kernel
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -o $w -- python3 bench.py $ARGS $EXTRA > $OUT/${w}_bench.log 2>&1 || exit 1
  grep '^{' $OUT/${w}_bench.log | tail -1 > $OUT/${w}_bench_line.json
  python3 profiles/summarize_trace.py $(find $OUT/$w -name '*kernel_trace.csv') --marker $MARK --per-step $PER --warmup 2 --steps 5 --by-grid "${BY_GRID:-_x3_kernel,igemm_conv_kernel,igemm_conv_ws_kernel,igemm_wgrad}" > $OUT/${w}_summary.txt || exit 1
  cp $(find $OUT/$w -name '*kernel_stats.csv') $OUT/${w}_kernel_stats.csv
  rm -rf $OUT/$w
  echo "$w done"
done
