#!/bin/bash
# usage (GPU box, from the repo root): tools/profile_round.sh <tag> [commit]
# Takes the round's evidence for the CURRENT tree and leaves it under gpurun_out/profiles_<tag>/ :
#   <tag>_bench_no_overlap_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --no-overlap` (one stream: a row's
#                                             average is that kernel's own duration)
#   <tag>_bench_no_overlap_under_rocprof.json the bench line printed by that run
#   <tag>_step_table.txt                      tools/trace_step.py: per-kernel table of one train step
#   hbm_traffic_current.json                  tools/pmc_traffic.py over two PMC passes (FETCH_SIZE, WRITE_SIZE), stamped with
#                                             bench.kernel_source_hash() -> copy to profiles/ so bench.py quotes `traffic`
#   <tag>_predict_kernel_stats.csv            the same for `bench.py --mode predict`
#   <tag>_trace_overlap.txt                   per-queue busy time of the normal two-stream step (tools/trace_overlap.py)
#   <tag>_x3_vendor_yardstick.txt             tools/x3_vendor_yardstick.py
set -e
TAG=$1; COMMIT=${2:-unknown}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-secondary --sustained-seconds 0 --pool 512"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 2 --no-overlap $COMMON \
    > $OUT/${TAG}_bench_no_overlap_under_rocprof.json 2> gpurun_out/prof_$TAG.err
S=$(ls gpurun_out/prof_$TAG/*/*_kernel_stats.csv | head -1); cp $S $OUT/${TAG}_bench_no_overlap_kernel_stats.csv
T=$(ls gpurun_out/prof_$TAG/*/*_kernel_trace.csv | head -1); python3 tools/trace_step.py $T 4 > $OUT/${TAG}_step_table.txt
echo "kernel stats done" 
# PMC passes: counters only with --kernel-trace (separate runs, as the guide prescribes); 1 warm-up + 3 timed + 3 idle-probe steps
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -- python3 bench.py --steps 3 --warmup 1 --no-kernel-timers $COMMON \
      > gpurun_out/pmc_${TAG}_$C.json 2> gpurun_out/pmc_${TAG}_$C.err
  echo "pmc $C done"
done
F=$(ls gpurun_out/pmc_${TAG}_FETCH_SIZE/*/*_counter_collection.csv | head -1)
W=$(ls gpurun_out/pmc_${TAG}_WRITE_SIZE/*/*_counter_collection.csv | head -1)
python3 tools/pmc_traffic.py $F $W 7 $OUT/hbm_traffic_current.json $COMMIT > gpurun_out/pmc_${TAG}_fold.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profp_$TAG -- python3 bench.py --mode predict --steps 10 --warmup 2 --pool 512 \
    > $OUT/${TAG}_predict_under_rocprof.json 2> gpurun_out/profp_$TAG.err
S=$(ls gpurun_out/profp_$TAG/*/*_kernel_stats.csv | head -1); cp $S $OUT/${TAG}_predict_kernel_stats.csv
# the two-stream step as it really runs: per-queue busy time and what the overlap costs (tools/trace_overlap.py)
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/profo_$TAG -- python3 bench.py --steps 6 --warmup 2 --no-kernel-timers $COMMON \
    > gpurun_out/profo_$TAG.json 2> gpurun_out/profo_$TAG.err
T=$(ls gpurun_out/profo_$TAG/*/*_kernel_trace.csv | head -1); python3 tools/trace_overlap.py $T 2 5 > $OUT/${TAG}_trace_overlap.txt
# the vendor bf16 GEMM doing the MFMA work of one bf16x3 product, beside the planes kernel
python3 tools/x3_vendor_yardstick.py 3 > $OUT/${TAG}_x3_vendor_yardstick.txt 2>&1
echo "profiles in $OUT"; ls -la $OUT
