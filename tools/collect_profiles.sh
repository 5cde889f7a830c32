#!/bin/bash
# Collect the per-round profile artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r01'
# Writes gpurun_out/profiles_<round>/ ; copy what should be judged into profiles/<round>/.
set -u
R=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$R
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in fir1024 updn43 iir8 fir127; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python $ROOT/bench.py --workload $w --no-cpu-baseline > /dev/null 2>&1
  cp $OUT/trace_$w/*/*kernel_stats.csv $OUT/kernel_stats_$w.csv
  rm -rf $OUT/trace_$w
done
# PMC passes for the headline kernel, each counter group in its own run
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$tag -- python $ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1
  cp $OUT/pmc_$tag/*/*counter_collection.csv $OUT/pmc_fir1024_$tag.csv 2>/dev/null
  cp $OUT/pmc_$tag/*/*kernel_trace.csv $OUT/pmc_fir1024_${tag}_trace.csv 2>/dev/null
  rm -rf $OUT/pmc_$tag
done
cd $ROOT
python bench.py > $OUT/bench_fir1024.json 2>$OUT/bench_fir1024.err
for w in updn43 iir8 fir127; do
  python bench.py --workload $w > $OUT/bench_$w.json 2>/dev/null
done
ls -la $OUT
