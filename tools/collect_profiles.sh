#!/bin/bash
# Collect the per-round profile artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r01'
#   gpurun --timeout 600 -- 'bash tools/collect_profiles.sh r03 pmc iir8 iirlp8'    (only the PMC passes of the workloads named, added to an existing collection)
# Writes gpurun_out/profiles_<round>/ ; copy what should be judged into profiles/<round>/.
set -u
R=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$R
PMC_ONLY=""
BENCH_ONLY=""
TABLES_ONLY=""
ONLY=""
if [ "${2:-}" = pmc ]; then shift 2; PMC_ONLY="$*"; mkdir -p $OUT;
elif [ "${2:-}" = only ]; then shift 2; ONLY="$*"; mkdir -p $OUT;   # kernel trace + PMC passes + bench line of the workloads named (a kernel changed after the collection)
elif [ "${2:-}" = tables ]; then TABLES_ONLY=1; mkdir -p $OUT;   # only the engine timing tables (they carry no source stamp: taken again with the final library)
elif [ "${2:-}" = bench ]; then BENCH_ONLY=1; mkdir -p $OUT;   # only the un-profiled bench lines (taken again once the reduced counters of this collection are in the tree, so that each quotes its traffic)
else rm -rf $OUT && mkdir -p $OUT; fi
cd /tmp && export TMPDIR=/tmp
RATE="upsample4 downsample3 firup12 firdn12 firup4 firdn4 rcup12 rcdn12 iirup2 iirdn3"
[ -n "$ONLY" ] && PMC_ONLY="$ONLY"
[ -z "$PMC_ONLY$BENCH_ONLY$TABLES_ONLY" -o -n "$ONLY" ] && for w in ${ONLY:-fir1024 updn43 iir8 fir127 iir8tp iir8cas iirlp8 iir8c64 fir1024c128 $RATE}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python $ROOT/bench.py --workload $w --no-cpu-baseline --no-other-configs --board-seconds 0 > $OUT/trace_bench_$w.json 2>/dev/null
  cp $OUT/trace_$w/*/*kernel_stats.csv $OUT/kernel_stats_$w.csv
  cp $OUT/trace_$w/*/*kernel_trace.csv $OUT/kernel_trace_$w.csv    # per-dispatch records: tools/reduce_pmc.py takes the steady-state duration from them (scratch: not committed)
  rm -rf $OUT/trace_$w
done
# PMC passes, each counter group in its own run (never combined with other trace domains)
[ -z "$BENCH_ONLY$TABLES_ONLY" ] && for w in ${PMC_ONLY:-fir1024 updn43 iir8 fir127 iir8tp iir8cas iirlp8 iir8c64 fir1024c128 $RATE}; do
  sets=("FETCH_SIZE" "WRITE_SIZE")
  [ $w = fir1024 ] && sets+=("SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES")
  # the issue / wait picture of config 4, before (cascade form) and after (parallel form)
  # (round 6: every row below half the roofline, not only the four of round 5)
  case $w in iir8|iir8cas|iir8tp|iir8c64|rcdn12|rcup12|iirup2|iirdn3|firdn12|firup4|firdn4|updn43|fir1024c128) sets+=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS");; esac
  for set in "${sets[@]}"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$tag -- python $ROOT/bench.py --workload $w --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs --board-seconds 0 > /dev/null 2>&1
    cp $OUT/pmc_$tag/*/*counter_collection.csv $OUT/pmc_${w}_$tag.csv 2>/dev/null
    rm -rf $OUT/pmc_$tag
  done
done
cd $ROOT
if [ -n "$ONLY" ]; then
  for w in $ONLY; do
    if [ $w = fir1024 ]; then python bench.py > $OUT/bench_fir1024.json 2>$OUT/bench_fir1024.err
    else python bench.py --workload $w --no-other-configs --no-cpu-baseline > $OUT/bench_$w.json 2>/dev/null; fi
  done
  ls -la $OUT | tail -5; exit 0
fi
[ -n "$PMC_ONLY" ] && { ls -la $OUT | tail -5; exit 0; }
if [ -z "$TABLES_ONLY" ]; then
python bench.py > $OUT/bench_fir1024.json 2>$OUT/bench_fir1024.err
for w in updn43 iir8 fir127 iir8tp iir8cas iirlp8 iir8c64 fir1024c128 $RATE; do
  python bench.py --workload $w --no-other-configs --no-cpu-baseline > $OUT/bench_$w.json 2>/dev/null
done
python bench.py --scaling strong --total-log2n 30 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_fir1024_2p30_one_gpu.json 2>/dev/null
[ -n "$BENCH_ONLY" ] && { ls -la $OUT | tail -3; exit 0; }
fi
python tools/power_probe.py idle copy fir1024 fir1024f32 fir1024f64 fir1024c128 updn43 fir127 iir8 iir8cas iir8c64 iirlp8 > $OUT/power_probe.txt 2>&1
python tools/ab_iir_par.py 26 > $OUT/ab_iir_par.txt 2>&1
python tools/time_fir_shapes.py > $OUT/fir_shapes.txt 2>&1
python tools/time_iir_up.py > $OUT/iir_up.txt 2>&1
python tools/time_iir_dn.py > $OUT/iir_dn.txt 2>&1
python tools/time_fir_c128.py > $OUT/fir_f64.txt 2>&1
# multirate_FIR.up: every engine (polyphase, walk over (tile, phase) pairs, the input-tile interpolators) and the default dispatch; .dn likewise
python tools/time_fir_up.py > $OUT/fir_up.txt 2>&1
python tools/time_fir_updn.py > $OUT/fir_updn.txt 2>&1
python tools/time_fir_filter.py > $OUT/fir_filter.txt 2>&1
python tools/acc_probe.py > $OUT/acc_probe.txt 2>&1
python tools/time_fir_dn.py > $OUT/fir_dn.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_strided_store.hip -o /tmp/ubss 2>/dev/null && /tmp/ubss > $OUT/ubench_strided_store.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_dp_pipes.hip -o /tmp/ubench_dp_pipes 2>/dev/null && /tmp/ubench_dp_pipes > $OUT/ubench_dp_pipes.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_f64_4x4.hip -o /tmp/ub44 2>/dev/null && /tmp/ub44 > $OUT/ubench_mfma_f64_4x4.txt 2>&1
ls -la $OUT
