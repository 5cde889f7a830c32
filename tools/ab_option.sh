# A/B of one option on bench workloads, one box: bash tools/ab_option.sh OPTION "v1 v2" "workloads" reps
OPT=$1; VALS=$2; WL=$3; REPS=${4:-3}
for w in $WL; do for i in $(seq $REPS); do line="$w"; for v in $VALS; do
  t=$(env SKDSP_$OPT=$v python bench.py --workload $w --no-cpu-baseline --no-other-configs --board-seconds 0 --steps 300 | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  line="$line $OPT=$v $t"; done; echo "$line"; done; done
