# SQ counter passes (each group its own rocprofv3 pass) around an arbitrary command:
#   gpurun --timeout 900 -- 'bash tools/pmc_sq_cmd.sh dn3 python tools/run_fir_call.py dn 1024 3 c64 100 fir_dn4k=2'
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc2; mkdir -p $OUT; TAG=$1; shift; CMD="$@"; cd /tmp; export TMPDIR=/tmp; i=0
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  (cd $ROOT && timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/out_${TAG}_$i.txt 2>$OUT/err_${TAG}_$i.txt)
  cp $OUT/p$i/*/*counter_collection.csv $OUT/${TAG}_set$i.csv 2>/dev/null; rm -rf $OUT/p$i
done
python - $OUT $TAG <<'P'
import sys,csv,glob,collections
out,w=sys.argv[1:3]
res={}
for f in sorted(glob.glob(out+'/'+w+'_set*.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'skdsp::' in r['Kernel_Name'] and 'fill' not in r['Kernel_Name'] and 'noise' not in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): res[k]=sum(v)/len(v)
wv=res.get('SQ_WAVES',1)
print(w, 'waves %d' % wv, ' '.join('%s/wave %.0f' % (k[3:], v/wv) for k,v in res.items() if k!='SQ_WAVES'), 'busy_us %.1f' % (res.get('SQ_BUSY_CYCLES',0)/32/2400), flush=True)
P
