cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc3
mkdir -p $O
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM GRBM_COUNT"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/$tag -- python $R/tools/prof_x3.py 0 8 > $O/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$O/*/')):
    f = glob.glob(d+'*/*_counter_collection.csv')
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'conv_fwd_x3' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    kt = glob.glob(d+'*/*_kernel_trace.csv')[0]
    durs=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(kt)) if 'conv_fwd_x3' in r['Kernel_Name']]
    print(d.split('/')[-2], 'launches', len(durs), 'avg_us %.1f' % (sum(durs[2:])/max(1,len(durs)-2)), {k: '%.5g' % (sum(v[2:])/max(1,len(v)-2)) for k,v in acc.items()})
PY
