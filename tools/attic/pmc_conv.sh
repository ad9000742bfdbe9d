cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc1
mkdir -p $O
for cfg in 6 20; do
  for shape in 0 11; do
   rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/a_${cfg}_${shape} -- python $R/tools/prof_conv.py $shape $cfg 6 > $O/a_${cfg}_${shape}.log 2>&1
   rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/b_${cfg}_${shape} -- python $R/tools/prof_conv.py $shape $cfg 6 > $O/b_${cfg}_${shape}.log 2>&1
  done
done
find $O -name "*.csv" | head -30
