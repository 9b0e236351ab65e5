cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_small
mkdir -p $O
for shp in 3x3 2x6; do
 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/mfma_$shp -o b --output-format csv -- python $R/tools/bench_small_sweeps.py --instances 4096 --steps 200 --repeats 1 --shapes $shp > $O/log_mfma_$shp.txt 2>&1
 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/lds_$shp -o b --output-format csv -- python $R/tools/bench_small_sweeps.py --instances 4096 --steps 200 --repeats 1 --shapes $shp > $O/log_lds_$shp.txt 2>&1
 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_SALU -d $O/inst_$shp -o b --output-format csv -- python $R/tools/bench_small_sweeps.py --instances 4096 --steps 200 --repeats 1 --shapes $shp > $O/log_inst_$shp.txt 2>&1
done
cd $O; python - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'combine_sweep' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
    print(f.split('/')[0], {k:(v/ max(1,n[k])) for k,v in agg.items()}, dict(n))
PY
