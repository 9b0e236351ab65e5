R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for attempt in 1 2 3; do
  rm -rf $O/pmc_valu
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_valu -o v -- python $R/tools/bench_cfg5_variants.py flip with_elements one_workgroup > $O/pmc_valu.log 2>&1
  if compgen -G "$O/pmc_valu/*.db" > /dev/null || compgen -G "$O/pmc_valu/*/*.db" > /dev/null; then break; fi
done
python $R/tools/summarize_rocprof.py $O $O/pmc_flip.md "cfg 5 shard kernels: instruction mix (tools/bench_cfg5_variants.py flip with_elements one_workgroup)" > /dev/null 2>&1
find $O -name "*.db" -delete
tail -3 $O/pmc_valu.log; grep -n "flip_duo\|sweep_duo\|ell_sweep_kernel" $O/pmc_flip.md | head -40
