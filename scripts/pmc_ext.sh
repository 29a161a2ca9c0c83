R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_ext
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_ext -- python3 $R/bench.py --steps 2 --warmup 1 > /dev/null 2>&1
cd $R; python - <<'PY'
import csv,glob,collections
p=glob.glob('gpurun_out/pmc_ext/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(p)):
    k=r['Kernel_Name'].split('(')[0]
    if 'extend16_kernel<' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    if r['Counter_Name']=='SQ_INSTS_VALU': n[k]+=1
for k in sorted(acc, key=lambda k:-acc[k]['GRBM_GUI_ACTIVE'])[:7]:
    v=acc[k]; L=n[k]
    print(k[:28], 'launches',L, 'VALU_inst/launch %.3e'%(v['SQ_INSTS_VALU']/L), 'valu_busy %.3f'%(4*v['SQ_ACTIVE_INST_VALU']/(1024*v['GRBM_GUI_ACTIVE']/8)), 'wait_any/wave_cyc %.3f'%(v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']), 'wait_inst %.3f'%(v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']), 'active_inst %.3f'%(v['SQ_ACTIVE_INST_ANY']/v['SQ_WAVE_CYCLES']), 'gui_ms %.3f'%(v['GRBM_GUI_ACTIVE']/L/8/2.0e6))
PY
rm -rf gpurun_out/pmc_ext
