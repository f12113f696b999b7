cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_IFETCH" "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-include-regex "k_cond<|k_compact|k_cellsort|k_move|k_coal<" --output-format csv -d gpurun_out/pmc_r01_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers > gpurun_out/pmc_r01_$i.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_r01_*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].replace('void ','')[:22]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            print(k, {c: '%.4g'%(sum(x)/len(x)) for c,x in v.items()}, 'n=%d'%len(list(v.values())[0]))
PY
