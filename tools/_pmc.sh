cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify > gpurun_out/st.json 2> gpurun_out/st.err
S=$(find gpurun_out/st -name "*kernel_stats.csv" | head -1); grep -i "dedupe\|scan64\|Cfg<512, 16, 9, 1, 4, 32, true>, 0, 0" $S | cut -c1-160
rm -rf gpurun_out/st
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-verify > gpurun_out/pmc_sq.json 2> gpurun_out/pmc_sq.err || { tail -5 gpurun_out/pmc_sq.err; exit 1; }
F=$(find gpurun_out/pmc_sq -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
best={}
for r in rows:
    k=r['Kernel_Name'][:40]
    if 'dedupe_kernel' in k:
        d=best.setdefault((k,r['Dispatch_Id']),{})
        d[r['Counter_Name']]=float(r['Counter_Value'])
for (k,i),v in best.items():
    print(k, i, {a: round(b/1e6,1) for a,b in v.items()})
PY
rm -rf gpurun_out/pmc_sq
