R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lone; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 32768 65536; do
  B="python3 $R/bench.py --workload risc0_2p16 --proofs $n --no-cpu-baseline --no-wire --no-mulmod --no-extra-legs --steps 2 --warmup 1"
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_$n -o p -- $B > $O/pmc_$n.log 2>&1 || { tail -5 $O/pmc_$n.log; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/pmc2_$n -o p -- $B > $O/pmc2_$n.log 2>&1 || { tail -5 $O/pmc2_$n.log; exit 1; }
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
for n in (32768, 65536):
    for sub in ('pmc_', 'pmc2_'):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for f in glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/lone/%s%d/**/*counter_collection.csv' % (sub, n)), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name'].split('(')[0]
                if 'miller2' in k or 'finalexp2' in k:
                    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, v in agg.items():
            print(n, sub, k.replace('zkv::',''), {c: '%.3g' % x for c, x in sorted(v.items())})
PY
