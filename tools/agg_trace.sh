# Per-dispatch kernel durations of one aggregate-mode batch (rocprofv3 kernel trace): tools/agg_trace.sh <sub> <mutate_every> <tag>
SUB=${1:-64}; MUT=${2:-64}; TAG=${3:-aggtrace}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export ZKV_SYNTH_WORKERS=1
rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 $R/tools/bench_aggregate.py --log2 20 --mutate $MUT --sub $SUB --steps 1 > $O/run.log 2>&1
rc=$?
python3 - <<PY
import csv, glob
fs = glob.glob('$O/raw/**/*kernel_trace.csv', recursive=True)
rows = []
for f in fs:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = open('$O/dispatches.txt', 'w')
for r in rows:
    name = r['Kernel_Name'].split('(')[0][-40:]
    out.write('%-42s %9.3f ms grid=%s\n' % (name, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r.get('Grid_Size', r.get('Grid_Size_X', '?'))))
out.close()
PY
tail -60 $O/dispatches.txt
exit $rc
