# Aggregate check at 2^20 SP1 proofs for 1, 2, 4, 8 proofs per Miller accumulator: tools/agg_group_sweep.sh  (on the GPU box)
for g in 1 2 4 8; do
  echo "== ZKV_AGG_GROUP=$g"
  ZKV_AGG_GROUP=$g python tools/bench_aggregate.py --log2 20 --sub 64,16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except ValueError: continue
    if d['aggregate']: print(d['mutate_every'], d['aggregate'], d['ms'], d['proofs_per_s'], d['stage_ms'], d['parity'], d['sub_batches'])
"
done
