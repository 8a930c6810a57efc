# threshold sweep for the two-wavefront Miller kernel
TAG=${1:-r3lat3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for db in 4096 0; do
for n in 1 128 256 384 512 768 1024; do
  ZKV_DUAL_BELOW=$db python bench.py --workload risc0_2p16 --proofs $n --steps 30 --warmup 3 --no-cpu-baseline --no-extra-legs --no-wire --no-mulmod 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dual_below=$db n=%d ms=%.3f proofs/s=%.0f stages=%s parity=%s' % ($n, j['ms_per_step'], j['value'], {k: round(v, 3) for k, v in j['stage_ms'].items()}, j['parity']['accept_reject_matches_construction']))"
done; done > $O/latency.txt 2>&1
cat $O/latency.txt
