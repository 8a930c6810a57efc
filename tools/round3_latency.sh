# single-proof and small-batch latency (bench.py --proofs N on the RISC Zero workload), library as built
TAG=${1:-r3lat}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for n in 1 64 1024 4096 8192 10240 16384; do
  python bench.py --workload risc0_2p16 --proofs $n --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --no-wire --no-mulmod 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n=%d ms=%.3f proofs/s=%.0f stages=%s parity=%s' % ($n, j['ms_per_step'], j['value'], {k: round(v, 3) for k, v in j['stage_ms'].items()}, j['parity']['accept_reject_matches_construction']))"
done > $O/latency.txt 2>&1
cat $O/latency.txt
