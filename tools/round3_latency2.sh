# threshold sweep between the one-proof-per-wavefront and the 16-lane kernels
TAG=${1:-r3lat2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for wb in 8192 0; do
for n in 512 1024 1536 2048 3072 4096; do
  ZKV_WAVE_BELOW=$wb python bench.py --workload risc0_2p16 --proofs $n --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --no-wire --no-mulmod 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wave_below=$wb n=%d ms=%.3f proofs/s=%.0f stages=%s parity=%s' % ($n, j['ms_per_step'], j['value'], {k: round(v, 3) for k, v in j['stage_ms'].items()}, j['parity']['accept_reject_matches_construction']))"
done; done > $O/latency.txt 2>&1
cat $O/latency.txt
