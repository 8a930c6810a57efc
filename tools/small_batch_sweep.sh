# Small-batch latency sweep: the 16-lane kernels (automatic below ZKV_WIDE_BELOW) against the lane-pair kernels (ZKV_WIDE_BELOW=0).
#   gpurun -- 'bash tools/small_batch_sweep.sh > gpurun_out/small_batch.txt'
for n in 1 1024 4096 8192 10240 12288 16384; do
  for wb in default 0; do
    if [ $wb = 0 ]; then export ZKV_WIDE_BELOW=0; else unset ZKV_WIDE_BELOW; fi
    python bench.py --workload risc0_2p16 --proofs $n --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --no-wire --no-mulmod 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n=%d wide_below=%s ms=%.3f proofs/s=%.0f stages=%s' % ($n, '$wb', j['ms_per_step'], j['value'], {k: round(v, 3) for k, v in j['stage_ms'].items()}))"
  done
done
