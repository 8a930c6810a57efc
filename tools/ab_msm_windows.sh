R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/msm16; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_capi_host.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for bits in 8 16; do
  for w in sp1_2p20 risc0_2p16; do
    ZKV_MSM_WINDOW_BITS=$bits timeout -k 10 300 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-wire --no-mulmod --no-extra-legs > $O/${w}_$bits.json 2> $O/${w}_$bits.err || { tail -3 $O/${w}_$bits.err; exit 1; }
    python - $O/${w}_$bits.json $w $bits <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], '%.3f M/s' % (d['value'] / 1e6), {k: round(v, 3) for k, v in d['stage_ms'].items()}, d['parity']['accept_reject_matches_construction'])
PY
  done
done
