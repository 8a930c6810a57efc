# Times the PLONK workload for every A/B library under csrc/build/ab (tools/ab_build.py) on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/ab_run_plonk.sh tag'
TAG=${1:-abp}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for lib in "" stylus_zkvm_verifiers_amd/csrc/build/ab/libzkv_*.so; do
  name=base; [ -n "$lib" ] && name=$(basename $lib .so)
  ZKV_LIB_PATH=${lib:+$R/$lib} timeout -k 10 200 python bench.py --workload plonk_2p18 --steps 3 --warmup 1 --no-cpu-baseline --no-wire --no-mulmod --no-extra-legs > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; exit 1; }
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('%-28s %.3f M/s  %s  parity %s' % (sys.argv[2], d['value'] / 1e6, {k: round(v, 2) for k, v in d['stage_ms'].items()}, d['parity']['accept_reject_matches_construction']))
PY
done
