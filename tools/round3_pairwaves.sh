# A/B: wavefronts per workgroup of k_miller2 / k_finalexp2 (1 = the shipped library, 2 / 4 = variants built with -DZKV_PAIR_WAVES)
TAG=${1:-r3pw}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for W in 1 2 4 1 2 4; do
  if [ $W = 1 ]; then unset ZKV_LIB_PATH; else export ZKV_LIB_PATH=$R/stylus_zkvm_verifiers_amd/variants/libzkv_pairwaves$W.so; fi
  for wl in risc0_2p16 sp1_2p20; do
    python bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs --no-wire --no-mulmod 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('waves_per_group=$W $wl ms=%.3f proofs/s=%.0f stages=%s parity=%s' % (j['ms_per_step'], j['value'], {k: round(v, 3) for k, v in j['stage_ms'].items()}, j['parity']['accept_reject_matches_construction']))"
  done
done > $O/pairwaves.txt 2>&1
cat $O/pairwaves.txt
