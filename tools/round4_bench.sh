# Round-4 bench pass: the default bench line and the other workloads.
#   gpurun --timeout 1190 -- 'bash tools/round4_bench.sh r4a'
TAG=${1:-r4a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 500 python bench.py --steps 5 --warmup 1 > $O/bench_sp1_2p20.json 2> $O/bench_sp1.err && head -c 300 $O/bench_sp1_2p20.json && echo &&
timeout -k 10 300 python bench.py --workload risc0_2p16 --steps 5 --warmup 1 > $O/bench_risc0_2p16.json 2> $O/bench_risc0.err && head -c 300 $O/bench_risc0_2p16.json && echo &&
timeout -k 10 300 python bench.py --workload mixed --steps 3 --warmup 1 > $O/bench_mixed_2p19.json 2> $O/bench_mixed.err && head -c 300 $O/bench_mixed_2p19.json && echo &&
timeout -k 10 300 python bench.py --workload plonk_2p18 --steps 3 --warmup 1 > $O/bench_plonk_2p18.json 2> $O/bench_plonk.err && head -c 300 $O/bench_plonk_2p18.json && echo
rc=$?
tail -3 $O/*.err
exit $rc
