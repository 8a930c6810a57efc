# Round-3 bench pass: default bench line, the multi-rank rehearsals on one GPU (self-launching bench.py, collectives over gloo), other workloads.
#   gpurun --timeout 1190 -- 'bash tools/round3_bench.sh r3a'
TAG=${1:-r3a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 500 python bench.py --steps 5 --warmup 1 > $O/bench_sp1_2p20.json 2> $O/bench_sp1.err && head -c 400 $O/bench_sp1_2p20.json && echo &&
timeout -k 10 400 python bench.py --gpus 2 --rehearse-single-gpu --steps 2 --warmup 1 --no-mulmod > $O/bench_mixed_rehearse_n2.json 2> $O/bench_n2.err && head -c 300 $O/bench_mixed_rehearse_n2.json && echo &&
timeout -k 10 400 python bench.py --gpus 4 --rehearse-single-gpu --steps 2 --warmup 1 --no-mulmod --proofs 262144 > $O/bench_mixed_rehearse_n4.json 2> $O/bench_n4.err && head -c 300 $O/bench_mixed_rehearse_n4.json && echo &&
timeout -k 10 300 python bench.py --workload mixed --steps 3 --warmup 1 > $O/bench_mixed_2p19.json 2> $O/bench_mixed.err &&
timeout -k 10 300 python bench.py --workload risc0_2p16 --steps 5 --warmup 1 > $O/bench_risc0_2p16.json 2> $O/bench_risc0.err &&
timeout -k 10 300 python bench.py --workload plonk_2p18 --steps 3 --warmup 1 > $O/bench_plonk_2p18.json 2> $O/bench_plonk.err
rc=$?
tail -3 $O/*.err
ls $O
exit $rc
