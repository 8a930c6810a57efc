set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r1k; mkdir -p $O
timeout -k 10 300 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
timeout -k 10 400 python bench.py --steps 5 --warmup 1 > $O/bench_risc0_2p16.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/ktrace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o pf -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o pw -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o ps -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_sq.log 2>&1
cd $R
timeout -k 10 200 python bench.py --workload sp1_2p20 --proofs 262144 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_sp1_2p18.json 2> $O/bench_sp1.err
timeout -k 10 300 python bench.py --workload mixed --proofs 131072 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_mixed_2p17.json 2> $O/bench_mixed.err
tail -c 1500 $O/bench_risc0_2p16.json; tail -3 $O/gpu_tests.log
