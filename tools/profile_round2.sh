# Round-2 measurement pass on the GPU box: gpu tests, the default bench line (sp1_2p20), rocprofv3 kernel trace and the PMC passes of
# the SAME command (each counter set in its own pass, --kernel-trace only), then the other workloads' bench lines.
#   gpurun --timeout 1190 -- 'bash tools/profile_round2.sh r2c'
set -e
TAG=${1:-r2c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || true
tail -2 $O/gpu_tests.log
timeout -k 10 500 python bench.py --steps 5 --warmup 1 > $O/bench_sp1_2p20.json 2> $O/bench.err
head -c 300 $O/bench_sp1_2p20.json; echo
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-wire --no-mulmod --no-extra-legs > $O/ktrace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o pf -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-wire --no-mulmod --no-extra-legs > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o pw -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-wire --no-mulmod --no-extra-legs > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o ps -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-wire --no-mulmod --no-extra-legs > $O/pmc_sq.log 2>&1
cd $R
timeout -k 10 300 python bench.py --workload risc0_2p16 --steps 5 --warmup 1 > $O/bench_risc0_2p16.json 2> $O/bench_risc0.err
timeout -k 10 300 python bench.py --workload mixed --steps 3 --warmup 1 > $O/bench_mixed_2p19.json 2> $O/bench_mixed.err
timeout -k 10 300 python bench.py --workload plonk_2p18 --steps 3 --warmup 1 > $O/bench_plonk_2p18.json 2> $O/bench_plonk.err
ls $O
