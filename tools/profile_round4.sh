# Round-4 measurement pass for ONE bench workload: the bench line, the rocprofv3 kernel trace and the PMC passes of the SAME command
# (each counter set in its own pass, --kernel-trace only; the program itself after `--`).
#   gpurun --timeout 1190 -- 'bash tools/profile_round4.sh r4p sp1_2p20 && bash tools/profile_round4.sh r4p risc0_2p16'
#   extra bench flags as the third argument, e.g. 'bash tools/profile_round4.sh r4p risc0_2p16 "--proofs 4096" risc0_4096'
TAG=${1:-r4p}; W=${2:-sp1_2p20}; EXTRA=${3:-}; NAME=${4:-$W}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_$NAME; mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --workload $W $EXTRA --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
head -c 250 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload $W $EXTRA --no-cpu-baseline --no-wire --no-mulmod --no-extra-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o kt -- $B --steps 3 --warmup 1 > $O/ktrace.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o pf -- $B --steps 1 --warmup 0 > $O/pmc_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o pw -- $B --steps 1 --warmup 0 > $O/pmc_write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o ps -- $B --steps 1 --warmup 0 > $O/pmc_sq.log 2>&1
rc=$?
cd $R
find $O -name "*.csv" | head -20
exit $rc
