# Kernel trace + PMC passes (each counter set in its own pass, --kernel-trace only) of one aggregate-mode pass over 2^20 SP1 proofs:
#   gpurun --timeout 1190 -- 'bash tools/profile_aggregate.sh r3agg 0 64'        (mutate_every, sub-batch)
TAG=${1:-r3agg}; MUT=${2:-0}; SUB=${3:-64}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_m${MUT}_s${SUB}; mkdir -p $O
cd $R
python tools/bench_aggregate.py --log2 20 --mutate $MUT --sub $SUB --steps 2 > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }     # also fills the synth cache
grep -v amdgpu $O/bench.log
cd /tmp && export TMPDIR=/tmp
export ZKV_SYNTH_WORKERS=1
B="python3 $R/tools/bench_aggregate.py --log2 20 --mutate $MUT --sub $SUB --no-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o kt -- $B --steps 3 > $O/ktrace.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o pf -- $B --steps 1 > $O/pmc_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o pw -- $B --steps 1 > $O/pmc_write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o ps -- $B --steps 1 > $O/pmc_sq.log 2>&1
rc=$?
cd $R
find $O -name "*.csv" | head -20
exit $rc
