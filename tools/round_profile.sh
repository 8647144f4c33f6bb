#!/bin/bash
# The round's evidence run on one MI355X box (dev tool): GPU test suite, the default bench line, and the two rocprofv3 passes of
# the bench command (kernel trace + stats; FETCH_SIZE in a pass of its own) summarised into gpurun_out/prof/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/prof
cd $R
if [ "$1" != "noprofile" ]; then
python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
fi
python bench.py > gpurun_out/prof/bench_100M_n1.json 2> gpurun_out/bench_n1.err; cp bench_extra.json gpurun_out/prof/bench_extra.json; tail -c 600 gpurun_out/prof/bench_100M_n1.json; echo
[ "$1" == "noprofile" ] && exit 0
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_stats /tmp/prof_pmc
rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o bench -- python3 $R/bench.py --steps 5 --warmup 2 > $R/gpurun_out/prof/bench_under_rocprof.json 2> $R/gpurun_out/rocprof_stats.err
DB=$(find /tmp/prof_stats -name "*.db" | head -1); echo "stats db: $DB"
python3 $R/tools/profile_summary.py stats $DB $R/gpurun_out/prof
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_pmc -o bench -- python3 $R/bench.py --steps 5 --warmup 2 > /dev/null 2> $R/gpurun_out/rocprof_pmc.err
DB=$(find /tmp/prof_pmc -name "*.db" | head -1); echo "pmc db: $DB"
python3 $R/tools/profile_summary.py pmc $DB $R/gpurun_out/prof | tee $R/gpurun_out/prof/pmc_groups.txt
