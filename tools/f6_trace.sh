#!/bin/bash
# Dev tool: kernel trace of batches of 256 queries through the int8 pass and through the FP6 first filter (tools/pass_pmc_run.py),
# per-kernel totals -> gpurun_out/f6_trace_<rows>.txt.   usage: bash tools/f6_trace.sh [rows=100000000]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROWS=${1:-100000000}
mkdir -p $R/gpurun_out
rm -rf /tmp/f6t
rocprofv3 --kernel-trace -d /tmp/f6t -o pass -- python3 $R/tools/pass_pmc_run.py $ROWS > $R/gpurun_out/f6_trace.out 2>&1
python3 $R/tools/stats_by_kernel.py $(find /tmp/f6t -name "*.db" | head -1) > $R/gpurun_out/f6_trace_$ROWS.txt 2>&1
cat $R/gpurun_out/f6_trace_$ROWS.txt | cut -c1-150
