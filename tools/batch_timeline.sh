#!/bin/bash
# kernel timeline of a topical batch (dev tool): bash tools/batch_timeline.sh [rows] [dist] [k]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/timeline
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -d /tmp/prof_tl -o tl -- python3 $R/tools/batch_timeline.py ${1:-100000000} ${2:-4} ${3:-10} 2 ${5:-256} > $R/gpurun_out/timeline/run.log 2>&1
tail -2 $R/gpurun_out/timeline/run.log
DB=$(find /tmp/prof_tl -name "*.db" | head -1)
python3 $R/tools/profile_summary.py timeline $DB $R/gpurun_out/timeline ${4:-70}
