#!/bin/bash
# rocprofv3 --kernel-trace --stats of any command, summarised per (kernel, grid) with scripts/summarize_trace.py.
#   usage (GPU box): bash scripts/profile_cmd.sh <tag> <command...>   ->  gpurun_out/<tag>_kernel_stats.csv, <tag>_per_shape.md, <tag>.log
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
( cd "$R" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- "$@" ) > "$R/gpurun_out/${tag}.log" 2>&1
st=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
tr=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$st" ] && cp "$st" "$R/gpurun_out/${tag}_kernel_stats.csv"
[ -n "$tr" ] && python "$R/scripts/summarize_trace.py" "$tr" > "$R/gpurun_out/${tag}_per_shape.md"
head -${HEAD:-30} "$R/gpurun_out/${tag}_per_shape.md"
