# rocprofv3 kernel TRACE (one row per dispatch, with start / end times and the queue) of the headline bench: bash tools/prof_trace_bench.sh <tag> [bench flags]
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r04t}; shift || true
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --headline-only --no-cpu-baseline --steps 24 "$@" > $O/bench_line.json 2> $O/bench.err
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $T > $R/gpurun_out/${TAG}_bench_timeline.txt
python3 $R/tools/trace_forward.py $T f0_stats_kernel 20 > $R/gpurun_out/${TAG}_bench_one_step.txt
rm -rf "$O/trace"
cat $R/gpurun_out/${TAG}_bench_timeline.txt
