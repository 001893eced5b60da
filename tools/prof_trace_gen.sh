# per-dispatch kernel trace of tools/gen_only.py, grouped by (kernel, grid): bash tools/prof_trace_gen.sh <tag>
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:?usage: bash tools/prof_trace_gen.sh <tag>}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/gen_only.py > $O/trace.log 2>&1
S=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_by_grid.py $S 13 > $R/gpurun_out/${TAG}_by_grid.txt
rm -rf "$O/trace"
head -24 $R/gpurun_out/${TAG}_by_grid.txt
