# rocprofv3 kernel stats of the headline bench run: bash tools/prof_bench.sh <tag>
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r02k}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --headline-only --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $S $R/gpurun_out/${TAG}_bench_headline_kernel_stats.csv
rm -rf "$O/stats"
tail -c 600 $O/bench_line.json
