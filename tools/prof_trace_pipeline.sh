# rocprofv3 kernel TRACE of the batch job (tools/bench_pipeline.py: wav files in, PCM16 files out, four jobs): what the GPU does in its
# steady state — kernels in flight, idle share — next to the same figures of bench.py's loop (tools/prof_trace_bench.sh)
#   bash tools/prof_trace_pipeline.sh <tag> [utterances]
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r05p}
N=${2:-4096}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/bench_pipeline.py $N 4 32 > $O/bench_pipeline.log 2> $O/bench_pipeline.err
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $T 64 > $R/gpurun_out/${TAG}_pipeline_timeline.txt
rm -rf "$O/trace"
grep -v amdgpu $O/bench_pipeline.log | tail -3
cat $R/gpurun_out/${TAG}_pipeline_timeline.txt
