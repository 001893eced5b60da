# launch-by-launch kernel trace of one ASR forward (batch 32 x 5 s): bash tools/prof_trace_asr.sh <tag>
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:?usage: bash tools/prof_trace_asr.sh <tag>}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/asr_only.py > $O/trace.log 2>&1
S=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_forward.py $S fbank_frames > $R/gpurun_out/${TAG}_asr_forward.txt
rm -rf "$O/trace"
cat $R/gpurun_out/${TAG}_asr_forward.txt
