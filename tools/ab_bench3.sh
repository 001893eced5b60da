set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
run() {
  local label=$1; shift
  env "$@" python3 bench.py --headline-only --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'])"
}
run "default            "
run "k1_gemm=2          " SATOOLS_AMD_CONV_OPTIONS=k1_gemm=2
run "k1_gemm=1          " SATOOLS_AMD_CONV_OPTIONS=k1_gemm=1
run "pair32s_waves=4    " SATOOLS_AMD_CONV_OPTIONS=pair32s_waves=4
run "hwq=8              " GPU_MAX_HW_QUEUES=8
run "hwq=4              " GPU_MAX_HW_QUEUES=4
run "hwq=24             " GPU_MAX_HW_QUEUES=24
run "default            "
