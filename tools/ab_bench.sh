# multi-job A/B of kernel choices on ONE box: the headline bench (four convert() jobs in flight) under different dispatch
# options.  bash tools/ab_bench.sh            (each line: options, x real-time, ms per step)
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
run() {
  local label=$1; shift
  env "$@" python3 bench.py --headline-only --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do
  run "default            "
  run "fuse_mrf=0         " SATOOLS_AMD_GEN_FUSE_MRF=0
  run "ups2=0             " SATOOLS_AMD_GEN_UPS2=0
  run "pair32s=0          " SATOOLS_AMD_CONV_OPTIONS=pair32s=0
  run "pair32w=0          " SATOOLS_AMD_CONV_OPTIONS=pair32w=0
  run "all round-3 off    " SATOOLS_AMD_GEN_FUSE_MRF=0 SATOOLS_AMD_GEN_UPS2=0 SATOOLS_AMD_CONV_OPTIONS=pair32s=0,pair32w=0
done
