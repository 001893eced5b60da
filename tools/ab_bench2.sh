# more multi-job A/Bs on one box (see ab_bench.sh): generator options that were tuned single-stream
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
run() {
  local label=$1; shift
  env "$@" python3 bench.py --headline-only --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'])"
}
run "default            "
for v in 0 1 2 7; do run "fuse_pair64=$v      " SATOOLS_AMD_GEN_FUSE_PAIR64=$v; done
for v in 1 2 3; do run "branch_streams=$v   " SATOOLS_AMD_GEN_BRANCH_STREAMS=$v; done
run "pair64w=1          " SATOOLS_AMD_CONV_OPTIONS=pair64w=1
run "lean_balance=0     " SATOOLS_AMD_CONV_OPTIONS=lean_balance=0
run "lean_balance=2     " SATOOLS_AMD_CONV_OPTIONS=lean_balance=2
run "lean3=0            " SATOOLS_AMD_CONV_OPTIONS=lean3=0
run "lean7=0            " SATOOLS_AMD_CONV_OPTIONS=lean7=0
run "lean11=0           " SATOOLS_AMD_CONV_OPTIONS=lean11=0
run "default            "
