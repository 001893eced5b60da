# multi-job A/B of the round-4 kernel choices on ONE box: the headline bench under different dispatch options and jobs in flight
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
run() {
  local label=$1; local jobs=$2; shift; shift
  env "$@" python3 bench.py --headline-only --no-cpu-baseline --jobs $jobs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label jobs $jobs', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do
  for jobs in 1 2 3 4 6; do
    run "default       " $jobs
    run "convring=0    " $jobs SATOOLS_AMD_CONV_OPTIONS=convring=0
  done
  run "multi_branch=0" 4 SATOOLS_AMD_GEN_MULTI_BRANCH=0
done
