# A/B of two builds of libsatools_hip.so on one box, interleaved: bash tools/ab_builds.sh <other.so> [command...]
# (default command: python tools/gen_time.py; the in-tree build is "B", the other one "A")
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
A=$1; shift
if [ $# -eq 0 ]; then set -- python tools/gen_time.py; fi
for i in 1 2 3; do
  echo -n "A: "; SATOOLS_AMD_LIB=$A "$@" 2>/dev/null | tail -1
  echo -n "B: "; "$@" 2>/dev/null | tail -1
done
