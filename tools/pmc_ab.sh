# usage (GPU box): bash tools/pmc_ab.sh <tag> <counter list in quotes> <bench_models args...>
# one rocprofv3 counter pass (kernel trace only) of tools/bench_models.py -> gpurun_out/pmc_<tag>/
set -e
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/bench_models.py "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
tail -1 $R/gpurun_out/pmc_$tag.log
