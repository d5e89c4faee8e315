# usage: bash tools/prof_one.sh <tag> <bench_models args...>   -> gpurun_out/prof_<tag>/ (rocprofv3 kernel stats)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/tools/bench_models.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1
grep coalitions $R/gpurun_out/prof_$tag.log
