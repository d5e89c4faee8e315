# usage (GPU box): bash tools/pmc_stream.sh -> gpurun_out/pmc_stream/{fetch,write,tcc}: three counter passes of tools/stream_kernels.py
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_stream/fetch -- python3 $R/tools/stream_kernels.py > $R/gpurun_out/pmc_stream.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_stream/write -- python3 $R/tools/stream_kernels.py > $R/gpurun_out/pmc_stream.write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_stream/tcc -- python3 $R/tools/stream_kernels.py > $R/gpurun_out/pmc_stream.tcc.log 2>&1
tail -3 $R/gpurun_out/pmc_stream.tcc.log
