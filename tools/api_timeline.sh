# developer tool: kernel + copy timeline of one API leg (rocprofv3, no counters)
mkdir -p gpurun_out/r2/tl
cd /tmp && export TMPDIR=/tmp
export LD_LIBRARY_PATH=/root/repo/zstdmt_amd/lib
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /root/repo/gpurun_out/r2/tl -o ${1:-lz4api} -- /root/repo/zstdmt_amd/bin/api_bench ${2:-lz4} ${3:-2147483648} ${4:-131072} > /root/repo/gpurun_out/r2/tl/run_${1:-lz4api}.log 2>&1
tail -1 /root/repo/gpurun_out/r2/tl/run_${1:-lz4api}.log
