# developer tool: the API legs with the pipeline trace (run on the GPU box)
mkdir -p gpurun_out/r2
B=zstdmt_amd/bin/api_bench
export LD_LIBRARY_PATH=zstdmt_amd/lib
N=8589934592
{
echo "== lz4"; GPUMT_TRACE=1 $B lz4 $N 131072 2>&1 | tail -4
echo "== lz4 hwq 4"; GPU_MAX_HW_QUEUES=4 GPUMT_TRACE=1 $B lz4 $N 131072 2>&1 | tail -4
echo "== zstd"; GPUMT_TRACE=1 $B zstd $N 0 2>&1 | tail -4
echo "== brotli"; GPUMT_TRACE=1 $B brotli $N 0 2>&1 | tail -4
} > gpurun_out/r2/api_trace3.txt
cat gpurun_out/r2/api_trace3.txt
