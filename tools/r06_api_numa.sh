#!/bin/bash
# Runs ON THE GPU BOX: the LZ4MT API legs with the process bound to each NUMA node of the host in turn (is the reader's memcpy
# into the pinned batch buffer slower than the callbacks alone because the buffer lives on the GPU's node and the caller elsewhere?)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
{
echo "# nodes: $(ls -d /sys/devices/system/node/node* | wc -l); GPU numa_node: $(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ')"
which numactl
for nd in /sys/devices/system/node/node*; do
  n=${nd##*node}; cpus=$(cat $nd/cpulist)
  echo "== node $n cpus $cpus"
  GPUMT_TRACE=1 ZMT_API_BOUND=1 taskset -c $cpus zstdmt_amd/bin/api_bench lz4 ${1:-4294967296} 131072 zstdmt_amd/lib/libzstdmt_amd.so 1 2>&1 | grep -v "^\[mt_pipe\] 33 batches in [0-9.]* s | reader: fill [0-9.]* wait [0-9.]* | device: launch 0.[23]" | tail -5
done
} > $O/r06_api_numa.txt 2>&1
cat $O/r06_api_numa.txt
