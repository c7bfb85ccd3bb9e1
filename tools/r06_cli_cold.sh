#!/bin/bash
# Runs ON THE GPU BOX: the lz4-mt command line tool, a fresh process per run (pinned / device buffers allocated cold), 4 GiB of the
# bench text from /dev/shm at the reference's default chunk (4 MiB): batches of 256 MiB (the old fixed size) against the chunk-aware default
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
python - <<PY
import ctypes as C, numpy as np
T = C.CDLL("zstdmt_amd/lib/libzmt_tools.so"); T.zmt_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
n = 4 << 30; hb = np.empty(n, np.uint8); T.zmt_gen_text(hb.ctypes.data, n, 20260926, 0, 32); hb.tofile("/dev/shm/zmt_cli_in")
PY
{
for rep in 1 2; do for mb in 256 ""; do for b in "" "-b 1"; do
  s=$(date +%s.%N)
  GPUMT_BATCH_MB=$mb zstdmt_amd/bin/lz4-mt -1 -T4 $b -c /dev/shm/zmt_cli_in > /dev/shm/zmt_cli_out
  e=$(date +%s.%N)
  GPUMT_BATCH_MB=$mb zstdmt_amd/bin/lz4-mt -d -T4 -c /dev/shm/zmt_cli_out > /dev/shm/zmt_cli_back
  f=$(date +%s.%N)
  echo "GPUMT_BATCH_MB='${mb}' chunk '${b:-default 4 MiB}': compress $(python3 -c "print(round($e - $s, 2))") s, decompress $(python3 -c "print(round($f - $e, 2))") s, $(stat -c %s /dev/shm/zmt_cli_out) bytes, $(cmp /dev/shm/zmt_cli_in /dev/shm/zmt_cli_back && echo same)"
done; done; done
} 2>&1 | tee $O/r06_cli_cold.txt
rm -f /dev/shm/zmt_cli_in /dev/shm/zmt_cli_out /dev/shm/zmt_cli_back
