#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the default bench line, the rocprofv3 kernel-trace summary of the
# same command, and the two separate PMC passes (FETCH_SIZE, WRITE_SIZE).  Everything lands under
# gpurun_out/; tools/pmc_summarize.py turns it into the files committed under profiles/.
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh'           (all four codecs)
#   gpurun --timeout 400 -- 'bash tools/profile_round.sh zstd'       (one of lz4 | zstd | brotli | snappy)
ONLY=${1:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
if [ $ONLY = all ] || [ $ONLY = lz4 ]; then
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
python bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- \
    python bench.py --only --steps 2 --warmup 1 --no-cpu > $O/bench_prof.json 2> $O/prof_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_fetch -- \
    python bench.py --only --steps 1 --warmup 0 --no-cpu > $O/bench_fetch.json 2> $O/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_write -- \
    python bench.py --only --steps 1 --warmup 0 --no-cpu > $O/bench_write.json 2> $O/prof_write.err
cat $O/bench_default.json
fi
if [ $ONLY = all ] || [ $ONLY = zstd ]; then
# configs[3]: zstd-mt level 1 (same workload text, 1 MiB chunks)
rm -rf $O/prof_zstd_stats $O/prof_zstd_fetch $O/prof_zstd_write
python bench.py --only --codec zstd > $O/bench_zstd.json 2> $O/bench_zstd.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_zstd_stats -- \
    python bench.py --only --codec zstd --steps 2 --warmup 1 --no-cpu > $O/bench_zstd_prof.json 2> $O/prof_zstd_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_zstd_fetch -- \
    python bench.py --only --codec zstd --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/prof_zstd_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_zstd_write -- \
    python bench.py --only --codec zstd --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/prof_zstd_write.err
cat $O/bench_zstd.json
fi
if [ $ONLY = all ] || [ $ONLY = zstd ] || [ $ONLY = zref ]; then
# zstd-mt decompress of reference-written level-1 streams (the leg of the default line): sequence pre-pass + frame decoder
rm -rf $O/prof_zref_stats $O/prof_zref_fetch $O/prof_zref_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_zref_stats -- \
    python bench.py --zref-only --gib 8 --steps 2 --warmup 1 --no-cpu > $O/bench_zref_prof.json 2> $O/prof_zref_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_zref_fetch -- \
    python bench.py --zref-only --gib 8 --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/prof_zref_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_zref_write -- \
    python bench.py --zref-only --gib 8 --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/prof_zref_write.err
cat $O/bench_zref_prof.json
fi
if [ $ONLY = all ] || [ $ONLY = brotli ]; then
# configs[4]: brotli-mt decompress (level-1 streams written by the reference build, 1 MiB chunks)
rm -rf $O/prof_brotli_stats $O/prof_brotli_fetch $O/prof_brotli_write
python bench.py --only --codec brotli > $O/bench_brotli.json 2> $O/bench_brotli.err
# (the decoder alone: the device-encoder leg of the bench would mix a second workload into its averages)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_brotli_stats -- \
    python bench.py --only --codec brotli --steps 2 --warmup 1 --no-cpu --no-encoder > $O/bench_brotli_prof.json 2> $O/prof_brotli_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_brotli_fetch -- \
    python bench.py --only --codec brotli --steps 1 --warmup 0 --no-cpu --no-encoder > /dev/null 2> $O/prof_brotli_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_brotli_write -- \
    python bench.py --only --codec brotli --steps 1 --warmup 0 --no-cpu --no-encoder > /dev/null 2> $O/prof_brotli_write.err
# the device encoder + the decode of its own streams (zmt_brotli_enc_kernel lines)
rm -rf $O/prof_brotli_enc_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_brotli_enc_stats -- \
    python bench.py --only --codec brotli --steps 1 --warmup 0 --no-cpu > /dev/null 2> $O/prof_brotli_enc_stats.err
cat $O/bench_brotli.json
fi
if [ $ONLY = all ] || [ $ONLY = snappy ]; then
# snappy-mt round trip at the default 64 KiB chunk (not a BASELINE config).  SNAPPY_DEC=1 selects
# the batched decoder
rm -rf $O/prof_snappy_stats $O/prof_snappy_fetch $O/prof_snappy_write
SD=${SNAPPY_DEC:-0}
python bench.py --codec snappy --snappy-dec $SD > $O/bench_snappy.json 2> $O/bench_snappy.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_snappy_stats -- \
    python bench.py --codec snappy --snappy-dec $SD --steps 2 --warmup 1 > $O/bench_snappy_prof.json 2> $O/prof_snappy_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_snappy_fetch -- \
    python bench.py --codec snappy --snappy-dec $SD --steps 1 --warmup 0 > /dev/null 2> $O/prof_snappy_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_snappy_write -- \
    python bench.py --codec snappy --snappy-dec $SD --steps 1 --warmup 0 > /dev/null 2> $O/prof_snappy_write.err
cat $O/bench_snappy.json
fi
