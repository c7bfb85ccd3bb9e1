/* CPU harness for the host-side frame splitters of the plain-stream paths (tests/test_host_extent.py):
 * lz4_frame_extent (lz4mt_engine.c) and zstd_frame_extent (zstdmt_engine.c) are static, so the engine
 * sources are included here; nothing that touches the device is called (linked with unresolved
 * gpumt_* symbols ignored).
 *   extent_harness lz4|zstd FILE      -> one line per frame: offset length bound flag
 *   extent_harness lz4|zstd FILE cut  -> number of proper prefixes of the first frame that are (wrongly)
 *                                        accepted as a complete frame: must print 0 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef HARNESS_ZSTD
#include "../../zstdmt_amd/csrc/host/zstdmt_engine.c"
#define EXTENT zstd_frame_extent
#else
#include "../../zstdmt_amd/csrc/host/lz4mt_engine.c"
#define EXTENT lz4_frame_extent
#endif

int main(int argc, char **argv)
{
	if (argc < 2)
		return 2;
	FILE *f = fopen(argv[1], "rb");
	if (!f)
		return 3;
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	uint8_t *buf = (uint8_t *)malloc((size_t)n + 16);
	if (fread(buf, 1, (size_t)n, f) != (size_t)n)
		return 4;
	fclose(f);
	if (argc > 2 && !strcmp(argv[2], "verdict")) {
		/* one word: what the incremental reader does with these bytes as the start of a frame */
		uint64_t bound = 0;
		int flag = 0;
		const size_t r = EXTENT(buf, (size_t)n, &bound, &flag);
		printf("%s\n", r == 0 ? "wait" : r == (size_t)-1 ? "invalid" : "frame");
		return 0;
	}
	if (argc > 2) {
		uint64_t bound = 0;
		int flag = 0;
		const size_t full = EXTENT(buf, (size_t)n, &bound, &flag);
		long wrong = 0;
		for (size_t k = 0; k < full; k++) {
			bound = 0;
			flag = 0;
			if (EXTENT(buf, k, &bound, &flag) != 0) /* neither complete nor invalid: a prefix of a good frame only waits */
				wrong++;
		}
		printf("%zu %ld\n", full, wrong);
		return 0;
	}
	size_t off = 0;
	while (off < (size_t)n) {
		uint64_t bound = 0;
		int flag = 0;
		const size_t len = EXTENT(buf + off, (size_t)n - off, &bound, &flag);
		printf("%zu %zu %llu %d\n", off, len, (unsigned long long)bound, flag);
		if (!len)
			break;
		off += len;
	}
	return 0;
}
