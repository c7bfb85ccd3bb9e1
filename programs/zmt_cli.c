/*
 * zmt_cli.c -- lz4-mt / zstd-mt command line tools over libzstdmt_amd.so.
 *
 * A small gzip-like front end with the option letters, file handling and -B report lines of the
 * reference CLI (/root/reference/programs/main.c:124-165 usage text, :166-170 + :238-243 "-B"
 * statistics line, :970-977 timing line), so that BASELINE.json's configurations can be typed as
 * written ("lz4-mt -1 -T4 FILE").  One source, the codec is chosen at build time (-DZMT_ZSTD, -DZMT_BROTLI, -DZMT_SNAPPY) like
 * the reference does with programs/lz4-mt.c / programs/zstd-mt.c; the personality (compress /
 * decompress / cat) follows argv[0].  -l lists compressed / uncompressed sizes (and crc32 + mtime with -v, -C
 * switches the crc off) in the reference's layout (programs/main.c:383-418).
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

#if defined(ZMT_BROTLI)
#include "brotli-mt.h"
#define PROGNAME "brotli-mt"
#define UNZIP "unbrotli-mt"
#define ZCAT "brotlicat-mt"
#define SUFFIX ".brot"
#define METHOD "brotli"
#define LEVEL_DEF 3
#define LEVEL_MIN BROTLIMT_LEVEL_MIN
#define LEVEL_MAX BROTLIMT_LEVEL_MAX
#define THREAD_MAX BROTLIMT_THREAD_MAX
#define MT(x) BROTLIMT_##x
typedef BROTLIMT_Buffer MT_Buffer;
typedef BROTLIMT_RdWr_t MT_RdWr_t;
#elif defined(ZMT_SNAPPY)
#include "snappy-mt.h"
#define PROGNAME "snappy-mt" /* programs/snappy-mt.c:13-22 */
#define UNZIP "unsnappy-mt"
#define ZCAT "snappycat-mt"
#define SUFFIX ".snp"
#define METHOD "snappy"
#define LEVEL_DEF 0
#define LEVEL_MIN 0
#define LEVEL_MAX 1
#define THREAD_MAX SNAPPYMT_THREAD_MAX
#define MT(x) SNAPPYMT_##x
typedef SNAPPYMT_Buffer MT_Buffer;
typedef SNAPPYMT_RdWr_t MT_RdWr_t;
#elif defined(ZMT_ZSTD)
#include "zstd-mt.h"
#define PROGNAME "zstd-mt"
#define UNZIP "unzstd-mt"
#define ZCAT "zstdcat-mt"
#define SUFFIX ".zst"
#define METHOD "zstd"
#define LEVEL_DEF 3
#define LEVEL_MIN ZSTDCB_LEVEL_MIN
#define LEVEL_MAX ZSTDCB_LEVEL_MAX
#define THREAD_MAX ZSTDCB_THREAD_MAX
#define MT(x) ZSTDCB_##x
typedef ZSTDCB_Buffer MT_Buffer;
typedef ZSTDCB_RdWr_t MT_RdWr_t;
#else
#include "lz4-mt.h"
#define PROGNAME "lz4-mt"
#define UNZIP "unlz4-mt"
#define ZCAT "lz4cat-mt"
#define SUFFIX ".lz4"
#define METHOD "lz4"
#define LEVEL_DEF 3 /* programs/lz4-mt.c:19 (LZ4HC) */
#define LEVEL_MIN LZ4MT_LEVEL_MIN
#define LEVEL_MAX LZ4MT_LEVEL_MAX
#define THREAD_MAX LZ4MT_THREAD_MAX
#define MT(x) LZ4MT_##x
typedef LZ4MT_Buffer MT_Buffer;
typedef LZ4MT_RdWr_t MT_RdWr_t;
#endif

enum { M_COMPRESS, M_DECOMPRESS, M_TEST, M_LIST };

static int o_mode = M_COMPRESS, o_level = LEVEL_DEF, o_threads = 0, o_chunk = 0, o_iter = 1;
static int o_stdout, o_force, o_keep, o_quiet, o_verbose = 1, o_timings; /* verbose: 1 default, -v adds, as the reference */
static const char *o_out, *o_suffix = SUFFIX;
static int exit_code;
/* -l: what the callbacks saw (programs/main.c:172-200) */
static int o_nocrc;
static unsigned long long l_read, l_written;
static unsigned int l_crc;

static unsigned int crc32_update(const unsigned char *buf, size_t size, unsigned int crc)
{
	static unsigned int table[256];
	static int init;
	if (!init) {
		for (unsigned int b = 0; b < 256; b++) {
			unsigned int r = b;
			for (int i = 0; i < 8; i++)
				r = (r & 1) ? (r >> 1) ^ 0xEDB88320U : r >> 1;
			table[b] = r;
		}
		init = 1;
	}
	crc = ~crc;
	while (size--)
		crc = table[(*buf++ ^ crc) & 0xFF] ^ (crc >> 8);
	return ~crc;
}

static void die(const char *msg, const char *arg)
{
	if (!o_quiet)
		fprintf(stderr, "%s: %s%s%s\n", PROGNAME, msg, arg ? ": " : "", arg ? arg : "");
	exit(1);
}

/* callbacks: the reference's ReadData / WriteData over stdio (programs/main.c:172-200) */
static int rd(void *arg, MT_Buffer *b)
{
	FILE *f = (FILE *)arg;
	size_t got = fread(b->buf, 1, b->size, f);
	if (got != b->size && ferror(f))
		return -1;
	b->size = got;
	if (o_mode == M_LIST)
		l_read += got;
	return 0;
}
static int wr(void *arg, MT_Buffer *b)
{
	FILE *f = (FILE *)arg;
	if (o_mode == M_LIST) {
		if (o_verbose > 1 && !o_nocrc)
			l_crc = crc32_update((const unsigned char *)b->buf, b->size, l_crc);
		l_written += b->size;
		return 0;
	}
	if (!f)
		return 0; /* -t: discard */
	return fwrite(b->buf, 1, b->size, f) == b->size ? 0 : -1;
}

static void usage(void)
{
	printf("\n Usage: %s [OPTION]... [FILE]...\n"
	       " Compress or uncompress FILEs on an MI355X (by default, compress FILEs in-place).\n\n"
	       "  -#    compression level (%d-%d, default %d)\n"
	       "  -c    write to standard output\n"
	       "  -d    decompress     -z    compress     -t    test integrity\n"
	       "  -f    overwrite existing files\n"
	       "  -o F  write output to file F ('-' = stdout)\n"
	       "  -k    keep input files\n"
	       "  -l    list compressed / uncompressed sizes (-v: method, crc32, date; -C: no crc32)\n"
	       "  -L    display the license     \n"
	       "  -q    quiet     -v    verbose     -h    this help     -V    version\n"
	       "  -S X  suffix of compressed files (default \"%s\")\n"
	       "  -T N  number of threads (validated, 1-%d; the work runs on the GPU)\n"
	       "  -b N  input chunk size in MiB (default: the codec's)\n"
	       "  -i N  iterations (files only)\n"
	       "  -B    print 'Level;Threads;InSize;OutSize;Frames' per file and 'Real;User;Sys;MaxMem'\n\n"
	       " Invoked as '%s': decompress; as '%s': decompress to stdout.\n"
	       " With no FILE, or when FILE is -, read standard input.\n\n",
	       PROGNAME, LEVEL_MIN, LEVEL_MAX, LEVEL_DEF, SUFFIX, THREAD_MAX, UNZIP, ZCAT);
	exit(0);
}

/* one stream through the library; returns NULL or an error text */
static const char *run(FILE *in, FILE *out)
{
	static int first = 1;
	MT_RdWr_t io;
	size_t rv;
	if (first && o_timings) {
		fprintf(stderr, "Level;Threads;InSize;OutSize;Frames\n");
		first = 0;
	}
	io.fn_read = rd;
	io.arg_read = in;
	io.fn_write = wr;
	io.arg_write = out;
	if (o_mode == M_COMPRESS) {
		MT(CCtx) *c = MT(createCCtx)(o_threads, o_level, o_chunk);
		if (!c)
			return "Allocating compression context failed (bad arguments or no gfx950 device)!";
		rv = MT(compressCCtx)(c, &io);
		if (MT(isError)(rv)) {
			MT(freeCCtx)(c);
			return MT(getErrorString)(rv);
		}
		if (o_timings)
			fprintf(stderr, "%d;%d;%lu;%lu;%lu\n", o_level, o_threads, (unsigned long)MT(GetInsizeCCtx)(c),
				(unsigned long)MT(GetOutsizeCCtx)(c), (unsigned long)MT(GetFramesCCtx)(c));
		MT(freeCCtx)(c);
	} else {
		MT(DCtx) *d = MT(createDCtx)(o_threads, o_chunk);
		if (!d)
			return "Allocating decompression context failed (bad arguments or no gfx950 device)!";
		rv = MT(decompressDCtx)(d, &io);
		if (MT(isError)(rv)) {
			MT(freeDCtx)(d);
			return MT(getErrorString)(rv);
		}
		if (o_timings)
			fprintf(stderr, "%d;%d;%lu;%lu;%lu\n", o_level, o_threads, (unsigned long)MT(GetInsizeDCtx)(d),
				(unsigned long)MT(GetOutsizeDCtx)(d), (unsigned long)MT(GetFramesDCtx)(d));
		MT(freeDCtx)(d);
	}
	return NULL;
}

static int ends_with(const char *s, const char *suf)
{
	size_t a = strlen(s), b = strlen(suf);
	return a >= b && !strcmp(s + a - b, suf);
}

static void do_file(const char *name)
{
	char outname[4096];
	FILE *in, *out = NULL;
	const char *err;
	int to_stdout = o_stdout || (o_out && !strcmp(o_out, "-"));
	if (!strcmp(name, "-")) {
		in = stdin;
		to_stdout = to_stdout || !o_out;
	} else if (!(in = fopen(name, "rb"))) {
		if (!o_quiet)
			fprintf(stderr, "%s: %s: %s\n", PROGNAME, name, strerror(errno));
		exit_code = 1;
		return;
	}
	outname[0] = 0;
	if (o_mode == M_LIST) {
		/* decode to nowhere, count (reference: MODE_LIST writes to /dev/null, main.c:932-937) */
		static int first_row = 1;
		struct stat stt;
		time_t mt = time(NULL);
		l_read = l_written = 0;
		l_crc = 0;
		if (in != stdin && fstat(fileno(in), &stt) == 0)
			mt = stt.st_mtime;
		err = run(in, NULL);
		if (in != stdin)
			fclose(in);
		if (first_row && o_verbose > 1)
			printf("%8s %8s %10s %8s %20s %20s %7s %s\n", "method", "crc32", "date", "time", "compressed",
			       "uncompressed", "ratio", "uncompressed_name");
		else if (first_row)
			printf("%20s %20s %7s %s\n", "compressed", "uncompressed", "ratio", "uncompressed_name");
		first_row = 0;
		if (err) {
			if (!o_quiet)
				fprintf(stderr, "%s: %s: %s\n", PROGNAME, name, err);
			exit_code = 1;
			if (o_verbose > 1)
				printf("%8s %8s %10s %8s %20s %20s %7s %s\n", "-", "-", "-", "-", "-", "-", "-", name);
			else
				printf("%20s %20s %7s %s\n", "-", "-", "-", name);
		} else if (o_verbose > 1) {
			char tb[30];
			strftime(tb, sizeof tb, "%Y-%m-%d %H:%M:%S", localtime(&mt));
			printf("%8s %08x %12s %20llu %20llu %6.2f%% %s\n", METHOD, l_crc, tb, l_read, l_written,
			       l_written ? 100 - (double)l_read * 100 / (double)l_written : 0.0, name);
		} else {
			printf("%20llu %20llu %6.2f%% %s\n", l_read, l_written,
			       l_written ? 100 - (double)l_read * 100 / (double)l_written : 0.0, name);
		}
		return;
	}
	if (o_mode == M_TEST) {
		out = NULL;
	} else if (to_stdout) {
		out = stdout;
	} else {
		if (o_out) {
			snprintf(outname, sizeof outname, "%s", o_out);
		} else if (o_mode == M_COMPRESS) {
			if (ends_with(name, o_suffix) && !o_force) {
				if (!o_quiet)
					fprintf(stderr, "%s: %s already has %s suffix -- unchanged\n", PROGNAME, name, o_suffix);
				fclose(in);
				return;
			}
			snprintf(outname, sizeof outname, "%s%s", name, o_suffix);
		} else {
			if (!ends_with(name, o_suffix)) {
				if (!o_quiet)
					fprintf(stderr, "%s: %s: unknown suffix -- ignored\n", PROGNAME, name);
				fclose(in);
				exit_code = 1;
				return;
			}
			snprintf(outname, sizeof outname, "%.*s", (int)(strlen(name) - strlen(o_suffix)), name);
		}
		if (!o_force && access(outname, F_OK) == 0) {
			if (!o_quiet)
				fprintf(stderr, "%s: %s already exists; not overwritten (use -f)\n", PROGNAME, outname);
			if (in != stdin)
				fclose(in);
			exit_code = 1;
			return;
		}
		if (!(out = fopen(outname, "wb")))
			die(strerror(errno), outname);
	}
	err = run(in, out);
	if (in != stdin)
		fclose(in);
	if (out && out != stdout && fclose(out))
		err = "write error";
	if (out == stdout)
		fflush(stdout);
	if (err) {
		if (!o_quiet)
			fprintf(stderr, "%s: %s: %s\n", PROGNAME, name, err);
		if (outname[0])
			remove(outname);
		exit_code = 1;
		return;
	}
	if (o_verbose > 1 && !o_quiet)
		fprintf(stderr, "%s: %s\n", name, o_mode == M_TEST ? "OK" : "done");
	if (outname[0] && !o_keep && !o_out && in != stdin && o_mode != M_TEST)
		remove(name);
}

int main(int argc, char **argv)
{
	const char *prog = strrchr(argv[0], '/');
	struct timeval t0, t1, dt;
	struct rusage ru;
	int opt, level_set = 0;
	prog = prog ? prog + 1 : argv[0];
	/* A command line run is a fresh process: every pinned buffer it uses is allocated cold (~0.3 s per GiB), which the library's
	 * chunk-aware batch size -- built for contexts that live on -- does not earn back on one file (4 GiB at the default 4 MiB chunk:
	 * 2.2 s with 256 MiB batches, 3.7 s with the library's 1 GiB; profiles/r06_sweeps/api_chunk_sizes.txt).  GPUMT_BATCH_MB overrides. */
	setenv("GPUMT_BATCH_MB", "256", 0);
	if (!strcmp(prog, UNZIP)) {
		o_mode = M_DECOMPRESS;
	} else if (!strcmp(prog, ZCAT)) {
		o_mode = M_DECOMPRESS;
		o_stdout = 1;
		o_force = 1;
	}
	while ((opt = getopt(argc, argv, "0123456789cdzfo:hklLqS:tvVT:b:i:BC")) != -1) {
		switch (opt) {
		case '0': case '1': case '2': case '3': case '4':
		case '5': case '6': case '7': case '8': case '9':
			o_level = (level_set ? o_level * 10 : 0) + (opt - '0');
			level_set = 1;
			break;
		case 'c': o_stdout = 1; o_keep = 1; break;
		case 'd': o_mode = M_DECOMPRESS; break;
		case 'z': o_mode = M_COMPRESS; break;
		case 'f': o_force = 1; break;
		case 'o': o_out = optarg; break;
		case 'h': usage(); break;
		case 'k': o_keep = 1; break;
		case 'q': o_quiet = 1; o_verbose = 0; break;
		case 'S': o_suffix = optarg; break;
		case 't': o_mode = M_TEST; break;
		case 'v': o_verbose++; break;
		case 'V': printf("%s (zstdmt_amd, MI355X)\n", PROGNAME); return 0;
		case 'T': o_threads = atoi(optarg); break;
		case 'b': o_chunk = atoi(optarg) * 1024 * 1024; break;
		case 'i': o_iter = atoi(optarg); break;
		case 'B': o_timings = 1; break;
		case 'l': o_mode = M_LIST; o_keep = 1; break;
		case 'C': o_nocrc = 1; break;
		case 'L':
			printf("%s (zstdmt_amd, MI355X): command line front end of libzstdmt_amd.so.\n"
			       "Written for this repository; the option letters and report layouts follow the\n"
			       "BSD-licensed zstdmt tools (https://github.com/mcmilk/zstdmt), whose code it does not contain.\n",
			       PROGNAME);
			return 0;
		default:
			usage();
		}
	}
	if (o_level < LEVEL_MIN || o_level > LEVEL_MAX)
		die("compression level out of range", NULL);
	if (o_threads == 0) {
		long n = sysconf(_SC_NPROCESSORS_ONLN); /* default as programs/main.c:758 */
		o_threads = n < 1 ? 1 : n > THREAD_MAX ? THREAD_MAX : (int)n;
	}
	if (o_threads < 1 || o_threads > THREAD_MAX)
		die("number of threads out of range", NULL);
	if (o_iter < 1)
		o_iter = 1;
	gettimeofday(&t0, NULL);
	if (optind >= argc) {
		if (o_iter != 1)
			die("You can not use stdin together with the -i option.", NULL);
		if (o_mode < M_TEST && !o_out && !o_force && isatty(fileno(stdout)))
			die("refusing to write binary data to a terminal (use -f or -c)", NULL);
		do_file("-");
	} else {
		for (int it = 0; it < o_iter; it++)
			for (int i = optind; i < argc; i++)
				do_file(argv[i]);
	}
	if (o_timings) {
		gettimeofday(&t1, NULL);
		timersub(&t1, &t0, &dt);
		getrusage(RUSAGE_SELF, &ru);
		fprintf(stderr, "Real;User;Sys;MaxMem\n%ld.%03ld;%ld.%03ld;%ld.%03ld;%ld\n", (long)dt.tv_sec,
			(long)dt.tv_usec / 1000, (long)ru.ru_utime.tv_sec, (long)ru.ru_utime.tv_usec / 1000,
			(long)ru.ru_stime.tv_sec, (long)ru.ru_stime.tv_usec / 1000, (long)ru.ru_maxrss);
	}
	return exit_code;
}
