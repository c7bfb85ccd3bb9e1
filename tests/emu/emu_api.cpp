/*
 * emu_api.cpp -- TEST HARNESS ONLY.  C entry points that run the repo's HIP kernels (compiled as
 * host C++ with -DZMT_EMU) under the fiber harness, mirroring the gpumt_* batch calls so the same
 * python parity checks can drive either.  Host pointers stand in for device pointers.
 */
#include "emu_runtime.h"

#include <algorithm>

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

extern "C" {
void zmt_xxh32_kernel(const u8 *, const u64 *, const u32 *, u32, u32 *, const u32 *, const u32 *, u32 *);
void zmt_lz4_enc3_u16_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *, unsigned long long *);
void zmt_lz4_enc3_p17_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *, unsigned long long *);
void zmt_lz4_enc3_u32_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *, unsigned long long *);
void zmt_lz4_enc5_u16_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *, unsigned long long *);
void zmt_lz4_enc5_p17_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *, unsigned long long *);
void zmt_lz4_enc5_u32_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *, unsigned long long *);
void zmt_lz4hc_enc_kernel(const u8 *, u64, u32, u32, u8 *, u64, u32 *, const u32 *, u8 *, int);
void zmt_lz4_dec_serial(const u8 *, const u64 *, const u32 *, u32, u8 *, const u64 *, u32 *, u32 *, u32 *, u32 *, u32);
void zmt_dec_nblk_kernel(const u32 *, u32, u32 *);
void zmt_dec_frames_kernel(const u8 *, const u64 *, const u32 *, u32, const u32 *, const u64 *, u64 *, u32 *, u32 *, u32 *, u32 *, u32 *, u32 *);
void zmt_dec_parse4_kernel(const u8 *, u64, const u64 *, const u32 *, const u64 *, uint16_t *, u32 *, u32 *);
#define C3_DECL(NAME) void NAME(const u8 *, u64, u32, u32, u8 *, const u64 *, const u32 *, const u64 *, const u64 *, const u32 *, const u32 *, const u32 *, const uint16_t *, const u32 *, const u32 *, u32 *);
C3_DECL(zmt_dec_copy3_w4_kernel)
C3_DECL(zmt_dec_copy3_w8_kernel)
C3_DECL(zmt_dec_copy3_w16_kernel)
void zmt_probe_kernel(const u8 *, const u64 *, const u32 *, u32, u32 *);
void zmt_scan_kernel(const u32 *, u32, u64 *);
void zmt_compact_kernel(const u8 *, u64, const u32 *, const u64 *, u32, u8 *);
void zmt_zstd_dec_kernel(const u8 *, u64, const u64 *, const u32 *, u32, u8 *, const u64 *, u32 *, u8 *, u32 *, u32 *, u32 *, u32, u8 *, u64);
void zmt_zstd_seq_kernel(const u8 *, u64, const u64 *, const u32 *, u32, const u64 *, const u32 *, const u32 *, u8 *, u64);
void zmt_brotli_enc_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
void zmt_brotli_enc_t2_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
void zmt_brotli_enc_t3_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
void zmt_brotli_assemble_kernel(u64, u32, u32, u32, u8 *, u64, const u32 *, u32 *);
void zmt_brotli_dec_kernel(const u8 *, const u64 *, const u32 *, u32, u8 *, const u64 *, const u32 *, u32 *, u32 *, u8 *, const u8 *, u32);
void zmt_brotli_dec4_kernel(const u8 *, const u64 *, const u32 *, u32, u8 *, const u64 *, const u32 *, u32 *, u32 *);
void zmt_zstd_dec_small_kernel(const u8 *, u64, const u64 *, const u32 *, u32, u8 *, const u64 *, u32 *, u8 *, u32 *, u32 *, u32 *, u8 *, u64);
void zmt_xxh64_verify_kernel(const u8 *, const u64 *, const u32 *, u32, const u32 *, const u32 *, u32 *);
void zmt_zstd_probe_kernel(const u8 *, const u64 *, const u32 *, u32, u32 *, u32 *);
void zmt_zstd_enc_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
void zmt_zstd_enc_t2_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
void zmt_zstd_enc_t3_kernel(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, u8 *);
void zmt_zstd_assemble_kernel(u64, u32, u32, u32, u8 *, u64, const u32 *, u32 *);
void zmt_snappy_enc_kernel(const u8 *, u64, u32, u32, u8 *, u64, u32 *);
void zmt_snappy_dec_kernel(const u8 *, const u64 *, const u32 *, u32, u8 *, const u64 *, const u32 *, u32 *, u32 *);
void zmt_snappy_dec2_kernel(const u8 *, const u64 *, const u32 *, u32, u8 *, const u64 *, const u32 *, u32 *, u32 *);
}

using emu::dim3;

extern "C" {

size_t emu_lz4_slot_stride(size_t chunk)
{
	size_t full = chunk / 65536, part = chunk % 65536;
	size_t b = 12 + 19 + 4 * (full + (part ? 1 : 0)) + chunk + 8;
	return (b + 255) & ~(size_t)255;
}

void emu_xxh32_batch(const u8 *base, const u64 *off, const u32 *len, u32 n, u32 *out)
{
	emu::launch(dim3{(n * 4 + 255) / 256, 1, 1}, dim3{256, 1, 1},
		    [=]() { zmt_xxh32_kernel(base, off, len, n, out, nullptr, nullptr, nullptr); });
}

void emu_lz4_compress_batch(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len)
{
	u32 nrec = n ? (u32)((n + chunk - 1) / chunk) : 1;
	std::vector<u64> off(nrec);
	std::vector<u32> len(nrec), chk(nrec);
	for (u32 i = 0; i < nrec; i++) {
		off[i] = (u64)i * chunk;
		len[i] = (u32)std::min<u64>(chunk, n - off[i]);
	}
	emu_xxh32_batch(in, off.data(), len.data(), nrec, chk.data());
	const u32 *chkp = chk.data();
	/* ZMT_EMU_LZ4_ENC=3 selects the probe-batch encoder (lz4_enc3.hip), anything else the window encoder (lz4_enc5.hip) */
	const char *ev = getenv("ZMT_EMU_LZ4_ENC");
	const bool v3 = ev && atoi(ev) == 3;
	typedef void (*enc_fn)(const u8 *, u64, u32, u32, u32, u8 *, u64, u32 *, const u32 *, unsigned long long *);
	const enc_fn k16 = v3 ? zmt_lz4_enc3_u16_kernel : zmt_lz4_enc5_u16_kernel;
	const enc_fn k17 = v3 ? zmt_lz4_enc3_p17_kernel : zmt_lz4_enc5_p17_kernel;
	const enc_fn k32 = v3 ? zmt_lz4_enc3_u32_kernel : zmt_lz4_enc5_u32_kernel;
	if (chunk <= 65536) {
		emu::launch(dim3{nrec, 1, 1}, dim3{64, 1, 1},
			    [=]() { k16(in, n, chunk, 0, nrec, slots, stride, rec_len, chkp, nullptr); });
	} else {
		emu::launch(dim3{nrec, 1, 1}, dim3{64, 1, 1},
			    [=]() { (chunk <= 131072 ? k17 : k32)(in, n, chunk, 0, nrec, slots, stride, rec_len, chkp, nullptr); });
		emu::launch(dim3{1, 1, 1}, dim3{64, 1, 1},
			    [=]() { k16(in, n, chunk, nrec - 1, nrec, slots, stride, rec_len, chkp, nullptr); });
	}
}

void emu_lz4hc_compress_batch(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, int level)
{
	u32 nrec = n ? (u32)((n + chunk - 1) / chunk) : 1;
	std::vector<u64> off(nrec);
	std::vector<u32> len(nrec), chk(nrec);
	for (u32 i = 0; i < nrec; i++) {
		off[i] = (u64)i * chunk;
		len[i] = (u32)std::min<u64>(chunk, n - off[i]);
	}
	emu_xxh32_batch(in, off.data(), len.data(), nrec, chk.data());
	const u32 *chkp = chk.data();
	/* fewer waves than records on purpose: the persistent loop and the table reset are exercised */
	const u32 grid = nrec > 2 ? (nrec + 1) / 2 : nrec;
	std::vector<u8> scratch((size_t)grid * 327936, 0xA5);
	u8 *sc = scratch.data();
	emu::launch(dim3{grid, 1, 1}, dim3{64, 1, 1},
		    [=]() { zmt_lz4hc_enc_kernel(in, n, chunk, nrec, slots, stride, rec_len, chkp, sc, level); });
}

void emu_lz4_compact(const u8 *slots, u64 stride, const u32 *rec_len, u32 nrec, u8 *stream, u64 *rec_off)
{
	emu::launch(dim3{1, 1, 1}, dim3{1024, 1, 1}, [=]() { zmt_scan_kernel(rec_len, nrec, rec_off); });
	emu::launch(dim3{nrec, 1, 1}, dim3{256, 1, 1},
		    [=]() { zmt_compact_kernel(slots, stride, rec_len, rec_off, nrec, stream); });
}

void emu_lz4_probe_sizes(const u8 *stream, const u64 *rec_off, const u32 *rec_len, u32 nrec,
			 u32 *out_len, u64 *out_off)
{
	emu::launch(dim3{(nrec + 255) / 256, 1, 1}, dim3{256, 1, 1},
		    [=]() { zmt_probe_kernel(stream, rec_off, rec_len, nrec, out_len); });
	emu::launch(dim3{1, 1, 1}, dim3{1024, 1, 1}, [=]() { zmt_scan_kernel(out_len, nrec, out_off); });
}

void emu_lz4_decompress_batch(int variant, const u8 *stream, u64 stream_bytes, const u64 *rec_off,
			      const u32 *rec_len, u32 nrec, u8 *out, u64 out_bytes, const u64 *out_off,
			      u32 *out_len, u32 *status)
{
	std::vector<u32> ce(nrec), cv(nrec);
	u32 *cep = ce.data(), *cvp = cv.data();
	/* variant: low 4 bits = pipeline (0 frames + parse4 + copy3, 1 frame-serial), bits 4.. = log2 of copy3's
	 * ring (0 = 12, the default of the product) */
	const int ring = ((variant >> 4) & 15) ? ((variant >> 4) & 15) : 12;
	variant &= 15;
	if (variant == 1) {
		emu::launch(dim3{nrec, 1, 1}, dim3{64, 1, 1}, [=]() {
			zmt_lz4_dec_serial(stream, rec_off, rec_len, nrec, out, out_off, out_len, status, cep, cvp, 0xFFFFFFFFu);
		});
	} else {
		size_t nblk_max = out_bytes / 65536 + nrec + 1;
		size_t ntok_max = stream_bytes / 3 + 128 * nblk_max + 256;
		/* scratch starts as garbage, like device memory */
		std::vector<u32> est(nrec, 0xA5A5A5A5u), rnb(nrec, 0xA5A5A5A5u), rfl(nrec, 0xA5A5A5A5u), bcs(nblk_max, 0x00A5A5A5u), bnt(nblk_max, 0xA5A5A5A5u), bol(nblk_max, 0xA5A5A5A5u);
		std::vector<u64> blk0(nrec + 1, 0xA5A5A5A5A5A5A5A5ull), bco(nblk_max, 0x00A5A5A5A5A5A5A5ull);
		std::vector<uint16_t> tok(ntok_max, 0xA5A5);
		u32 *estp = est.data(), *rnbp = rnb.data(), *rflp = rfl.data(), *bcsp = bcs.data(), *bntp = bnt.data(), *bolp = bol.data();
		u64 *blk0p = blk0.data(), *bcop = bco.data();
		uint16_t *tokp = tok.data();
		emu::launch(dim3{(nrec + 255) / 256, 1, 1}, dim3{256, 1, 1}, [=]() { zmt_dec_nblk_kernel(out_len, nrec, estp); });
		emu::launch(dim3{1, 1, 1}, dim3{1024, 1, 1}, [=]() { zmt_scan_kernel(estp, nrec, blk0p); });
		emu::launch(dim3{(nrec + 255) / 256, 1, 1}, dim3{256, 1, 1}, [=]() {
			zmt_dec_frames_kernel(stream, rec_off, rec_len, nrec, out_len, blk0p, bcop, bcsp, rnbp, rflp, status, cep, cvp);
		});
		emu::launch(dim3{(u32)((nblk_max + 63) / 64), 1, 1}, dim3{64, 1, 1}, [=]() {
			zmt_dec_parse4_kernel(stream, stream_bytes, bcop, bcsp, blk0p + nrec, tokp, bntp, bolp);
		});
		if (getenv("ZMT_EMU_DEBUG")) {
			for (size_t b = 0; b < blk0[nrec]; b++)
				fprintf(stderr, "blk %zu coff=%llu csize=%x ntok=%u olen=%u\n", b, (unsigned long long)bco[b], bcs[b], bnt[b], bol[b]);
			for (u32 r = 0; r < nrec; r++)
				fprintf(stderr, "rec %u status=%u nblk=%u\n", r, status[r], rnb[r]);
		}
		emu::launch(dim3{nrec, 1, 1}, dim3{64, 1, 1}, [=]() {
				if (ring == 12)
					zmt_dec_copy3_w4_kernel(stream, stream_bytes, 0, nrec, out, out_off, out_len, blk0p, bcop, bcsp, rnbp, rflp, tokp, bntp, bolp, status);
				else if (ring == 13)
					zmt_dec_copy3_w8_kernel(stream, stream_bytes, 0, nrec, out, out_off, out_len, blk0p, bcop, bcsp, rnbp, rflp, tokp, bntp, bolp, status);
				else
					zmt_dec_copy3_w16_kernel(stream, stream_bytes, 0, nrec, out, out_off, out_len, blk0p, bcop, bcsp, rnbp, rflp, tokp, bntp, bolp, status);
		});
		emu::launch(dim3{nrec, 1, 1}, dim3{64, 1, 1}, [=]() {
			zmt_lz4_dec_serial(stream, rec_off, rec_len, nrec, out, out_off, out_len, status, cep, cvp, 100u);
		});
	}
	emu::launch(dim3{(nrec * 4 + 255) / 256, 1, 1}, dim3{256, 1, 1},
		    [=]() { zmt_xxh32_kernel(out, out_off, out_len, nrec, nullptr, cep, cvp, status); });
}


/* zstd: probe (content sizes + status), then wave-per-record decode; literal scratch starts as garbage */
void emu_zstd_probe(const u8 *stream, const u64 *rec_off, const u32 *rec_len, u32 nrec, u32 *out_len, u32 *status)
{
	emu::launch(dim3{(nrec + 255) / 256, 1, 1}, dim3{256, 1, 1},
		    [=]() { zmt_zstd_probe_kernel(stream, rec_off, rec_len, nrec, out_len, status); });
}

static u32 g_zstd_marks;
u32 emu_zstd_last_marks() { return g_zstd_marks; }
void emu_zstd_decompress_batch(const u8 *stream, u64 stream_bytes, const u64 *rec_off, const u32 *rec_len,
			       u32 nrec, u8 *out, const u64 *out_off, u32 *out_len, u32 *status)
{
	std::vector<u8> lit((size_t)nrec * 327936u, 0xA5); /* GPUMT_ZSTD_DEC_SCRATCH */
	u8 *litp = lit.data();
	std::vector<u32> ce(nrec, 0xA5A5A5A5u), cv(nrec, 0xA5A5A5A5u);
	u32 *cep = ce.data(), *cvp = cv.data();
	/* the sequence pre-pass first (as gpumt_zstd_decompress_batch; EMU_ZSTD_SEQ=1: none), into a buffer as large as the
	 * output that starts as garbage */
	u64 total = 0;
	for (u32 r = 0; r < nrec; r++)
		total = out_off[r] + out_len[r] > total ? out_off[r] + out_len[r] : total;
	const char *e = getenv("EMU_ZSTD_SEQ");
	const bool seq_on = !(e && atoi(e) == 1);
	std::vector<u8> seqv(seq_on ? (size_t)total + 32 : 0, 0xA5);
	u8 *seqbuf = seq_on ? seqv.data() + 16 : nullptr;
	const u64 seqcap = total; /* the capacity argument of zstd_dec_seq.h */
	if (seq_on)
		emu::launch(dim3{nrec, 1, 1}, dim3{64, 1, 1}, [=]() {
			zmt_zstd_seq_kernel(stream, stream_bytes, rec_off, rec_len, nrec, out_off, out_len, status, seqbuf, seqcap);
		});
	/* (test hook: how many blocks the pre-pass marked, read back from the record headers of zstd_dec_seq.h) */
	g_zstd_marks = 0;
	if (seq_on)
		for (u32 r = 0; r < nrec; r++) {
			if (out_len[r] <= 131072u || status[r] != 0)
				continue;
			u32 nh = (out_len[r] >> 14) & ~1u;
			nh = nh < 64 ? 64 : nh > 8192 ? 8192 : nh;
			const u32 *hdr = (const u32 *)(seqbuf + ((out_off[r] + 7) & ~7ull));
			for (u32 i = 0; i < nh; i++)
				g_zstd_marks += hdr[i] != 0;
		}
	/* small-table variant first, then the general one for the records it handed over (status 101) */
	emu::launch(dim3{nrec, 1, 1}, dim3{64, 1, 1}, [=]() {
		zmt_zstd_dec_small_kernel(stream, stream_bytes, rec_off, rec_len, nrec, out, out_off, out_len, litp, status, cep, cvp, seqbuf, seqcap);
	});
	if (getenv("ZMT_EMU_DEBUG"))
		for (u32 r = 0; r < nrec; r++)
			fprintf(stderr, "zstd rec %u after small kernel: status %u\n", r, status[r]);
	emu::launch(dim3{nrec, 1, 1}, dim3{64, 1, 1}, [=]() {
		zmt_zstd_dec_kernel(stream, stream_bytes, rec_off, rec_len, nrec, out, out_off, out_len, litp, status, cep, cvp, 101u, seqbuf, seqcap);
	});
	emu::launch(dim3{(nrec * 4 + 255) / 256, 1, 1}, dim3{256, 1, 1},
		    [=]() { zmt_xxh64_verify_kernel(out, out_off, out_len, nrec, cep, cvp, status); });
}


/* brotli compress: block encoder on `grid` persistent waves (scratch starts as garbage), then assemble;
 * slot geometry is the zstd one */
void emu_brotli_compress_batch_level(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid, int level);
void emu_brotli_compress_batch(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid)
{
	emu_brotli_compress_batch_level(in, n, chunk, slots, stride, rec_len, grid, 1);
}
/* quality -> tier as gpumt_brotli_level_tier: 0-3, 4-8, 9-11 */
void emu_brotli_compress_batch_level(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid, int level)
{
	u32 nrec = n ? (u32)((n + chunk - 1) / chunk) : 1;
	u32 bpr = (chunk + 131071) / 131072;
	u32 nblk = nrec * bpr;
	if (grid > nblk)
		grid = nblk;
	std::vector<u32> blk_len(nblk, 0xA5A5A5A5u);
	std::vector<u8> seq((size_t)grid * (3 * 32768 * 4), 0xA5);
	u32 *bl = blk_len.data();
	u8 *sq = seq.data();
	emu::launch(dim3{grid, 1, 1}, dim3{64, 1, 1}, [=]() {
		if (level <= 3)
			zmt_brotli_enc_kernel(in, n, chunk, nblk, bpr, slots, stride, bl, sq);
		else if (level <= 8)
			zmt_brotli_enc_t2_kernel(in, n, chunk, nblk, bpr, slots, stride, bl, sq);
		else
			zmt_brotli_enc_t3_kernel(in, n, chunk, nblk, bpr, slots, stride, bl, sq);
	});
	emu::launch(dim3{nrec, 1, 1}, dim3{256, 1, 1},
		    [=]() { zmt_brotli_assemble_kernel(n, chunk, nrec, bpr, slots, stride, bl, rec_len); });
}

/* brotli: `grid` persistent waves, each with its own (garbage-initialised) scratch */
void emu_brotli_decompress_batch(const u8 *stream, const u64 *rec_off, const u32 *rec_len, u32 nrec, u8 *out,
				 const u64 *out_off, const u32 *out_cap, u32 *out_len, u32 *status,
				 const u8 *blob, u32 grid)
{
	if (grid > nrec)
		grid = nrec;
	std::vector<u8> scr((size_t)grid * 825856u, 0xA5);
	u8 *sp = scr.data();
	/* EMU_BROTLI_DEC=1: the general kernel alone (gpumt_set_variant("brotli_dec", 1)); default: dec4 first, the
	 * general kernel for the records it hands over (status 102) */
	const char *v = getenv("EMU_BROTLI_DEC");
	u32 want = 0xFFFFFFFFu;
	if (!(v && atoi(v) == 1)) {
		emu::launch(dim3{(nrec + 3) / 4, 1, 1}, dim3{64, 1, 1}, [=]() { /* B4_NG = 4 streams per wave */
			zmt_brotli_dec4_kernel(stream, rec_off, rec_len, nrec, out, out_off, out_cap, out_len, status);
		});
		if (getenv("ZMT_EMU_DEBUG"))
			for (u32 r = 0; r < nrec; r++)
				fprintf(stderr, "brotli dec4: rec %u status %u\n", r, status[r]);
		want = 102u;
	}
	emu::launch(dim3{grid, 1, 1}, dim3{64, 1, 1}, [=]() {
		zmt_brotli_dec_kernel(stream, rec_off, rec_len, nrec, out, out_off, out_cap, out_len, status, sp, blob, want);
	});
}

size_t emu_zstd_slot_stride(size_t chunk)
{
	size_t nb = chunk ? (chunk + 131071) / 131072 : 1;
	return (32 + nb * (131072 + 16) + 255) & ~(size_t)255;
}

/* zstd compress: block encoder on `grid` persistent waves (scratch starts as garbage), then assemble */
void emu_zstd_compress_batch_level(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid, int level);
void emu_zstd_compress_batch(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid)
{
	emu_zstd_compress_batch_level(in, n, chunk, slots, stride, rec_len, grid, 1);
}
/* level -> tier as gpumt_zstd_level_tier: 1-2, 3-9, 10-22 */
void emu_zstd_compress_batch_level(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid, int level)
{
	u32 nrec = n ? (u32)((n + chunk - 1) / chunk) : 1;
	u32 bpr = (chunk + 131071) / 131072;
	u32 nblk = nrec * bpr;
	if (grid > nblk)
		grid = nblk;
	std::vector<u32> blk_len(nblk, 0xA5A5A5A5u);
	std::vector<u8> seq((size_t)grid * (3 * 32768 * 4 + 16 * 20544 + 131072 + 64), 0xA5);
	u32 *bl = blk_len.data();
	u8 *sq = seq.data();
	emu::launch(dim3{grid, 1, 1}, dim3{64, 1, 1}, [=]() {
		if (level <= 2)
			zmt_zstd_enc_kernel(in, n, chunk, nblk, bpr, slots, stride, bl, sq);
		else if (level <= 9)
			zmt_zstd_enc_t2_kernel(in, n, chunk, nblk, bpr, slots, stride, bl, sq);
		else
			zmt_zstd_enc_t3_kernel(in, n, chunk, nblk, bpr, slots, stride, bl, sq);
	});
	emu::launch(dim3{nrec, 1, 1}, dim3{256, 1, 1},
		    [=]() { zmt_zstd_assemble_kernel(n, chunk, nrec, bpr, slots, stride, bl, rec_len); });
}

/* snappy: `grid` persistent waves over the records */
size_t emu_snappy_slot_stride(size_t chunk) { return (16 + 32 + chunk + chunk / 6 + 255) & ~(size_t)255; }

void emu_snappy_compress_batch(const u8 *in, u64 n, u32 chunk, u8 *slots, u64 stride, u32 *rec_len, u32 grid)
{
	u32 nrec = n ? (u32)((n + chunk - 1) / chunk) : 1;
	if (grid > nrec)
		grid = nrec;
	emu::launch(dim3{grid, 1, 1}, dim3{64, 1, 1},
		    [=]() { zmt_snappy_enc_kernel(in, n, chunk, nrec, slots, stride, rec_len); });
}

/* EMU_SNAPPY_DEC=1 in the environment selects the batched decoder (gpumt_set_variant("snappy_dec", 1)) */
void emu_snappy_decompress_batch(const u8 *stream, const u64 *rec_off, const u32 *rec_len, u32 nrec, u8 *out,
				 const u64 *out_off, const u32 *out_cap, u32 *out_len, u32 *status, u32 grid)
{
	if (grid > nrec)
		grid = nrec;
	const char *e = getenv("EMU_SNAPPY_DEC");
	if (e && *e == '1')
		emu::launch(dim3{grid, 1, 1}, dim3{64, 1, 1}, [=]() {
			zmt_snappy_dec2_kernel(stream, rec_off, rec_len, nrec, out, out_off, out_cap, out_len, status);
		});
	else
		emu::launch(dim3{grid, 1, 1}, dim3{64, 1, 1}, [=]() {
			zmt_snappy_dec_kernel(stream, rec_off, rec_len, nrec, out, out_off, out_cap, out_len, status);
		});
}

} /* extern "C" */
