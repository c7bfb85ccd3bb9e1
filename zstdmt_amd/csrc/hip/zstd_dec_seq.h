/*
 * zstd_dec_seq.h -- what the sequence pre-pass (zstd_dec_seq.hip) and the frame decoder (zstd_dec.hip) agree on: where a
 * record's region of the sequence buffer lies and how it is laid out.
 *
 * `seqbuf` is as large as the batch's output (+ 8).  Record r owns [align8(out_off[r]), out_off[r] + out_len[r]) of it:
 *   u32 hdr[nhdr]     hdr[i] = 0: block i of the frame (counting every block) is the decoder's;
 *                     else 1 + index of the block's first sequence in seq[]
 *   u64 seq[]         ll | ml << 18 | offset value << 36, in stream order
 * Records that cannot hold more than one block have no region (neither kernel touches the buffer for them), and
 * neither have records whose region would end behind the buffer: its capacity in bytes, `seqcap`, is an argument of
 * both kernels (a caller whose out_bytes understates the real extent loses the pre-pass for the records beyond it,
 * nothing else).
 */
#pragma once

#ifndef ZS_NB
#define ZS_NB 8u /* blocks of a frame decoded side by side, four lanes each */
#endif

static __device__ __forceinline__ bool zs_eligible(u64 seqcap, u64 out_off, u32 out_len)
{
	return out_len > 131072u && out_off + out_len <= seqcap;
}
static __device__ __forceinline__ u32 zs_nhdr(u32 out_len)
{
	u32 n = (out_len >> 14) & ~1u;
	n = n < 64u ? 64u : n;
	return n > 8192u ? 8192u : n;
}
static __device__ __forceinline__ u64 *zs_region(u8 *seqbuf, u64 out_off) { return (u64 *)(seqbuf + ((out_off + 7) & ~7ull)); }
static __device__ __forceinline__ u32 zs_seqcap(u64 out_off, u32 out_len)
{
	const u64 words = (out_off + out_len - ((out_off + 7) & ~7ull)) >> 3;
	const u32 h = zs_nhdr(out_len) / 2;
	return words > h ? (u32)(words - h) : 0u;
}
