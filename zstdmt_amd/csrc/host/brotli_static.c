/*
 * brotli_static.c -- embeds csrc/data/brotli_static.bin (the constant data RFC 7932 defines for
 * every brotli decoder: static dictionary, word transforms, context lookup; layout and provenance
 * in tools/gen_brotli_tables.py) into the library.  The device copy is made by gpumt.hip on the
 * first brotli call.
 */
__asm__(".section .rodata\n"
	".balign 64\n"
	".global zmt_brotli_static\n"
	"zmt_brotli_static:\n"
	".incbin \"brotli_static.bin\"\n"
	".global zmt_brotli_static_end\n"
	"zmt_brotli_static_end:\n"
	".byte 0\n"
	".previous\n");
