#!/usr/bin/env python3
"""Writes zstdmt_amd/csrc/data/brotli_static.bin: the constant data RFC 7932 defines for every
brotli decoder -- the static dictionary (Appendix A, 122 784 bytes, CRC-32 0x5136cb04), the
121 word transforms (Appendix B) and the literal context lookup table (section 7.1).

The RFC text is not available offline, so the bytes are read out of the brotli 1.0.9 runtime the
image ships (/opt/conda/lib/libbrotlicommon.so.1: BrotliGetDictionary, BrotliGetTransforms,
_kBrotliContextLookupTable) and checked against the CRC the RFC states.  This is data of the
format, not code; the committed blob is what the product and the oracle both load.

Blob layout (little endian):
  +0      "BRST" u32 version=1
  +8      u32 off_dict, u32 off_ctx, u32 off_tr, u32 off_psmap, u32 off_ps, u32 ps_size, u32 ntr, u32 npsmap
  off_dict  122784 B   dictionary words, by length 4..24
  off_ctx   2048 B     context lookup: 4 modes x (256 for p1, 256 for p2)
  off_tr    121 x 3 B  (prefix id, transform type, suffix id)
  off_psmap n x u16    offset of affix id inside the affix pool
  off_ps    ps_size B  affix pool: length byte + characters, per affix
"""
import ctypes as C
import os
import struct
import sys
import zlib

LIB = "/opt/conda/lib/libbrotlicommon.so.1"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zstdmt_amd", "csrc", "data",
                   "brotli_static.bin")


class Dict(C.Structure):
    _fields_ = [("size_bits_by_length", C.c_uint8 * 32), ("offsets_by_length", C.c_uint32 * 32),
                ("data_size", C.c_size_t), ("data", C.POINTER(C.c_uint8))]


class Transforms(C.Structure):
    _fields_ = [("prefix_suffix_size", C.c_uint16), ("prefix_suffix", C.POINTER(C.c_uint8)),
                ("prefix_suffix_map", C.POINTER(C.c_uint16)), ("num_transforms", C.c_uint32),
                ("transforms", C.POINTER(C.c_uint8)), ("params", C.POINTER(C.c_uint8)),
                ("cutOffTransforms", C.c_int16 * 10)]


def main():
    lib = C.CDLL(LIB)
    lib.BrotliGetDictionary.restype = C.POINTER(Dict)
    lib.BrotliGetTransforms.restype = C.POINTER(Transforms)
    d = lib.BrotliGetDictionary().contents
    t = lib.BrotliGetTransforms().contents
    assert d.data_size == 122784, d.data_size
    dict_bytes = bytes(d.data[:d.data_size])
    assert zlib.crc32(dict_bytes) == 0x5136CB04, hex(zlib.crc32(dict_bytes))
    bits = list(d.size_bits_by_length)
    offs = list(d.offsets_by_length)
    assert bits[4:25] == [10, 10, 11, 11, 10, 10, 10, 10, 10, 9, 9, 8, 7, 7, 8, 7, 7, 6, 6, 5, 5], bits
    o = 0
    for ln in range(4, 25):
        assert offs[ln] == o
        o += ln << bits[ln]
    assert o == 122784
    ntr = t.num_transforms
    assert ntr == 121
    tr = bytes(t.transforms[:3 * ntr])
    ps = bytes(t.prefix_suffix[:t.prefix_suffix_size])
    nmap = max(max(tr[0::3]), max(tr[2::3])) + 1
    psmap = list(t.prefix_suffix_map[:nmap])
    assert all(x < 21 for x in tr[1::3]), "no shift transforms in the RFC set"
    ctx = bytes((C.c_uint8 * 2048).in_dll(lib, "_kBrotliContextLookupTable"))
    hdr_size = 8 + 8 * 4
    off_dict = 64
    off_ctx = off_dict + len(dict_bytes)
    off_tr = off_ctx + 2048
    off_psmap = (off_tr + len(tr) + 1) & ~1
    off_ps = off_psmap + 2 * nmap
    blob = bytearray(off_ps + len(ps))
    blob[0:8] = b"BRST" + struct.pack("<I", 1)
    blob[8:8 + 32] = struct.pack("<8I", off_dict, off_ctx, off_tr, off_psmap, off_ps, len(ps), ntr, nmap)
    assert hdr_size <= off_dict
    blob[off_dict:off_ctx] = dict_bytes
    blob[off_ctx:off_tr] = ctx
    blob[off_tr:off_tr + len(tr)] = tr
    blob[off_psmap:off_ps] = struct.pack(f"<{nmap}H", *psmap)
    blob[off_ps:] = ps
    with open(OUT, "wb") as f:
        f.write(blob)
    print(f"{OUT}: {len(blob)} bytes, {ntr} transforms, {nmap} affixes, affix pool {len(ps)} B, "
          f"crc32(blob)={zlib.crc32(bytes(blob)):08x}")
    if "-v" in sys.argv:
        for i in range(ntr):
            p, ty, s = tr[3 * i:3 * i + 3]
            ps_ = ps[psmap[p] + 1:psmap[p] + 1 + ps[psmap[p]]]
            ss_ = ps[psmap[s] + 1:psmap[s] + 1 + ps[psmap[s]]]
            print(i, repr(ps_), ty, repr(ss_))


if __name__ == "__main__":
    main()
