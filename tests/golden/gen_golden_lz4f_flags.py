"""Generates tests/golden/lz4f_flags/*: LZ4 frames with the two LZ4F features lz4-mt never writes but
liblz4 (the decoder behind lib/lz4-mt_decompress.c:349-362) accepts -- block checksums and a
dictionary id -- wrapped as lz4-mt records (12-byte skippable header + frame).  Written with the
image's liblz4 1.9.3 (LZ4F_compressFrame); run in the build container:
    python tests/golden/gen_golden_lz4f_flags.py
Each case: <name>.rec (the record), and manifest.json with the SHA-256 of the content."""
import ctypes as C
import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from cases import text  # noqa: E402

L = C.CDLL("/opt/conda/lib/liblz4.so.1")


class FrameInfo(C.Structure):
    _fields_ = [("blockSizeID", C.c_uint), ("blockMode", C.c_uint), ("contentChecksumFlag", C.c_uint),
                ("frameType", C.c_uint), ("contentSize", C.c_ulonglong), ("dictID", C.c_uint),
                ("blockChecksumFlag", C.c_uint)]


class Prefs(C.Structure):
    _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint),
                ("favorDecSpeed", C.c_uint), ("reserved", C.c_uint * 3)]


L.LZ4F_compressFrameBound.restype = C.c_size_t
L.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.POINTER(Prefs)]
L.LZ4F_compressFrame.restype = C.c_size_t
L.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(Prefs)]
L.LZ4F_isError.argtypes = [C.c_size_t]


def frame(data, block_checksum=0, dict_id=0, content_checksum=1, linked=True, content_size=True, bsid=4):
    p = Prefs()
    p.frameInfo.blockSizeID = bsid
    p.frameInfo.blockMode = 0 if linked else 1
    p.frameInfo.contentChecksumFlag = content_checksum
    p.frameInfo.contentSize = len(data) if content_size else 0
    p.frameInfo.dictID = dict_id
    p.frameInfo.blockChecksumFlag = block_checksum
    p.compressionLevel = 1
    cap = L.LZ4F_compressFrameBound(len(data), C.byref(p))
    out = C.create_string_buffer(cap)
    n = L.LZ4F_compressFrame(out, cap, data, len(data), C.byref(p))
    assert not L.LZ4F_isError(n)
    return out.raw[:n]


def record(fr):
    return struct.pack("<III", 0x184D2A50, 4, len(fr)) + fr


CASES = {
    "bcheck_text_300k": dict(data=text(300000, 11), block_checksum=1),
    "bcheck_indep_text_200k": dict(data=text(200000, 12), block_checksum=1, linked=False),
    "dictid_text_150k": dict(data=text(150000, 13), dict_id=0x12345678),
    "bcheck_dictid_nocc_text_70k": dict(data=text(70000, 14), block_checksum=1, dict_id=7, content_checksum=0),
    "bcheck_random_100k": dict(data=bytes((i * 2654435761 >> 13) & 255 for i in range(100000)), block_checksum=1),
}

if __name__ == "__main__":
    out_dir = os.path.join(HERE, "lz4f_flags")
    os.makedirs(out_dir, exist_ok=True)
    man = {}
    for name, kw in CASES.items():
        data = kw.pop("data")
        rec = record(frame(data, **kw))
        with open(os.path.join(out_dir, name + ".rec"), "wb") as f:
            f.write(rec)
        man[name] = {"content_len": len(data), "content_sha256": hashlib.sha256(data).hexdigest(),
                     "record_len": len(rec), "flg": rec[12 + 4], "prefs": kw}
    with open(os.path.join(out_dir, "manifest.json"), "w") as f:
        json.dump({"generator": "gen_golden_lz4f_flags.py, liblz4 1.9.3", "cases": man}, f, indent=1)
    print({k: (v["record_len"], hex(v["flg"])) for k, v in man.items()})
