"""Pin the oracle (CPU restatement) to the reference: committed golden vectors produced by the
reference's own lz4-mt sources (tests/golden/gen_golden.py) and SURVEY.md Appendix C answers."""
import json
import os

import pytest

import helpers as H
from cases import CASES, KNOWN_HEX

with open(os.path.join(H.GOLDEN_DIR, "manifest.json")) as _f:
    MAN = json.load(_f)["cases"]


def golden_stream(name):
    e = MAN[name]
    if "out_hex" in e:
        return bytes.fromhex(e["out_hex"])
    if "out_file" in e:
        with open(os.path.join(H.GOLDEN_DIR, e["out_file"]), "rb") as f:
            return f.read()
    return None


def test_case_list_matches_manifest():
    assert set(CASES) == set(MAN)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_compress_matches_reference(name):
    chunk, thunk = CASES[name]
    data = thunk()
    e = MAN[name]
    assert len(data) == e["in_len"] and H.sha256(data) == e["in_sha256"], "input generator drifted"
    s = H.oracle_compress(data, chunk)
    assert len(s) == e["out_len"]
    assert H.sha256(s) == e["out_sha256"]
    g = golden_stream(name)
    if g is not None:
        assert s == g
    # stats the reference reported (SURVEY 8b "Stats")
    assert e["outsize"] == len(s) and e["insize"] == len(data)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_decompress_golden(name):
    chunk, thunk = CASES[name]
    data = thunk()
    g = golden_stream(name)
    if g is None:
        g = H.oracle_compress(data, chunk)
        assert H.sha256(g) == MAN[name]["out_sha256"]
    back = H.oracle_decompress(g, max(len(data), 65536))
    assert back == data


@pytest.mark.parametrize("name", sorted(KNOWN_HEX))
def test_survey_known_answers(name):
    chunk, thunk = CASES[name]
    assert H.oracle_compress(thunk(), chunk).hex() == KNOWN_HEX[name]


def test_frame_bound():
    lib = H.oracle()
    assert lib.zo_lz4f_bound(131072) == 131107      # SURVEY section 8 config arithmetic
    assert lib.zo_lz4f_bound(4 << 20) == 4194587
    assert lib.zo_lz4f_bound(0) == 27


def test_xxh32_vectors():
    lib = H.oracle()
    # published XXH32 test vectors (xxHash spec) + cross-check with the python xxhash module
    assert lib.zo_xxh32(b"", 0, 0) == 0x02CC5D05
    import xxhash
    for n in (1, 3, 4, 15, 16, 17, 31, 32, 33, 100, 4096, 65537):
        d = H.lcg(n, n)
        assert lib.zo_xxh32(d, n, 0) == xxhash.xxh32(d, seed=0).intdigest()
        assert lib.zo_xxh32(d, n, 77) == xxhash.xxh32(d, seed=77).intdigest()


@pytest.mark.parametrize("mutate", ["magic", "hc", "blocksize", "checksum", "truncate", "trailing",
                                    "skipmagic", "skiplen", "offset0"])
def test_oracle_rejects_corrupt(mutate):
    data = CASES["text_128k"][1]()
    s = bytearray(H.oracle_compress(data, 128 * 1024))
    if mutate == "magic":
        s[12] ^= 1
    elif mutate == "hc":
        s[12 + 14] ^= 0x10
    elif mutate == "blocksize":
        s[12 + 15 + 2] ^= 0x40
    elif mutate == "checksum":
        s[-1] ^= 0x80
    elif mutate == "truncate":
        s = s[:-5]
    elif mutate == "trailing":
        s[8] += 1
        s += b"\0"
    elif mutate == "skipmagic":
        s[0] ^= 1
    elif mutate == "skiplen":
        s[4] = 8
    elif mutate == "offset0":
        # first sequence's offset -> 0 (invalid): token at block start
        p = 12 + 15 + 4
        lit = s[p] >> 4
        q = p + 1
        if lit == 15:
            while s[q] == 255:
                lit += 255
                q += 1
            lit += s[q]
            q += 1
        s[q + lit] = 0
        s[q + lit + 1] = 0
    assert H.oracle_decompress(bytes(s), len(data)) is None


def _flag_cases():
    import json
    d = os.path.join(H.GOLDEN_DIR, "lz4f_flags")
    man = json.load(open(os.path.join(d, "manifest.json")))["cases"]
    return d, man


def test_oracle_decodes_block_checksum_and_dictid_frames():
    """LZ4F features lz4-mt never writes but liblz4 accepts (fixtures written by liblz4 1.9.3,
    tests/golden/gen_golden_lz4f_flags.py): the oracle decodes them and rejects a damaged block."""
    d, man = _flag_cases()
    for name, e in man.items():
        rec = open(os.path.join(d, name + ".rec"), "rb").read()
        out = H.oracle_decompress(rec, e["content_len"] + 64)
        assert out is not None and len(out) == e["content_len"] and H.sha256(out) == e["content_sha256"], name
        if e["flg"] & 0x10:
            bad = bytearray(rec)
            bad[12 + 40] ^= 0x01
            assert H.oracle_decompress(bytes(bad), e["content_len"] + 64) is None, name


with open(os.path.join(H.GOLDEN_DIR, "lz4hc", "manifest.json")) as _f:
    HCMAN = json.load(_f)["cases"]


@pytest.mark.parametrize("level", [3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("name", sorted(HCMAN))
def test_oracle_hc_matches_reference_golden(name, level):
    """LZ4 HC levels 3..9 (lz4hc_oracle.c) against digests of what the reference wrote
    (tests/golden/gen_golden_lz4hc.py)."""
    chunk, thunk = CASES[name]
    data = thunk()
    e = HCMAN[name]["levels"][str(level)]
    assert H.sha256(data) == HCMAN[name]["in_sha256"]
    s = H.oracle_compress_level(data, chunk, level)
    assert len(s) == e["out_len"] and H.sha256(s) == e["out_sha256"]
    if "out_hex" in e:
        assert s.hex() == e["out_hex"]
    assert H.oracle_decompress(s, max(len(data), 65536)) == data
