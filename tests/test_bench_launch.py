"""bench.py --gpus N starts N ranks by itself (one process per GPU) and shards ONE job across them
(strong scaling, contiguous chunk ranges): checked here with --dry-run, which joins the ranks over
gloo and touches no GPU."""
import json
import os
import subprocess
import sys

import helpers as H


def _run(*extra):
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    out = subprocess.check_output([sys.executable, os.path.join(H.ROOT, "bench.py"), "--dry-run", *extra],
                                  env=env, timeout=300, stderr=subprocess.DEVNULL)
    return json.loads(out.decode().strip().splitlines()[-1])


def test_two_ranks_distinct_devices_and_tiled_chunks():
    r = _run("--gpus", "2")
    assert r["dry_run"] and r["n_gpus"] == 2 and r["scaling"] == "strong"
    ranks = sorted(r["ranks"], key=lambda x: x["rank"])
    assert [x["rank"] for x in ranks] == [0, 1]
    assert [x["local_rank"] for x in ranks] == [0, 1] and [x["device"] for x in ranks] == [0, 1]
    assert len({x["pid"] for x in ranks}) == 2
    # one 8 GiB job of 65 536 chunks, split in two contiguous halves
    assert ranks[0]["chunks"] == [0, 32768] and ranks[1]["chunks"] == [32768, 65536]
    # the multi-GPU line's extra keys: reassembly mode / time and one entry of kernel times per rank
    assert r["gather"] == "none" and "gather_ms" in r
    assert [x["rank"] for x in r["per_rank_ms"]] == [0, 1]
    assert set(r["per_rank_ms"][0]) == {"rank", "k_enc", "k_dec", "compress_leg", "decompress_leg"}


def test_gather_modes_are_accepted():
    for mode in ("rccl", "d2h"):
        r = _run("--gpus", "2", "--gather", mode)
        assert r["gather"] == mode and r["n_gpus"] == 2
    assert _run("--gpus", "2", "--gather")["gather"] == "rccl"   # bare flag = round 2's behaviour


def test_three_ranks_weak_and_decompress_mode():
    r = _run("--gpus", "3", "--scaling", "weak", "--mode", "decompress", "--gib", "1")
    assert r["n_gpus"] == 3 and r["scaling"] == "weak" and r["mode"] == "decompress"
    assert all(x["chunks"] == [0, 8192] for x in r["ranks"])


def test_single_process_default():
    r = _run()
    assert r["n_gpus"] == 1 and r["ranks"][0]["chunks"] == [0, 65536]
