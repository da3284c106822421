"""bench.py's job sizing for --gpus N (no GPU): weak scaling = one unit of N launch-sized blocks with unit-wide group indices; strong scaling = one batch
dealt to the ranks; the word string a rank's results travel in."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_weak_scaling_is_one_unit_of_launch_sized_blocks():
    import bench

    world, groups, S = 3, 300, 2
    parts = [bench.rank_batch(groups, S, r, world, "weak") for r in range(world)]
    G = parts[0][0]["num_groups"]
    assert all(p[0]["num_groups"] == G for p in parts), "every rank keeps a launch-sized block: per-GPU work does not shrink with N"
    assert all(p[1] == sum(q[0]["num_clusters"] for q in parts) or p[1] == p[0]["num_clusters"] * world for p in parts)
    idx = np.concatenate([p[0]["group_index"] for p in parts])
    assert len(np.unique(idx)) == world * G and idx.min() == 0 and idx.max() == world * G - 1, "unit-wide group indices: a group's seeds do not depend on its rank"
    assert not np.array_equal(parts[0][0]["kmer_counts"][:1000], parts[1][0]["kmer_counts"][:1000]), "the blocks are different groups"
    one = bench.rank_batch(groups, S, 0, 1, "weak")
    assert one[0]["num_groups"] == G and np.array_equal(one[0]["group_index"], parts[0][0]["group_index"]), "rank 0's block is the N = 1 batch"


def test_strong_scaling_deals_one_batch():
    import bench

    world, groups, S = 2, 300, 2
    parts = [bench.rank_batch(groups, S, r, world, "strong") for r in range(world)]
    unit = parts[0][2]
    assert sum(p[0]["num_groups"] for p in parts) == unit["num_groups"] and all(p[1] == unit["num_clusters"] for p in parts)
    ids = np.sort(np.concatenate([p[3] for p in parts]))
    assert np.array_equal(ids, np.arange(unit["num_groups"]))
    for p in parts:
        assert np.array_equal(p[0]["group_index"], unit["group_index"][p[3]])


def test_result_words_round_trip():
    """the wire string of a launch's results (include/btgpu.h: bt_gibbs_result_words; host/Comm.cpp builds the same from host arrays): header,
    sizes per cluster, keys, counts, pad to an even word count, statistics"""
    from bayestyper_amd import lib

    C, S = 3, 2
    res = {"dip_off": np.asarray([0, 2, 3, 6], np.uint64), "h1": np.asarray([0, 0, 1, 0, 1, 65535], np.uint16), "h2": np.asarray([0, 1, 1, 65535, 65535, 65535], np.uint16),
           "freq": np.arange(12, dtype=np.uint32).reshape(6, S), "cell_off": np.asarray([0, 4, 8, 12], np.uint64), "stats": np.linspace(0, 1, 12 * 12).reshape(12, 3, 4)}
    sizes = np.stack([np.diff(res["dip_off"]), np.diff(res["cell_off"])], axis=1).astype(np.uint32).reshape(-1)
    keys = res["h1"].astype(np.uint32) | (res["h2"].astype(np.uint32) << 16)
    head = np.concatenate([np.asarray([C, 6, 12, S], np.uint32), sizes, keys, res["freq"].reshape(-1)])
    assert len(head) % 2 == 0
    for pad in (0, 1):   # (an odd count before the statistics gets one pad word: here by way of one more entry)
        if pad:
            r2 = dict(res, dip_off=np.asarray([0, 2, 3, 7], np.uint64), h1=np.append(res["h1"], 7).astype(np.uint16), h2=np.append(res["h2"], 9).astype(np.uint16),
                      freq=np.arange(14, dtype=np.uint32).reshape(7, S))
            sizes2 = np.stack([np.diff(r2["dip_off"]), np.diff(r2["cell_off"])], axis=1).astype(np.uint32).reshape(-1)
            k2 = r2["h1"].astype(np.uint32) | (r2["h2"].astype(np.uint32) << 16)
            h2 = np.concatenate([np.asarray([C, 7, 12, S], np.uint32), sizes2, k2, r2["freq"].reshape(-1)])
            assert len(h2) % 2 == 1
            w, want = np.concatenate([h2, np.zeros(1, np.uint32), res["stats"].reshape(-1).view(np.uint32)]), r2
        else:
            w, want = np.concatenate([head, res["stats"].reshape(-1).view(np.uint32)]), res
        got, used = lib.parse_result_words(np.concatenate([w, w]))   # (a rank's string: its launches' strings one after the other)
        assert used == len(w)
        for key in want:
            assert np.array_equal(got[key], want[key]), key


def test_profile_summaries_are_matched_per_part():
    """a profiles/ summary carries one hash per part (Gibbs launch / KMC scan): bench.py uses it for roofline.traffic / issue_frac only when that
    part's sources are the ones the library was built from; older summaries (one hash over everything) still match their exact tree"""
    import bench

    hg, hk, hall = bench.source_hash("gibbs"), bench.source_hash("kmc"), bench.source_hash()
    assert len({hg, hk, hall}) == 3
    assert bench._matches({"source_hash_gibbs": hg, "source_hash_kmc": "0" * 16, "source_hash": "0" * 16}, "gibbs")
    assert not bench._matches({"source_hash_gibbs": hg, "source_hash_kmc": "0" * 16, "source_hash": hall}, "kmc")
    assert bench._matches({"source_hash": hall}, "kmc") and not bench._matches({"source_hash": "0" * 16}, "gibbs")
