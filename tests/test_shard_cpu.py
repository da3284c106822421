"""N>1 path on CPU: two ranks (gloo) each genotype their shard of a batch, rank 0 gathers the posterior summaries.

The sampler on each rank is the ORACLE here (no GPU in this container; tests/ may use it): what is under test is the host
logic the multi-GPU path adds — the assignment, the sub-batch extraction that keeps global group indices (=> identical seeds),
the variable-size gather and the scatter back into cluster order — checked against the unsharded run, which must be identical.
"""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from bayestyper_amd import shard, synth  # noqa: E402


def _batch(S=2):
    parts = [synth.make_batch("A", 9, S, seed=5, templates=3), synth.make_batch("B", 4, S, seed=6, templates=2), synth.make_batch("C", 3, S, seed=7)]
    flat = synth.concat(parts)
    flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
    return flat


def _run_oracle(flat, S):
    import _oracle

    orc = _oracle.load_oracle()
    lut_g, lut_n = _oracle.build_luts(orc, S)
    og = _oracle.OrcGibbs(orc, flat, lut_g, lut_n, seed=42, chains=2, burn=10, iters=30)
    og.run(1)
    res = og.results()
    og.close()
    return shard.summary_from_results(res, flat["num_clusters"], S)


def _worker(rank, world, port, out_path):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    S = 2
    flat = _batch(S)
    parts = shard.assign_groups(shard.group_cost(flat), world)
    mine = shard.take_groups(flat, parts[rank])
    local = _run_oracle(mine, S)
    full = shard.gather_summaries(local, shard.cluster_ids_of(flat, parts[rank]), flat["num_clusters"], rank, world)
    # the noise drivers' only exchange: sum of the per-rank noise-count histograms
    import torch

    hist = torch.full((S, 256), rank + 1, dtype=torch.int64)
    shard.allreduce_noise_counts(hist)
    assert int(hist[0, 0]) == sum(range(1, world + 1))
    if rank == 0:
        np.save(out_path, full)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_assignment_is_a_balanced_partition():
    flat = _batch()
    cost = shard.group_cost(flat)
    for world in (1, 2, 3, 8):
        parts = shard.assign_groups(cost, world)
        allg = np.sort(np.concatenate(parts))
        assert np.array_equal(allg, np.arange(flat["num_groups"]))
        loads = np.array([cost[p].sum() for p in parts])
        assert loads.max() - loads.min() <= cost.max() + 1e-9   # LPT serpentine bound


def test_take_groups_roundtrip():
    flat = _batch()
    same = shard.take_groups(flat, np.arange(flat["num_groups"]))
    for k, v in flat.items():
        if isinstance(v, np.ndarray) and k in same:
            assert np.array_equal(np.asarray(same[k]).reshape(-1), v.reshape(-1)), k
    # empty shard and a permuted shard are well-formed
    assert shard.take_groups(flat, [])["num_groups"] == 0
    sub = shard.take_groups(flat, [15, 0, 12])
    assert sub["num_groups"] == 3 and list(sub["group_index"]) == [15, 0, 12]
    assert sub["kmer_off"][-1] * sub["S"] == len(sub["kmer_counts"])


def test_two_rank_gloo_gather_matches_unsharded():
    import torch.multiprocessing as mp

    S = 2
    flat = _batch(S)
    ref = _run_oracle(flat, S)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "gathered.npy")
        mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
        got = np.load(out)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)
    assert (ref[:, :, 1] > 0).any()


# ---- the host layer's rank exchange (bthost::Comm) over its files transport: three processes, no GPU --------------------------------
def _comm_rank(rank, world, id_file, q):
    os.environ.update({"BT_WORLD": str(world), "BT_RANK": str(rank), "BT_COMM_ID_FILE": id_file, "BT_COMM_TRANSPORT": "files"})
    import ctypes as C

    from bayestyper_amd.host import dll

    err = C.create_string_buffer(512)
    dll.bth_comm_selftest.argtypes = [C.c_uint, C.c_char_p, C.c_uint]
    rc = dll.bth_comm_selftest(25, err, len(err))
    q.put((rank, rc, err.value.decode()))


def test_host_comm_files_transport_three_ranks(tmp_path):
    """bthost::Comm (the exchange steps `bayesTyper genotype` uses between its ranks) over the files transport: all-reduce, all-gather of byte
    strings of different lengths, gather of words to rank 0 with an empty contribution, 25 rounds, three processes"""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    id_file = str(tmp_path / "run.comm_id")
    procs = [ctx.Process(target=_comm_rank, args=(r, 3, id_file, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(30)
    assert got == [(0, 0, ""), (1, 0, ""), (2, 0, "")], got
    assert not os.path.exists(id_file + ".d") or not os.listdir(id_file + ".d")   # the exchange files are gone
