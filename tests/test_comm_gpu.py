"""libbtcomm.so (include/btcomm.h) on the GPU: the RCCL exchange steps.  The box of the GPU tests has one GPU, so the collectives are
exercised with a one-rank communicator (every call must then behave as its definition says for world_size = 1); with two or more GPUs
visible a two-process run checks the real exchange (all-reduce sums, gather in rank order, all-to-all routing)."""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _comm_lib():
    from bayestyper_amd import lib

    dll = C.CDLL(os.path.join(ROOT, "bayestyper_amd", "libbtcomm.so"))
    vp = C.c_void_p
    dll.bt_comm_unique_id.argtypes = [vp]
    dll.bt_comm_init.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    dll.bt_comm_destroy.argtypes = [vp]
    dll.bt_comm_rank.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    dll.bt_comm_allreduce_hist.argtypes = [vp, vp, C.c_uint64]
    dll.bt_comm_gather_summaries.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp]
    dll.bt_comm_alltoallv_matches.argtypes = [vp, vp, vp, vp, C.c_uint64, vp]
    dll.bt_comm_allgatherv.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp]
    return lib, dll


def _rank_body(rank, world, id_bytes, queue):
    lib, dll = _comm_lib()
    ctx = lib.Ctx(rank)
    h = C.c_void_p()
    ident = (C.c_uint8 * 128)(*id_bytes)
    lib.check(dll.bt_comm_init(ctx.h, ident, rank, world, C.byref(h)))
    r, w = C.c_int(), C.c_int()
    lib.check(dll.bt_comm_rank(h, C.byref(r), C.byref(w)))
    assert (r.value, w.value) == (rank, world)
    # all-reduce: rank r contributes (r + 1) * arange
    hist = (np.arange(3 * 256, dtype=np.uint64) * np.uint64(rank + 1))
    d = ctx.to_device(hist)
    lib.check(dll.bt_comm_allreduce_hist(h, d.ptr, len(hist)))
    ctx.sync()
    got = d.download(np.uint64, len(hist))
    assert np.array_equal(got, np.arange(3 * 256, dtype=np.uint64) * np.uint64(world * (world + 1) // 2))
    # gather: rank r sends 10 + 3r words of value r
    local = np.full(10 + 3 * rank, rank, np.uint32)
    dl = ctx.to_device(local)
    total = sum(10 + 3 * q for q in range(world))
    out = ctx.buffer(4 * total)
    offs = np.zeros(world + 1, np.uint64)
    lib.check(dll.bt_comm_gather_summaries(h, dl.ptr, len(local), out.ptr, total, offs.ctypes.data))
    ctx.sync()
    assert list(offs) == [sum(10 + 3 * q for q in range(i)) for i in range(world + 1)]
    if rank == 0:
        g = out.download(np.uint32, total)
        assert np.array_equal(g, np.concatenate([np.full(10 + 3 * q, q, np.uint32) for q in range(world)]))
    # all-to-all: rank r sends (r + q + 1) * 18 bytes of value 16 * r + q to rank q
    send_sizes = np.array([(rank + q + 1) * 18 for q in range(world)], np.uint64)
    send = np.concatenate([np.full(int(send_sizes[q]), 16 * rank + q, np.uint8) for q in range(world)])
    ds = ctx.to_device(send)
    recv_total = sum((q + rank + 1) * 18 for q in range(world))
    dr = ctx.buffer(recv_total)
    recv_sizes = np.zeros(world, np.uint64)
    lib.check(dll.bt_comm_alltoallv_matches(h, ds.ptr, send_sizes.ctypes.data, dr.ptr, recv_total, recv_sizes.ctypes.data))
    ctx.sync()
    assert list(recv_sizes) == [(q + rank + 1) * 18 for q in range(world)]
    assert np.array_equal(dr.download(np.uint8, recv_total), np.concatenate([np.full((q + rank + 1) * 18, 16 * q + rank, np.uint8) for q in range(world)]))
    # too small a receive buffer is an error, not a truncation — and an error on EVERY rank, also when only ONE rank's buffer is short
    # (the capacities travel with the sizes: nobody is left waiting in a send)
    assert dll.bt_comm_alltoallv_matches(h, ds.ptr, send_sizes.ctypes.data, dr.ptr, recv_total - (1 if rank == world - 1 else 0), recv_sizes.ctypes.data) != 0
    assert dll.bt_comm_gather_summaries(h, dl.ptr, len(local), out.ptr, total - (1 if rank == 0 else 0), offs.ctypes.data) != 0
    assert dll.bt_comm_alltoallv_matches(h, None, send_sizes.ctypes.data, dr.ptr, recv_total, recv_sizes.ctypes.data) != 0     # null send buffer with sizes
    # all-gather of variable-length byte strings: rank r contributes 7 + 5r bytes of value 100 + r
    mine = np.full(7 + 5 * rank, 100 + rank, np.uint8)
    dm = ctx.to_device(mine)
    ag_total = sum(7 + 5 * q for q in range(world))
    dg = ctx.buffer(ag_total)
    ag_offs = np.zeros(world + 1, np.uint64)
    lib.check(dll.bt_comm_allgatherv(h, dm.ptr, len(mine), dg.ptr, ag_total, ag_offs.ctypes.data))
    ctx.sync()
    assert list(ag_offs) == [sum(7 + 5 * q for q in range(i)) for i in range(world + 1)]
    assert np.array_equal(dg.download(np.uint8, ag_total), np.concatenate([np.full(7 + 5 * q, 100 + q, np.uint8) for q in range(world)]))
    assert dll.bt_comm_allgatherv(h, dm.ptr, len(mine), dg.ptr, ag_total - (1 if rank == 0 else 0), ag_offs.ctypes.data) != 0
    dm.free(), dg.free()
    lib.check(dll.bt_comm_destroy(h))
    for b in (d, dl, out, ds, dr):
        b.free()
    ctx.close()
    if queue is not None:
        queue.put(rank)


def test_one_rank_communicator(gpu_ctx):
    _, dll = _comm_lib()
    ident = (C.c_uint8 * 128)()
    assert dll.bt_comm_unique_id(ident) == 0
    _rank_body(0, 1, bytes(ident), None)


def test_two_ranks_when_two_gpus_are_visible():
    from bayestyper_amd import lib

    n = C.c_int()
    lib.bt_device_count(C.byref(n))
    if n.value < 2:
        pytest.skip("one GPU visible: the two-rank exchange needs two")
    _, dll = _comm_lib()
    ident = (C.c_uint8 * 128)()
    assert dll.bt_comm_unique_id(ident) == 0
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_rank_body, args=(r, 2, bytes(ident), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1] and all(p.exitcode == 0 for p in procs)


def test_noise_chain_on_the_device_with_the_enqueued_all_reduce(gpu_ctx, oracle):
    """bt_gibbs_noise_chain, the noise drivers' chain without a host round trip: (1) against the host loop (sweep, noise counts, libstdc++'s gamma draws of
    CountDistribution) — the same rates within 1e-12, and the generator handed back continues the host's stream; (2) with the reduction hook the ranks of a
    multi-GPU run install — bt_comm_allreduce_hist enqueued on the context's stream between the histogram and the draw (a one-rank communicator here: identity)
    — the chain gives the same rates again; (3) without a sampler (a rank that holds no group in this chain) the draws still advance the generator."""
    import _oracle
    from bayestyper_amd import comm as btcomm, lib, synth
    from bayestyper_amd.host import count_model

    S, n_it = 3, 40
    flat = synth.concat([synth.make_batch("A", 50, S, seed=3, templates=5), synth.make_batch("C", 2, S, seed=4), synth.make_batch("B", 6, S, seed=5, templates=2)])
    kw = dict(seed=9, chains=1, burn=10, iters=30, noise_seeding=1)

    def cd():
        d = count_model.CountDistribution(S, prior=(1.0, 0.01), seed=kw["seed"])
        for s in range(S):
            d.set_genomic(s, 15.0, 30.0)
        return d

    # host loop
    c0 = cd()
    g = lib.Gibbs(gpu_ctx, flat, *c0.tables(), **kw)
    g.init_chain(0)
    want = []
    for it in range(n_it):
        g.sweep(1, it >= 10)
        c0.sample_noise_parameters(g.noise_counts())
        g.set_noise_lut(c0.noise_table())
        want.append(c0.noise_rates().copy())
    res_host = g.results()
    g.close()
    want = np.array(want)

    def device_chain(reduce):
        c1 = cd()
        g1 = lib.Gibbs(gpu_ctx, flat, *c1.tables(), **kw)
        g1.init_chain(0)
        m = lib.NoiseModel(gpu_ctx, [(1.0, 0.01)] * S)
        m.set_generator(*c1.export_generator())
        got = m.chain(g1, n_it, 10, reduce)
        gen = m.get_generator()
        res = g1.results()
        m.close(), g1.close()
        return got, gen, res

    got, gen, res_dev = device_chain(None)
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    w_host, s_host = c0.export_generator()
    assert np.array_equal(gen[0], w_host) and (not w_host[625] or abs(gen[1] - s_host) <= 1e-12 * max(1.0, abs(s_host)))   # (the saved variate counts only while it is available)
    for k in ("dip_off", "h1", "h2", "freq", "cell_off"):
        assert np.array_equal(res_host[k], res_dev[k]), k
    one = btcomm.Comm(gpu_ctx, btcomm.unique_id(), 0, 1)
    got2, gen2, _ = device_chain(lambda d_hist, n: one.allreduce(d_hist, n))
    assert np.array_equal(got2, got) and np.array_equal(gen2[0], gen[0])
    # a rank without groups: zero histograms, the prior's posterior, the generator advances
    m = lib.NoiseModel(gpu_ctx, [(1.0, 0.01)] * S)
    c2 = cd()
    before = c2.export_generator()
    m.set_generator(*before)
    r = m.chain(None, 5, 5, lambda d_hist, n: one.allreduce(d_hist, n))
    for it in range(5):
        c2.sample_noise_parameters(np.zeros(S * 256, np.uint64))
        assert np.allclose(r[it], c2.noise_rates(), rtol=1e-12, atol=0)
    assert np.array_equal(m.get_generator()[0], c2.export_generator()[0])
    m.close(), one.close()


def test_bench_sharded_precheck_on_a_one_rank_communicator():
    """bench.py --gpus N runs sharded_precheck before anything is timed (a 2 000-group unit sharded over the ranks, summaries gathered through
    bt_comm_gather_summaries, compared with rank 0's unsharded run).  One GPU here: the same code with a one-rank communicator — every step but the exchange
    between different GPUs — must pass, and must FAIL when the gathered summaries are not the unsharded run's."""
    import sys

    import torch

    sys.path.insert(0, ROOT)
    import bench
    from bayestyper_amd import comm as btcomm, lib

    class Dist:   # (the gloo group of a real run only carries rank 0's verdict to the others)
        @staticmethod
        def broadcast(t, src):
            return None

    ctx = lib.Ctx(0)
    dev = torch.device("cuda", 0)
    c = btcomm.Comm(ctx, btcomm.unique_id(), 0, 1)
    assert bench.sharded_precheck(ctx, c, Dist, torch, dev, 0, 1, 3) is True

    class Broken(btcomm.Comm):   # a gather that loses a word must be noticed
        def gather_words(self, d_local_ptr, local_words, d_out_ptr, out_capacity_words):
            offs = super().gather_words(d_local_ptr, local_words, d_out_ptr, out_capacity_words)
            self.ctx.sync()
            torch.cuda.synchronize()
            lib.check(lib.bt_memset(self.ctx.h, d_out_ptr, 0, 64))
            self.ctx.sync()
            return offs

    c.close()
    b = Broken(ctx, btcomm.unique_id(), 0, 1)
    with pytest.raises(RuntimeError, match="pre-check failed"):
        bench.sharded_precheck(ctx, b, Dist, torch, dev, 0, 1, 3)
    b.close()
    ctx.close()
