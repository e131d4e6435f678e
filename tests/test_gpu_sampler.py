"""The device-resident ensemble sampler against a host replay of the same algorithm with the same
Philox4x32-10 random numbers: identical accept/reject decisions, hence identical chains."""
import numpy as np
import pytest

from lf_testlib import make_inputs, synth

pytestmark = pytest.mark.gpu

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (numpy uint64 arithmetic)."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & MASK for x in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0 & MASK), np.uint64(k1 & MASK)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return c0, c1, c2, c3


def draw(step, half, w, stream, seed):
    w = np.asarray(w, dtype=np.uint64)
    c0 = np.full_like(w, step & MASK)
    c1 = np.full_like(w, ((step >> 32) & MASK) ^ ((half << 31) & MASK))
    return philox4x32(c0, c1, w, np.full_like(w, stream), seed & MASK, (seed >> 32) & MASK)


def u53(hi, lo):
    return (((hi << np.uint64(32)) | lo) >> np.uint64(11)).astype(np.float64) * 1.1102230246251565e-16


def host_replay(ctx, pos, nsteps, seed, a=2.0):
    W, nd = pos.shape
    half = W // 2
    p = pos.copy()
    lp = ctx.lnprob_batch(p)
    chain = np.empty((W, nsteps, nd))
    lnps = np.empty((W, nsteps))
    nacc = np.zeros(W, dtype=np.int64)
    w = np.arange(half)
    for step in range(nsteps):
        for h in (0, 1):
            r0, r1, r2, _ = draw(step, h, w, 0, seed)
            z = ((a - 1.0) * u53(r0, r1) + 1.0) ** 2 / a
            j = (1 - h) * half + ((r2 * np.uint64(half)) >> np.uint64(32)).astype(np.int64)
            k = h * half + w
            prop = p[j] - (p[j] - p[k]) * z[:, None]
            newlp = ctx.lnprob_batch(prop)
            q0, q1, _, _ = draw(step, h, w, 1, seed)
            with np.errstate(all="ignore"):
                lnq = (nd - 1.0) * np.log(z) + newlp - lp[k]
                acc = (np.log(u53(q0, q1)) < lnq) & (newlp > -np.inf)
            p[k[acc]] = prop[acc]
            lp[k[acc]] = newlp[acc]
            nacc[k[acc]] += 1
            chain[k, step] = p[k]
            lnps[k, step] = lp[k]
    return chain, lnps, nacc


@pytest.mark.parametrize("variant,n,W", [("fixcomp", 3000, 16), ("free", 2000, 32), ("zevol", 2000, 20)])
def test_device_chain_equals_host_replay(variant, n, W):
    from lumfuncmcmc_amd.capi import LFContext
    from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
    inp = make_inputs(variant, n, seed=31)
    ctx = LFContext(inp)
    pos = synth.walkers(variant, W, seed=32)
    pos[1] = 99.0                                  # one walker starts outside the prior (-inf)
    nsteps, seed = 25, 0x1234567890ABCDEF
    ds = DeviceEnsembleSampler(ctx, W, seed=seed, capacity=nsteps)
    ds.run_mcmc(pos, nsteps)
    chain, lnps, nacc = host_replay(ctx, pos, nsteps, seed)
    assert ds.chain.shape == (W, nsteps, ctx.ndim)
    assert np.array_equal(ds.naccepted, nacc)
    assert np.array_equal(ds.chain, chain)
    assert np.array_equal(ds.lnprobability, lnps)
    assert 0 < ds.acceptance_fraction.mean() < 1
    # continuing = one longer run
    ds2 = DeviceEnsembleSampler(ctx, W, seed=seed, capacity=nsteps)
    ds2.run_mcmc(pos, 10)
    ds2.run_mcmc(None, 15)
    assert np.array_equal(ds2.chain, ds.chain)
    with pytest.raises(Exception):
        ds2.run_mcmc(None, 1)                      # capacity exceeded is an error, not a wrap-around
    ds.close(); ds2.close(); ctx.close()


def test_device_sampler_recovers_a_posterior():
    """End to end on the likelihood itself: sample the fixed-completeness posterior of a catalogue drawn
    from a known Schechter function and check that the chain concentrates near the truth."""
    from lumfuncmcmc_amd.capi import LFContext
    from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
    inp = make_inputs("fixcomp", 20000, seed=41)
    ctx = LFContext(inp)
    rng = np.random.default_rng(0)
    p0 = np.array([42.5, -2.0, -1.49]) + 0.05 * rng.normal(size=(32, 3))
    ds = DeviceEnsembleSampler(ctx, 32, seed=7, capacity=400)
    ds.run_mcmc(p0, 400)
    lp = ds.lnprobability
    assert np.isfinite(lp[:, -1]).all()
    assert np.median(lp[:, -50:]) > np.median(lp[:, :5])        # climbs towards the mode
    assert 0.05 < ds.acceptance_fraction.mean() < 0.95
    assert np.all(np.isfinite(ds.acor))
    ds.close(); ctx.close()


def test_sharded_form_gives_the_same_chain_single_rank():
    from lumfuncmcmc_amd.capi import LFContext
    from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
    inp = make_inputs("free", 2500, seed=51)
    ctx = LFContext(inp)
    W, nsteps, seed = 24, 12, 99
    pos = synth.walkers("free", W, seed=52)
    a = DeviceEnsembleSampler(ctx, W, seed=seed, capacity=nsteps)
    a.run_mcmc(pos, nsteps)
    b = DeviceEnsembleSampler(ctx, W, seed=seed, capacity=nsteps)
    b.enqueue_sharded(pos, nsteps)
    b.sync()
    assert np.array_equal(a.chain, b.chain) and np.array_equal(a.lnprobability, b.lnprobability)
    assert np.array_equal(a.naccepted, b.naccepted)
    a.close(); b.close(); ctx.close()


def _sharded_worker(rank, world, port, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)                      # rehearsal on a one-GPU box: every rank shares device 0
    from lf_testlib import make_inputs, synth
    from lumfuncmcmc_amd.capi import LFContext
    from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
    inp = make_inputs("zevol", 3000, seed=61)
    ctx = LFContext(inp)
    W, nsteps, seed = 22, 8, 1234               # half = 11: ragged over 2 ranks (6 + 5)
    pos = synth.walkers("zevol", W, seed=62)
    s = DeviceEnsembleSampler(ctx, W, seed=seed, capacity=nsteps)
    s.enqueue_sharded(pos, nsteps)
    s.sync()
    ref = None
    if rank == 0:
        f = DeviceEnsembleSampler(ctx, W, seed=seed, capacity=nsteps)
        f.run_mcmc(pos, nsteps)
        ref = (f.chain, f.lnprobability)
        f.close()
    q.put((rank, s.chain, s.lnprobability, ref))
    dist.barrier()
    s.close(); ctx.close()
    dist.destroy_process_group()


def test_sharded_sampler_two_ranks_on_one_gpu():
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, c0, l0, ref), (_, c1, l1, _) = res
    assert np.array_equal(c0, c1) and np.array_equal(l0, l1)          # ranks stay in lock-step
    assert np.array_equal(c0, ref[0]) and np.array_equal(l0, ref[1])  # and reproduce the one-GPU chain


def _srcshard_worker(rank, world, port, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from lf_testlib import make_inputs, synth
    from lumfuncmcmc_amd.capi import LFContext
    from lumfuncmcmc_amd.dist import SourceShardedLnProb
    out = {}
    for variant in ("free", "fixcomp", "zevol"):
        inp = make_inputs(variant, 5003, seed=71)
        th = synth.walkers(variant, 9, seed=72)
        th[3, 0] = 40.2                          # underflows on the rank that holds the brightest source
        th[4, 1] = 6.0                           # outside the prior
        sh = SourceShardedLnProb(inp, 0)
        got = sh(th)
        sh.close()
        ref = None
        if rank == 0:
            c = LFContext(inp)
            ref = c.lnprob_batch(th)
            c.close()
        out[variant] = (got, ref)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_source_sharded_lnprob_two_ranks_on_one_gpu():
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_srcshard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for variant in ("free", "fixcomp", "zevol"):
        g0, ref = res[0][1][variant]
        g1, _ = res[1][1][variant]
        assert np.array_equal(g0, g1)                               # the all-reduce gives every rank the same sums
        assert np.array_equal(np.isinf(g0), np.isinf(ref)) and np.isinf(ref).sum() == 2
        fin = np.isfinite(ref)
        np.testing.assert_allclose(g0[fin], ref[fin], rtol=1e-13)   # only the summation order differs


def _srcshard_sampler_worker(rank, world, port, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from lf_testlib import make_inputs, synth
    from lumfuncmcmc_amd.capi import LFContext
    from lumfuncmcmc_amd.dist import shard_sources
    from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
    out = {}
    for variant in ("free", "zevol"):
        inp = make_inputs(variant, 6007, seed=81)
        ctx = LFContext(shard_sources(inp, rank, world))
        ctx.set_option("grid_share", rank + 65536 * world)
        W, nsteps, seed = 26, 10, 4321
        pos = synth.walkers(variant, W, seed=82)
        s = DeviceEnsembleSampler(ctx, W, seed=seed, capacity=nsteps)
        s.enqueue_sharded(pos, nsteps, shard="sources")
        s.sync()
        ref = None
        if rank == 0:
            full = LFContext(inp)
            f = DeviceEnsembleSampler(full, W, seed=seed, capacity=nsteps)
            f.run_mcmc(pos, nsteps)
            ref = (f.chain, f.lnprobability, f.naccepted)
            f.close(); full.close()
        out[variant] = (s.chain, s.lnprobability, s.naccepted, ref)
        s.close(); ctx.close()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_source_sharded_sampler_two_ranks_on_one_gpu():
    """Strong-scaling form for small ensembles: every rank holds half of every field's sources and half of the grid
    chunks, evaluates the WHOLE half-ensemble on its shard, all-reduce(SUM), accept everywhere.  The accept decisions
    compare lnprob differences of O(1) against sums whose order changed by ~1e-9 absolute, so over a short chain they
    agree with the one-GPU chain; positions then agree exactly and lnprob to the summation order."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_srcshard_sampler_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for variant in ("free", "zevol"):
        c0, l0, n0, ref = res[0][1][variant]
        c1, l1, n1, _ = res[1][1][variant]
        assert np.array_equal(c0, c1) and np.array_equal(l0, l1) and np.array_equal(n0, n1)   # ranks in lock-step
        assert np.array_equal(n0, ref[2])                      # the same accept decisions as on one GPU
        assert np.array_equal(c0, ref[0])                      # hence the same positions, bit for bit
        fin = np.isfinite(ref[1])
        assert np.array_equal(np.isfinite(l0), fin)
        np.testing.assert_allclose(l0[fin], ref[1][fin], rtol=1e-13)
