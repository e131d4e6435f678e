"""The N>1 path on CPU: world_size-2 (and 3, ragged) gloo process groups.  The per-rank evaluator
is the NumPy oracle (allowed in tests only) so that what is exercised here is the walker slicing,
the padded all-gather and the lock-step sampler - the same code bench.py and the classes run
over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lf_oracle as O
    from lf_testlib import make_inputs
    from lumfuncmcmc_amd import synth
    from lumfuncmcmc_amd.dist import ShardedLnProb
    from lumfuncmcmc_amd.sampler import EnsembleSampler
    inp = make_inputs("fixcomp", 400, seed=3, S=24)
    calls = []

    def local_eval(t):                       # stands in for LFContext.lnprob_torch
        calls.append(t.shape[0])
        return torch.from_numpy(O.lnprob_batch(inp, t.numpy()))

    sh = ShardedLnProb(local_eval, 3, torch.device("cpu"))
    th = synth.walkers("fixcomp", B, seed=4)
    got = sh(th)
    ref = O.lnprob_batch(inp, th)
    # deferred gathers (bench.py's plain loop: a block's gather overlaps the next block's evaluation): five independent
    # blocks in a row, complete after flush() - and a buffer is only reused once its gather of two calls ago is done
    blocks = [torch.from_numpy(synth.walkers("fixcomp", B, seed=20 + i)) for i in range(5)]
    outs = [sh.evaluate_tensor(b, defer=True) for b in blocks[:2]]
    kept = []
    for b in blocks[2:]:
        kept.append([o.clone() for o in outs[-2:]])            # (cloned while still valid: the NEXT call may reuse the older one)
        outs.append(sh.evaluate_tensor(b, defer=True))
    sh.flush()
    for i in (3, 4):
        assert np.array_equal(outs[i].numpy(), O.lnprob_batch(inp, blocks[i].numpy())), i
    assert np.array_equal(sh.evaluate_tensor(blocks[0]).numpy(), O.lnprob_batch(inp, blocks[0].numpy()))
    # a short lock-step chain: every rank runs the same sampler on the sharded callable
    smp = EnsembleSampler(8, 3, sh, seed=11)
    p, lp, _ = smp.run_mcmc(synth.walkers("fixcomp", 8, seed=5), 3)
    q.put((rank, got, ref, list(calls), p, lp))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 16), (2, 7), (3, 10)])
def test_sharded_lnprob_gloo(world, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    per = (B + world - 1) // world
    for rank, got, ref, calls, p, lp in res:
        assert np.array_equal(got, ref)                 # every rank holds the whole block, in order
        assert calls[0] == max(0, min(per, B - rank * per))    # and evaluated only its own slice
        assert np.array_equal(p, res[0][4]) and np.array_equal(lp, res[0][5])   # chains stay identical


def test_slice_bounds():
    from lumfuncmcmc_amd.dist import slice_bounds
    b, per = slice_bounds(10, 4)
    assert per == 3 and b == [(0, 3), (3, 6), (6, 9), (9, 10)]
    b, per = slice_bounds(2, 4)
    assert per == 1 and b == [(0, 1), (1, 2), (2, 2), (2, 2)]


def _fit_worker(rank, world, port, q, diverge):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lf_oracle as O
    from lumfuncmcmc_amd import synth
    from lumfuncmcmc_amd.dist import ShardedLnProb
    from lumfuncmcmc_amd.model import LumFuncMCMC
    from lumfuncmcmc_amd.sampler import EnsembleSampler
    np.random.seed(1000 + 17 * rank)             # every process has its OWN numpy state, as under torchrun
    cat = synth.catalogue(300, seed=5)
    fi = cat["field_ind"]
    m = LumFuncMCMC(synth.split_fields(cat["z"], fi), lum=synth.split_fields(cat["lum"], fi),
                    lum_e=synth.split_fields(cat["lum_e"], fi), Flim=list(synth.FLIM), alpha=synth.ALPHA_C,
                    Omega_0=list(synth.OMEGA_0), sch_al=synth.SCH_AL, sch_al_lims=synth.SCH_AL_LIMS, Lstar=synth.LSTAR,
                    Lstar_lims=[41.5, 44.5], phistar=synth.PHISTAR, phistar_lims=[-4.0, 0.0], Lc=synth.LC, Lh=synth.LH,
                    nwalkers=8, nsteps=4, min_comp_frac=0.0, field_ind=fi, fix_comp=True, Flim_lims=synth.FLIM_LIMS,
                    alpha_lims=synth.ALPHA_LIMS)
    inp = m.kernel_inputs()
    inp["lims"] = {k: list(v) for k, v in inp["lims"].items()}

    def local_eval(t):                           # stands in for LFContext.lnprob_torch (tests may use the oracle)
        return torch.from_numpy(O.lnprob_batch(inp, t.numpy()))

    m.lnprob_fn = ShardedLnProb(local_eval, 3, torch.device("cpu"), check_theta=True)
    if diverge:
        # what the broadcast prevents: a sampler that is NOT seeded from rank 0 proposes different moves per rank
        try:
            s = EnsembleSampler(8, 3, m.lnprob_fn)          # seed=None: per-process entropy
            s.run_mcmc(m.get_init_walker_values(), 2)
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
        return                                   # (no barrier: the ranks may have failed at different calls)
    m.fit_model()
    replay = None
    if rank == 0:
        s = EnsembleSampler(8, 3, lambda th: O.lnprob_batch(inp, np.atleast_2d(th)), seed=m.sampler_seed)
        s.run_mcmc(m.start_pos, 4)
        replay = s.chain
    q.put((rank, m.sampler.chain, m.sampler.lnprobability, replay))
    dist.barrier()
    dist.destroy_process_group()


def test_fit_model_unseeded_ranks_stay_in_lock_step():
    """fit_model() over a sharded callable with NOTHING pre-seeded: start positions and the sampler seed must come from
    rank 0, or the ranks propose different moves and all-gather lnprob of different theta blocks (silently wrong)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, q, False)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, c0, l0, replay), (_, c1, l1, _) = res
    assert np.array_equal(c0, c1) and np.array_equal(l0, l1)
    assert np.array_equal(c0, replay)            # = the one-rank chain from the same start and seed
    assert np.isfinite(l0).any()


def test_sharded_lnprob_detects_diverged_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all("different theta blocks" in r[1] for r in res), res
