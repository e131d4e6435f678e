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
