"""The ensemble sampler on the batched boundary (CPU: a Gaussian target stands in for lnprob)."""
import numpy as np
import pytest

from lumfuncmcmc_amd.sampler import EnsembleSampler, integrated_time


def test_gaussian_target_moments():
    mu = np.array([1.0, -2.0, 0.5])
    sig = np.array([0.5, 2.0, 1.0])
    calls = []

    def lnp(block):
        calls.append(block.shape)
        return -0.5 * np.sum(((block - mu) / sig) ** 2, axis=1)

    s = EnsembleSampler(32, 3, lnp, seed=1)
    p0 = np.random.default_rng(0).normal(size=(32, 3))
    s.run_mcmc(p0, 1500)
    assert s.chain.shape == (32, 1500, 3) and s.lnprobability.shape == (32, 1500)
    # one call for the start, then two half-ensemble blocks per step
    assert calls[0] == (32, 3) and all(c == (16, 3) for c in calls[1:]) and len(calls) == 1 + 2 * 1500
    flat = s.chain[:, 500:, :].reshape(-1, 3)
    assert np.allclose(flat.mean(axis=0), mu, atol=0.15)
    assert np.allclose(flat.std(axis=0), sig, rtol=0.15)
    assert 0.2 < s.acceptance_fraction.mean() < 0.9
    tau = s.acor
    assert tau.shape == (3,) and np.all(np.isfinite(tau)) and np.all(tau > 0)


def test_minus_inf_is_never_accepted_and_nan_raises():
    def lnp(block):
        out = -0.5 * np.sum(block ** 2, axis=1)
        out[block[:, 0] > 1.0] = -np.inf
        return out
    s = EnsembleSampler(8, 2, lnp, seed=3)
    p0 = np.random.default_rng(1).uniform(-1, 1, size=(8, 2))
    s.run_mcmc(p0, 200)
    assert (s.chain[:, :, 0] <= 1.0).all()
    with pytest.raises(ValueError):
        EnsembleSampler(8, 2, lambda b: np.full(len(b), np.nan), seed=1).run_mcmc(p0, 1)
    with pytest.raises(ValueError):
        EnsembleSampler(3, 2, lnp)


def test_short_chain_autocorr_is_finite():
    x = np.random.default_rng(0).normal(size=(50, 8))
    assert np.isfinite(integrated_time(x)) and integrated_time(x[:3]) == 1.0


def test_same_seed_same_chain():
    f = lambda b: -0.5 * np.sum(b ** 2, axis=1)
    p0 = np.random.default_rng(2).normal(size=(8, 2))
    a = EnsembleSampler(8, 2, f, seed=7); a.run_mcmc(p0, 20)
    b = EnsembleSampler(8, 2, f, seed=7); b.run_mcmc(p0, 20)
    assert np.array_equal(a.chain, b.chain)


def test_config1_cpu_plumbing():
    """BASELINE config 1 on the CPU: 1k synthetic sources, 32 walkers, 50 steps, the NumPy path (the
    oracle stands in for lnprob here - test infrastructure - to exercise the sampler and the sample
    bookkeeping of fit_model without a GPU)."""
    import os
    import lf_oracle as O
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fixcomp_n1000.npz"))
    inp = O.inputs_from_golden(g, "fixcomp")
    rng = np.random.default_rng(0)
    p0 = np.array([42.5, -2.0, -1.49]) + 0.05 * rng.normal(size=(32, 3))
    s = EnsembleSampler(32, 3, lambda b: O.lnprob_batch(inp, b), seed=4)
    s.run_mcmc(p0, 50)
    assert s.chain.shape == (32, 50, 3) and np.isfinite(s.lnprobability).all()
    tau = np.max(s.acor)
    burn = min(int(tau * 3), 25)
    samples = np.concatenate([s.chain, s.lnprobability[:, :, None]], axis=2)[:, burn:, :].reshape(-1, 4)
    assert samples.shape[1] == 4 and samples.shape[0] == 32 * (50 - burn)
    assert s.nevals == 32 * 51
