"""The one-launch form's hand-over by polling (csrc/lf_free.h: PART_EMPTY; option "poll", default on) against the tile counter
(poll = 0): the same partial sums added in the same order, so the same bits - for every variant, across changes of the batch
size, after calls that leave sums in the partial-sum buffers (the two-piece diagnostics take three launches), with rows on the
careful path (their tiles count, the others poll) and inside the device sampler."""
import numpy as np
import pytest

from lf_testlib import make_inputs, synth

pytestmark = pytest.mark.gpu


def _rows(variant, B, seed):
    th = synth.walkers(variant, B, seed=seed)
    if B >= 16:
        th[1, 0] = 40.2                                     # underflow zone: -inf
        th[2, 3 if variant == "zevol" else 1] = 6.0         # outside the prior
        if variant == "zevol":                              # near the underflow boundary: the careful path (one tile only)
            th[11, :3] = (40.75, 40.8, 40.85)
        elif variant == "fixcomp":
            th[11, 0] = 40.75
    return th


@pytest.mark.parametrize("variant,n", [("free", 1000000), ("free", 3001), ("zevol", 200003), ("fixcomp", 50021)])
def test_polling_equals_counting(variant, n):
    from lumfuncmcmc_amd.capi import LFContext
    inp = make_inputs(variant, n, seed=211)
    ctx = LFContext(inp, max_batch=64)
    got = {}
    for poll in (1, 0, 1):
        ctx.set_option("poll", poll)
        out = []
        for B in (128, 40, 300, 1, 128):                    # (grows the workspace on the way: the slots are refilled)
            th = _rows(variant, B, 212 + B)
            out.append(ctx.lnprob_batch(th))
            assert ctx.last_launch()["fused"], ctx.last_launch()
            if B == 40:
                ctx.lnprob_pieces(th)                       # three launches: leaves sums behind in the buffers
            out.append(ctx.lnprob_batch(th))                # and again: the finisher left the slots empty
        got[poll] = got.get(poll, []) + [np.concatenate(out)]
    ctx.close()
    a, b, c = got[1][0], got[0][0], got[1][1]
    assert not np.isnan(a).any() and np.isfinite(a).sum() > 0.8 * len(a)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, c)


@pytest.mark.parametrize("variant", ["free", "zevol", "fixcomp"])
def test_polling_inside_the_sampler(variant):
    from lumfuncmcmc_amd.capi import LFContext
    from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
    inp = make_inputs(variant, 100003, seed=221)
    chains = []
    for poll in (1, 0):
        ctx = LFContext(inp)
        ctx.set_option("poll", poll)
        W = 64
        ds = DeviceEnsembleSampler(ctx, W, seed=5, capacity=40)
        ds.run_mcmc(synth.walkers(variant, W, seed=222), 30)
        chains.append((ds.chain.copy(), ds.lnprobability.copy()))
        assert ctx.last_launch()["fused"]
        ds.close()
        ctx.close()
    np.testing.assert_array_equal(chains[0][0], chains[1][0])
    np.testing.assert_array_equal(chains[0][1], chains[1][1])
