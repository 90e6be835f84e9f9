"""CPU: OptimizeIteration's cross-fragment reduction against a trace made by
the reference's own analysis/optimize.py (oracle/make_optimize_goldens.py ->
tests/golden/optimize_trace_reference.npz): two chained iterations, fragment draws by
the reference's constructor, per-fragment refactors incl. NaN rows (an empty
field of view) and zeros (a bit that is never seen), optimize_background off
and on.  Bar: bit-exact -- the reduction is the same numpy call sequence
(optimize.py:328-345, 358-373, 406-416)."""
import os

import numpy as np
import pytest

import merfish_oracle as oracle
from conftest import GOLDEN_DIR, ArrayCodebook
from merlin_amd import optimize


@pytest.fixture(scope='module')
def g7():
    g = np.load(os.path.join(GOLDEN_DIR, 'optimize_trace_reference.npz'))
    return {k: g[k] for k in g.files}


def _eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize('mode', ['plain_', 'bg_'])
def test_oracle_reduction_matches_reference_trace(g7, mode):
    for it in ('it0_', 'it1_'):
        p = mode + it
        assert np.isnan(g7[p + 'scale_refactors']).all(axis=1).any()   # NaN rows
        assert (g7[p + 'scale_refactors'][:, -1] == 0).any()           # the 0 -> 1 rule
        sf = oracle.reduce_scale_factors(g7[p + 'scale_refactors'],
                                         g7[p + 'previous_scale_factors'])
        bg = oracle.reduce_backgrounds(g7[p + 'background_refactors'],
                                       g7[p + 'previous_backgrounds'],
                                       g7[p + 'previous_scale_factors'])
        assert _eq(sf, g7[p + 'scale_factors'])
        assert _eq(bg, g7[p + 'backgrounds'])
    # the chain: iteration 1 started from iteration 0's result
    assert _eq(g7[mode + 'it1_previous_scale_factors'][0],
               g7[mode + 'it0_scale_factors'])
    assert _eq(g7[mode + 'it1_previous_backgrounds'][0],
               g7[mode + 'it0_backgrounds'])


def test_oracle_initial_scale_factors_match_reference_trace(g7):
    imgs = g7['images']
    hist = np.zeros((16, 65534))
    for img in imgs:
        for b in range(16):
            hist[b] += np.histogram(img[b], bins=np.arange(0, 65535, 1))[0]
    assert _eq(oracle.initial_scale_factors(hist),
               g7['plain_it0_previous_scale_factors'][0])
    assert _eq(optimize.initial_scale_factors(hist),
               g7['plain_it0_previous_scale_factors'][0])


@pytest.mark.parametrize('mode', ['plain_', 'bg_'])
def test_oracle_fragments_match_reference_trace(g7, mode):
    """decode -> extract_refactors of every fragment, oracle vs the reference's
    _run_analysis (scale_refactors, background_refactors, barcode_counts)."""
    dm = oracle.normalized_codebook(g7['codebook'])
    p = mode + 'it0_'
    sf, bg = g7[p + 'previous_scale_factors'][0], g7[p + 'previous_backgrounds'][0]
    for k, (f, z) in enumerate(g7[p + 'fov_index']):
        di, pm, npt, d = oracle.decode_pixels(g7['images'][f][:, z], dm, sf, bg)
        rf, bgr, seen = oracle.extract_refactors(di, pm, npt, dm, 5, mode == 'bg_')
        assert _eq(seen, g7[p + 'barcode_counts'][k])
        # the reference's pixel magnitudes go through BLAS sdot (summation
        # order of the BLAS build), the oracle's through the sequential sum:
        # float32 rounding of the magnitudes (2e-7) carries into the means
        np.testing.assert_allclose(rf, g7[p + 'scale_refactors'][k], rtol=1e-6,
                                   atol=0, equal_nan=True)
        np.testing.assert_allclose(bgr, g7[p + 'background_refactors'][k],
                                   rtol=1e-6, atol=1e-9, equal_nan=True)
        assert _eq(np.isnan(rf), np.isnan(g7[p + 'scale_refactors'][k]))
        assert _eq(rf == 0, g7[p + 'scale_refactors'][k] == 0)


@pytest.mark.parametrize('mode', ['plain_', 'bg_'])
def test_host_iteration_matches_reference_trace(g7, mode):
    """merlin_amd.optimize.OptimizeIteration fed the reference's per-fragment
    results: same draws from the same seed, same factors, backgrounds and
    histories, bit for bit."""
    cb = ArrayCodebook(g7['codebook'])
    F = int(g7['F'])
    rng = np.random.RandomState(int(g7['seed']))
    prev = None
    its = []
    for k in range(2):
        p = mode + 'it{}_'.format(k)
        table = {(int(f), int(z)): i for i, (f, z) in enumerate(g7[p + 'fov_index'])}

        def fragment(decoder, fov, z, sf, bg, p=p, table=table):
            i = table[(fov, z)]
            # every fragment sees the previous iteration's factors
            assert _eq(sf, g7[p + 'previous_scale_factors'][i])
            assert _eq(bg, g7[p + 'previous_backgrounds'][i])
            return (g7[p + 'scale_refactors'][i], g7[p + 'background_refactors'][i],
                    g7[p + 'barcode_counts'][i])

        it = optimize.OptimizeIteration(
            cb, {'fov_per_iteration': F, 'optimize_background': mode == 'bg_'},
            fovs=range(len(g7['images'])), zCount=g7['images'].shape[2],
            previousIteration=prev,
            initialScaleFactors=g7[p + 'previous_scale_factors'][0] if k == 0 else None,
            fragmentFunction=fragment, seed=rng).run()
        assert it.parameters['fov_index'] == g7[p + 'fov_index'].tolist()
        assert _eq(it.get_scale_factors(), g7[p + 'scale_factors'])
        assert _eq(it.get_backgrounds(), g7[p + 'backgrounds'])
        assert _eq(it.get_scale_factor_history(), g7[p + 'scale_factor_history'])
        assert _eq(it.get_barcode_count_history(), g7[p + 'barcode_count_history'])
        prev = it
        its.append(it)
    assert its[1].get_scale_factor_history().shape == (2, 16)
