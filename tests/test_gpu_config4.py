"""GPU: a scaled-down BASELINE config 4 on one GPU -- the scale-factor loop
(chained OptimizeIteration passes over random (fov, z) fragments,
analysis/optimize.py:73-131, 311-345) followed by the sharded fragment loop of
Decode (core/analysistask.py:285-297, analysis/decode.py:92-175) with the
factors the last pass produced -- against the oracle's decode ->
extract_refactors -> nanmedian loop and its per-FOV barcode tables.  And
bench.py's two-rank self-launch (gloo, both ranks on this one card)."""
import json
import os
import subprocess
import sys
import warnings

import numpy as np
import pandas
import pytest

import merfish_oracle as oracle
from conftest import ROOT, ArrayCodebook, frame_to_array, load_golden, synth_stack

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('background', [False, True])
def test_config4_scaled_down_matches_oracle(gpu_device, background):
    import torch
    from merlin_amd import optimize
    from merlin_amd.decode import Decode
    cb = load_golden('g2_synth192')['codebook']
    dm = oracle.normalized_codebook(cb)
    n_fovs, Z, F, passes = 8, 2, 4, 2
    stacks = [synth_stack(cb, (Z, 192, 256), 150, seed=1234 + f, bg=100)
              for f in range(n_fovs)]
    dev = [torch.from_numpy(s).to(gpu_device) for s in stacks]
    hist = sum(optimize.pixel_histogram(d) for d in dev)
    init = optimize.initial_scale_factors(optimize.summed_pixel_histogram(hist))
    params = {'fov_per_iteration': F, 'optimize_background': background}
    chain = optimize.run_optimize_chain(
        ArrayCodebook(cb), lambda f, z: dev[f][:, z], list(range(n_fovs)), Z,
        init, passes, params, gpu_device, None, seed=99)
    assert len(chain) == passes

    # the oracle's loop over the same draws
    sf, bg = init.astype(np.float64), np.zeros(16)
    rng = np.random.RandomState(99)
    for k, it in enumerate(chain):
        draw = [[int(rng.choice(list(range(n_fovs)))),
                 int(rng.choice(list(range(Z))))] for _ in range(F)]
        assert it.parameters['fov_index'] == draw
        rfs, bgrs, seens = [], [], []
        for f, z in draw:
            di, pm, npt, d = oracle.decode_pixels(stacks[f][:, z], dm, sf, bg)
            rf, bgr, seen = oracle.extract_refactors(di, pm, npt, dm, 5,
                                                     background)
            rfs.append(rf), bgrs.append(bgr), seens.append(seen)
        prev = np.tile(sf, (F, 1))
        esf = oracle.reduce_scale_factors(np.array(rfs), prev)
        ebg = oracle.reduce_backgrounds(np.array(bgrs), np.tile(bg, (F, 1)), prev)
        np.testing.assert_allclose(it.get_scale_factors(), esf, rtol=1e-12,
                                   equal_nan=True)
        np.testing.assert_allclose(it.get_backgrounds(), ebg, rtol=1e-12,
                                   atol=1e-12, equal_nan=True)
        assert np.array_equal(it.get_barcode_count_history()[-1],
                              np.mean(seens, axis=0))
        sf, bg = esf, ebg
    assert np.all(np.isfinite(sf)) and sf.min() > 0

    # Decode of every FOV with the chain's factors, "two ranks" on one card:
    # the union of the shards is every FOV, each table equals the oracle's
    dparams = {'crop_width': 6, 'minimum_area': 1}
    task = Decode(ArrayCodebook(cb), dparams, device=gpu_device)
    gsf, gbg = chain[-1].get_scale_factors(), chain[-1].get_backgrounds()
    got = {}
    for rank in range(2):
        for f, bc, vol in task.run_fovs_sharded(lambda f: dev[f], gsf, gbg,
                                                list(range(n_fovs)), rank, 2):
            assert f % 2 == rank and f not in got
            got[f] = bc
    assert sorted(got) == list(range(n_fovs))
    total = 0
    for f in range(n_fovs):
        frames = []
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for z in range(Z):
                di, pm, npt, d = oracle.decode_pixels(stacks[f][:, z], dm, sf, bg,
                                                      0.5167, 1, 1)
                frames.append(oracle.extract_all_barcodes(
                    cb.shape[0], di, pm, npt, d, f, 6, z, None, 1))
            exp = pandas.concat(frames)
        r, e = frame_to_array(got[f], 16), frame_to_array(exp, 16)
        assert r.shape == e.shape
        for c in (0, 1, 4, 13):
            assert np.array_equal(r[:, c], e[:, c])
        np.testing.assert_allclose(r[:, :14], e[:, :14], rtol=1e-12, atol=1e-12)
        assert np.array_equal(r[:, 14:], e[:, 14:])
        assert np.array_equal(got[f].index.to_numpy(), exp.index.to_numpy())
        total += len(r)
    assert total > 200


def test_bench_two_ranks_gloo_one_card(gpu_device):
    """`python bench.py --gpus 2 --backend gloo` with no launcher: two rank
    processes (sharing this card), n_gpus == 2 in the line, config 4 sharded
    over them."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
         '--backend', 'gloo', '--steps', '4', '--warmup', '4',
         '--config4-fovs', '6', '--config4-fragments', '4',
         '--config4-size', '256'],
        env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2
    assert out['optimize_exchange']['rccl_ranks'] == 2
    c4 = out['config4']
    assert c4['ranks'] == 2 and c4['fovs'] == 6
    assert c4['exchange_own_rows_intact'] is True
    assert c4['decode']['records'] > 0
    assert 'roofline' in out and out['roofline']['frac'] > 0


@pytest.mark.parametrize('mode', ['plain_', 'bg_'])
def test_optimize_chain_matches_reference_trace(gpu_device, mode):
    """Two chained OptimizeIteration passes on the GPU against the trace the
    reference's own analysis/optimize.py produced on the same images
    (tests/golden/optimize_trace_reference.npz): same fragment draws, per-fragment
    refactors / counts, scale factors, backgrounds and histories.  Counts and
    the NaN / zero pattern are exact; values agree to the float32 rounding of
    the pixel magnitudes (the reference's go through BLAS sdot)."""
    import torch
    from conftest import GOLDEN_DIR
    from merlin_amd import optimize
    g = np.load(os.path.join(GOLDEN_DIR, 'optimize_trace_reference.npz'))
    g7 = {k: g[k] for k in g.files}
    imgs = [torch.from_numpy(i).to(gpu_device) for i in g7['images']]
    hist = sum(optimize.pixel_histogram(d) for d in imgs)
    init = optimize.initial_scale_factors(optimize.summed_pixel_histogram(hist))
    assert np.array_equal(init, g7[mode + 'it0_previous_scale_factors'][0])
    chain = optimize.run_optimize_chain(
        ArrayCodebook(g7['codebook']), lambda f, z: imgs[f][:, z],
        list(range(len(imgs))), g7['images'].shape[2], init, 2,
        {'fov_per_iteration': int(g7['F']), 'optimize_background': mode == 'bg_'},
        gpu_device, None, seed=int(g7['seed']))
    for k, it in enumerate(chain):
        p = mode + 'it{}_'.format(k)
        assert it.parameters['fov_index'] == g7[p + 'fov_index'].tolist()
        assert np.array_equal(it._counts, g7[p + 'barcode_counts'])
        exp = g7[p + 'scale_refactors']
        assert np.array_equal(np.isnan(it._refactors), np.isnan(exp))
        assert np.array_equal(it._refactors == 0, exp == 0)
        np.testing.assert_allclose(it._refactors, exp, rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(it._bg_refactors, g7[p + 'background_refactors'],
                                   rtol=1e-6, atol=1e-9, equal_nan=True)
        np.testing.assert_allclose(it.get_scale_factors(), g7[p + 'scale_factors'],
                                   rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(it.get_backgrounds(), g7[p + 'backgrounds'],
                                   rtol=1e-6, atol=1e-7, equal_nan=True)
        np.testing.assert_allclose(it.get_scale_factor_history(),
                                   g7[p + 'scale_factor_history'], rtol=1e-6)
        assert np.array_equal(it.get_barcode_count_history(),
                              g7[p + 'barcode_count_history'])
        # the fragments' barcode tables: as many rows as the reference wrote
        assert [len(it.barcodes[i]) for i in range(int(g7['F']))] == \
            g7[p + 'barcodes_written'].tolist()
    # the bit no barcode uses keeps its initial factor (refactors == 0 -> 1)
    assert chain[-1].get_scale_factors()[-1] == init[-1]
