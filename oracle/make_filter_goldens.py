"""Golden vectors for barcode filtering from the REFERENCE's own
``merlin/analysis/filterbarcodes.py``.

Run in the build container only (the reference does not travel):

    /opt/conda/bin/python3.9 oracle/make_filter_goldens.py

``filterbarcodes.py`` is imported unmodified; the modules it subclasses
(``merlin.core.analysistask``, ``merlin.analysis.decode``) pull in MERlin's
data-set / storage stack (PyTables, dotenv, boto ...), which is out of scope
and not importable here, so they are replaced by stub base classes and the
task object is created without its constructor; a small in-memory data set
serves the ``load_numpy_analysis_result`` / ``load_analysis_task`` calls its
arithmetic methods make.  Every number saved comes out of the reference's own
methods: ``_extract_counts``, ``get_blank_fraction_histogram``,
``calculate_misidentification_rate_for_threshold``,
``calculate_threshold_for_misidentification_rate``,
``calculate_barcode_count_for_threshold``, ``extract_barcodes_with_threshold``.

Output: ``tests/golden/fb_adaptive.npz`` (data only).
"""
import os
import sys
import types

import numpy as np
import pandas

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')


def install_stubs():
    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m
    pkg('merlin', os.path.join(REF, 'merlin'))
    pkg('merlin.core', os.path.join(REF, 'merlin', 'core'))
    pkg('merlin.analysis', os.path.join(REF, 'merlin', 'analysis'))
    at = types.ModuleType('merlin.core.analysistask')

    class AnalysisTask(object):
        pass

    class ParallelAnalysisTask(AnalysisTask):
        pass
    at.AnalysisTask = AnalysisTask
    at.ParallelAnalysisTask = ParallelAnalysisTask
    sys.modules['merlin.core.analysistask'] = at
    sys.modules['merlin.core'].analysistask = at
    dec = types.ModuleType('merlin.analysis.decode')

    class BarcodeSavingParallelAnalysisTask(ParallelAnalysisTask):
        pass
    dec.BarcodeSavingParallelAnalysisTask = BarcodeSavingParallelAnalysisTask
    sys.modules['merlin.analysis.decode'] = dec
    sys.modules['merlin.analysis'].decode = dec


class FakeCodebook:
    def __init__(self, n, blanks):
        self.n, self.blanks = n, list(blanks)

    def get_blank_indexes(self):
        return self.blanks

    def get_coding_indexes(self):
        return [i for i in range(self.n) if i not in self.blanks]


class FakeDecodeTask:
    def __init__(self, codebook, distance_threshold):
        self._cb = codebook
        self.parameters = {'distance_threshold': distance_threshold}

    def get_codebook(self):
        return self._cb


class FakeDataSet:
    def __init__(self, decodeTask):
        self.results = {}
        self.decodeTask = decodeTask

    def load_numpy_analysis_result(self, name, task, *a):
        return self.results[name]

    def load_analysis_task(self, name):
        return self.decodeTask


def synthetic_barcodes(n, fov, nCodes, blanks, rng):
    ids = rng.integers(0, nCodes, n)
    isBlank = np.isin(ids, blanks)
    # blanks: dimmer, farther, smaller; and thinned (they are rarer per code)
    keep = ~isBlank | (rng.random(n) < 0.25)
    ids, isBlank = ids[keep], isBlank[keep]
    n = len(ids)
    inten = np.where(isBlank, rng.lognormal(5.7, 0.3, n), rng.lognormal(6.0, 0.3, n))
    dist = np.where(isBlank, rng.beta(4, 3, n), rng.beta(3, 4, n)) * 0.5167
    area = np.where(isBlank, rng.integers(1, 6, n), rng.integers(1, 9, n))
    return pandas.DataFrame({
        'barcode_id': ids.astype(np.uint16), 'fov': np.uint16(fov),
        'mean_intensity': inten.astype(np.float32),
        'min_distance': dist.astype(np.float32),
        'area': area.astype(np.uint16)})


def main():
    install_stubs()
    from merlin.analysis import filterbarcodes as fb
    rng = np.random.default_rng(2024)
    nCodes, blanks = 140, list(range(130, 140))
    cb = FakeCodebook(nCodes, blanks)
    decodeTask = FakeDecodeTask(cb, 0.5167)
    ds = FakeDataSet(decodeTask)
    task = fb.GenerateAdaptiveThreshold.__new__(fb.GenerateAdaptiveThreshold)
    task.dataSet = ds
    task.parameters = {'tolerance': 0.001, 'decode_task': 'Decode'}

    fovs = [synthetic_barcodes(60000, f, nCodes, blanks, rng) for f in range(3)]
    allBarcodes = pandas.concat(fovs).reset_index(drop=True)
    # bins exactly as GenerateAdaptiveThreshold._run_analysis makes them
    areaBins = np.arange(1, 35)
    distanceBins = np.arange(0, decodeTask.parameters['distance_threshold'] + 0.02, 0.01)
    maxIntensity = np.log10(np.max([f['mean_intensity'].max() for f in fovs]))
    intensityBins = np.arange(0, 2 * maxIntensity, maxIntensity / 100)
    blankCounts = np.zeros((len(intensityBins) - 1, len(distanceBins) - 1, len(areaBins) - 1))
    codingCounts = np.zeros(blankCounts.shape)
    for barcodes in fovs:
        blankCounts += fb.GenerateAdaptiveThreshold._extract_counts(
            barcodes[barcodes['barcode_id'].isin(cb.get_blank_indexes())],
            intensityBins, distanceBins, areaBins)
        codingCounts += fb.GenerateAdaptiveThreshold._extract_counts(
            barcodes[barcodes['barcode_id'].isin(cb.get_coding_indexes())],
            intensityBins, distanceBins, areaBins)
    ds.results.update(area_bins=areaBins, distance_bins=distanceBins,
                      intensity_bins=intensityBins, blank_counts=blankCounts,
                      coding_counts=codingCounts)

    fraction = task.get_blank_fraction_histogram()
    probe = np.array([0.05, 0.2, 0.3, 0.5, 1.0, 2.5])
    with np.errstate(divide='ignore', invalid='ignore'):
        rates = np.array([task.calculate_misidentification_rate_for_threshold(t) for t in probe])
    counts = np.array([task.calculate_barcode_count_for_threshold(t) for t in probe])
    targets = np.array([0.05, 0.1, 0.2])
    thresholds = np.array([task.calculate_threshold_for_misidentification_rate(t)
                           for t in targets])
    kept = np.array([np.isin(np.arange(len(allBarcodes)),
                             task.extract_barcodes_with_threshold(t, allBarcodes).index)
                     for t in thresholds])
    # an out-of-range row set: values beyond every bin edge clip to the nearest bin
    edge = pandas.DataFrame({
        'barcode_id': np.uint16([0, 1, 135, 2, 3]), 'fov': np.uint16(0),
        'mean_intensity': np.float32([1e9, 1.0, 50.0, 10 ** intensityBins[5], 300.0]),
        'min_distance': np.float32([0.0, 0.9, 0.3, distanceBins[7], 0.2]),
        'area': np.uint16([1, 200, 3, 34, 33])})
    keptEdge = np.isin(np.arange(len(edge)),
                       task.extract_barcodes_with_threshold(thresholds[0], edge).index)
    path = os.path.join(OUT, 'fb_adaptive.npz')
    np.savez_compressed(
        path, barcode_id=allBarcodes['barcode_id'].to_numpy(),
        fov=allBarcodes['fov'].to_numpy(),
        mean_intensity=allBarcodes['mean_intensity'].to_numpy(),
        min_distance=allBarcodes['min_distance'].to_numpy(),
        area=allBarcodes['area'].to_numpy(), blanks=np.array(blanks), n_codes=nCodes,
        area_bins=areaBins, distance_bins=distanceBins, intensity_bins=intensityBins,
        blank_counts=blankCounts, coding_counts=codingCounts, fraction=fraction,
        probe=probe, rates=rates, counts=counts, targets=targets,
        thresholds=thresholds, kept=kept,
        edge_barcode_id=edge['barcode_id'].to_numpy(),
        edge_mean_intensity=edge['mean_intensity'].to_numpy(),
        edge_min_distance=edge['min_distance'].to_numpy(),
        edge_area=edge['area'].to_numpy(), edge_kept=keptEdge)
    print('fb_adaptive: {} barcodes, thresholds {}, kept {}, rates {}'.format(
        len(allBarcodes), thresholds, kept.sum(axis=1), rates))


if __name__ == '__main__':
    main()
