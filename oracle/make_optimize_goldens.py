"""Golden OptimizeIteration trace from the REFERENCE's own
``merlin/analysis/optimize.py`` (SURVEY 8c's G6).

Run in the build container only (the reference does not travel):

    /opt/conda/bin/python3.9 oracle/make_optimize_goldens.py

``optimize.py`` and ``merlin/util/decoding.py`` are imported unmodified.  What
``optimize.py`` subclasses and calls around its arithmetic -- MERlin's task
base class, data set, preprocess task and barcode database (PyTables, dotenv,
file storage: out of scope and not importable here) -- is replaced by small
in-memory stand-ins, exactly as ``make_filter_goldens.py`` does for the
filtering task; the shims for ``cv2`` (absent: the oracle's blur restatement),
``np.int`` and two pandas spellings are those of ``make_goldens.py``.

Two chained iterations (``previous_iteration``), each with its own random
fragment draw made by the reference's constructor (``np.random.choice``,
optimize.py:43-50, global seed set here), run through the reference's
``_run_analysis`` for every fragment, then its ``get_scale_factors``,
``get_backgrounds``, ``get_scale_factor_history`` and
``get_barcode_count_history``; once with ``optimize_background`` off and once
on.  One of the FOVs is pure background (nothing decodes: NaN refactors), and
the last bit is never seen, so its refactor is zero in every fragment (the
``refactors == 0 -> 1`` rule of optimize.py:333).

Output: ``tests/golden/optimize_trace_reference.npz`` (data only).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402  (shims, test codebook, synthetic stacks)


def install_stubs():
    mg.install_shims()           # merlin namespace, cv2 stub, np.int, pandas

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m
    pkg('merlin.analysis', os.path.join(REF, 'merlin', 'analysis'))
    dec = types.ModuleType('merlin.analysis.decode')

    class BarcodeSavingParallelAnalysisTask(object):
        """Stand-in for analysis/decode.py:14-33 + core/analysistask.py: keeps
        the constructor's attributes and hands out the in-memory database."""

        def __init__(self, dataSet, parameters=None, analysisName=None):
            self.dataSet = dataSet
            self.parameters = {} if parameters is None else parameters
            self.analysisName = analysisName
            self._done = False

        def is_complete(self):
            return self._done

        def get_barcode_database(self):
            return self.dataSet.barcodeDB

    dec.BarcodeSavingParallelAnalysisTask = BarcodeSavingParallelAnalysisTask
    sys.modules['merlin.analysis.decode'] = dec
    sys.modules['merlin.analysis'].decode = dec


class FakeCodebook(mg.ArrayCodebook):
    def get_barcode_count(self):
        return self._b.shape[0]

    def get_bit_count(self):
        return self._b.shape[1]

    def get_bit_names(self):
        return ['bit{}'.format(i + 1) for i in range(self._b.shape[1])]


class FakeDataOrganization:
    def get_data_channel_for_bit(self, name):
        return int(name[3:]) - 1

    def get_data_channel_color(self, channel):
        return '750' if channel % 2 else '650'


class FakePreprocess:
    def __init__(self, codebook, images):
        self._cb, self._images = codebook, images

    def get_codebook(self):
        return self._cb

    def get_processed_image_set(self, fov, zIndex=None, chromaticCorrector=None):
        return self._images[fov][:, zIndex]

    def get_pixel_histogram(self):
        # analysis/preprocess.py:23-33, 110-131: per-FOV histograms, summed
        B = self._images[0].shape[0]
        hist = np.zeros((B, 65534))
        for img in self._images:
            for b in range(B):
                hist[b] += np.histogram(img[b], bins=np.arange(0, 65535, 1))[0]
        return hist


class FakeDB:
    def __init__(self):
        self.written = {}

    def write_barcodes(self, frame, fov=None):
        self.written[fov] = frame


class FakeLogger:
    def info(self, *a):
        pass


class FakeDataSet:
    def __init__(self, fovs, zCount):
        self._fovs, self._z = list(fovs), zCount
        self.results = {}
        self.tasks = {}
        self.barcodeDB = FakeDB()

    def get_fovs(self):
        return self._fovs

    def get_z_positions(self):
        return list(np.arange(self._z) * 1.5)

    def get_logger(self, task):
        return FakeLogger()

    def get_data_organization(self):
        return FakeDataOrganization()

    def load_analysis_task(self, name):
        return self.tasks[name]

    def save_numpy_analysis_result(self, arr, name, analysisName, resultIndex=None):
        self.results[(analysisName, name, resultIndex)] = np.array(arr)

    def load_numpy_analysis_result(self, name, analysisName, resultIndex=None):
        key = (analysisName, name, resultIndex)
        if key not in self.results:
            raise FileNotFoundError(str(key))
        return self.results[key]

    def save_pickle_analysis_result(self, *a, **k):
        pass


def run_chain(optimize, codebook, images, zCount, F, background, seed):
    ds = FakeDataSet(range(len(images)), zCount)
    ds.tasks['Pre'] = FakePreprocess(FakeCodebook(codebook), images)
    np.random.seed(seed)
    out = {}
    prev = None
    for k, name in enumerate(('Optimize1', 'Optimize2')):
        params = {'preprocess_task': 'Pre', 'warp_task': 'Warp',
                  'fov_per_iteration': F, 'optimize_background': background}
        if prev is not None:
            params['previous_iteration'] = prev
        task = optimize.OptimizeIteration(ds, params, name)
        ds.tasks[name] = task
        for i in range(task.fragment_count()):
            task._run_analysis(i)
        task._done = True
        tag = 'it{}_'.format(k)
        out[tag + 'fov_index'] = np.array(task.parameters['fov_index'])
        out[tag + 'scale_factors'] = task.get_scale_factors()
        out[tag + 'backgrounds'] = task.get_backgrounds()
        for res in ('scale_refactors', 'background_refactors', 'barcode_counts',
                    'previous_scale_factors', 'previous_backgrounds'):
            out[tag + res] = np.array([
                ds.load_numpy_analysis_result(res, name, resultIndex=i)
                for i in range(F)])
        out[tag + 'scale_factor_history'] = task.get_scale_factor_history()
        out[tag + 'barcode_count_history'] = task.get_barcode_count_history()
        out[tag + 'barcodes_written'] = np.array(
            [len(ds.barcodeDB.written[i]) for i in range(F)])
        prev = name
    return out


def main():
    install_stubs()
    from merlin.analysis import optimize
    codebook = mg.load_test_codebook()
    Z, H, W = 2, 64, 96
    # spots only from codewords that do not use the last bit: no barcode with
    # that bit is ever seen, its scale refactor is 0 in every fragment and the
    # `refactors == 0 -> 1` rule (optimize.py:333) must leave its factor alone.
    # The bit's image still carries the ordinary background.
    sub = codebook[codebook[:, -1] == 0]
    images = [mg.synthetic_stack(sub, (Z, H, W), 60, seed=500 + f,
                                 amp=(1500, 6000)) for f in range(3)]
    # a field of view with nothing in it: a fragment drawn from it sees no
    # barcode and returns NaN refactors
    rng = np.random.default_rng(9)
    images.append(rng.poisson(100, size=images[0].shape).astype(np.uint16))
    out = {'codebook': codebook, 'images': np.array(images), 'seed': 4, 'F': 6}
    for name, bgopt in (('plain_', False), ('bg_', True)):
        res = run_chain(optimize, codebook, images, Z, 6, bgopt, 4)
        out.update({name + k: v for k, v in res.items()})
    path = os.path.join(OUT, 'optimize_trace_reference.npz')
    np.savez_compressed(path, **out)
    for name in ('plain_', 'bg_'):
        print(name, 'draw', out[name + 'it0_fov_index'].tolist(),
              out[name + 'it1_fov_index'].tolist())
        print('  sf it0', np.round(out[name + 'it0_scale_factors'], 3))
        print('  sf it1', np.round(out[name + 'it1_scale_factors'], 3))
        print('  bg it1', np.round(out[name + 'it1_backgrounds'], 3))
        print('  refactors it0 nan rows', np.isnan(out[name + 'it0_scale_refactors']).all(axis=1),
              'zeros', int((out[name + 'it0_scale_refactors'] == 0).sum()))
        print('  barcodes written', out[name + 'it0_barcodes_written'], out[name + 'it1_barcodes_written'])
    print('wrote', path, os.path.getsize(path))


if __name__ == '__main__':
    main()
