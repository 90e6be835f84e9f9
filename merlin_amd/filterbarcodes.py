"""Barcode filtering on the decode path's output: counterpart of
merlin/analysis/filterbarcodes.py.

* ``filter_barcodes``      FilterBarcodes._run_analysis (filterbarcodes.py:52-63)
* ``AdaptiveThreshold``    the arithmetic of GenerateAdaptiveThreshold
  (:93-330) and AdaptiveFilterBarcodes (:363-377): 3-D histograms of
  (log10 mean intensity, min distance, area) for blank and coding barcodes,
  the normalised blank-fraction histogram, the Newton search for the threshold
  that meets a misidentification rate, and the per-barcode selection.

Table arithmetic on <= 1e5 records per FOV: host work (numpy, as in the
reference).  What is new is the multi-GPU step: with FOVs sharded over ranks
every rank histograms its own FOVs and ``all_reduce`` sums the two count
histograms over ``torch.distributed`` (RCCL on GPUs, gloo on CPU) -- a true
sum all-reduce, after which every rank computes the same threshold.
"""
from typing import Optional

import numpy as np
import pandas
from scipy import optimize


def filter_barcodes(barcodeDB, fov: int, areaThreshold: int = 3,
                    intensityThreshold: float = 200,
                    distanceThreshold: float = 1e6) -> pandas.DataFrame:
    """filterbarcodes.py:33-38, 52-63: defaults of FilterBarcodes."""
    return barcodeDB.get_filtered_barcodes(
        areaThreshold, intensityThreshold,
        distanceThreshold=distanceThreshold, fov=fov)


class AdaptiveThreshold(object):

    def __init__(self, codebook, distance_threshold: float = 0.5167,
                 tolerance: float = 0.001, areaBins=None, distanceBins=None,
                 intensityBins=None):
        """Bin defaults of GenerateAdaptiveThreshold._run_analysis
        (filterbarcodes.py:255-263); ``distance_threshold`` is the decode
        task's."""
        self._codebook = codebook
        self.parameters = {'tolerance': tolerance}
        self._areaBins = np.arange(1, 35) if areaBins is None \
            else np.asarray(areaBins)
        self._distanceBins = np.arange(0, distance_threshold + 0.02, 0.01) \
            if distanceBins is None else np.asarray(distanceBins)
        self._intensityBins = None
        self._blankCounts = None
        self._codingCounts = None
        if intensityBins is not None:
            self.set_intensity_bins(intensityBins)

    # ---------------------------------------------------------------- bins --

    def set_intensity_bins(self, intensityBins) -> None:
        self._intensityBins = np.asarray(intensityBins, dtype=np.float64)
        shape = (len(self._intensityBins) - 1, len(self._distanceBins) - 1,
                 len(self._areaBins) - 1)
        self._blankCounts = np.zeros(shape)
        self._codingCounts = np.zeros(shape)

    def set_intensity_bins_from_maximum(self, maxMeanIntensity: float) -> None:
        """filterbarcodes.py:293-296: 200 bins up to twice the log10 of the
        largest mean intensity seen in the sampled FOVs.  dtype archaeology:
        the database column is float32, so the reference takes log10 in
        float32, and reference-era numpy (value-based casting) then promotes
        ``2 * x`` and ``x / 100`` to float64; both are reproduced here."""
        maxIntensity = float(np.log10(np.float32(maxMeanIntensity)))
        self.set_intensity_bins(np.arange(0, 2 * maxIntensity,
                                          maxIntensity / 100))

    def get_area_bins(self) -> np.ndarray:
        return self._areaBins

    def get_distance_bins(self) -> np.ndarray:
        return self._distanceBins

    def get_intensity_bins(self) -> np.ndarray:
        return self._intensityBins

    # -------------------------------------------------------------- counts --

    def _features(self, barcodes: pandas.DataFrame) -> np.ndarray:
        """[n,3]: log10(mean_intensity), min_distance, area in the columns'
        common dtype, like the reference's ``barcodes[[...]].values``: float32
        for tables read back from the barcode database (so the logarithm is a
        float32 one), float64 for the decoder's own frames; then widened to
        float64 (exact) for the bin search."""
        f = barcodes[['mean_intensity', 'min_distance', 'area']].to_numpy(
            copy=True)
        if f.dtype.kind != 'f':
            f = f.astype(np.float64)
        with np.errstate(divide='ignore', invalid='ignore'):
            f[:, 0] = np.log10(f[:, 0])
        return f.astype(np.float64)

    def _edges(self):
        return (self._intensityBins, self._distanceBins, self._areaBins)

    def _count_cube(self, features: np.ndarray) -> np.ndarray:
        """3-D occupancy counts with np.histogramdd's conventions (what
        _extract_counts uses, filterbarcodes.py:229-235): half-open bins, the
        last one closed on the right, points outside any axis dropped."""
        shape = tuple(len(e) - 1 for e in self._edges())
        flat = np.zeros(features.shape[0], dtype=np.int64)
        inside = np.ones(features.shape[0], dtype=bool)
        for axis, edges in enumerate(self._edges()):
            v = features[:, axis]
            k = np.searchsorted(edges, v, side='right') - 1
            k[v == edges[-1]] = shape[axis] - 1
            inside &= (k >= 0) & (k < shape[axis]) & ~np.isnan(v)
            flat = flat * shape[axis] + np.clip(k, 0, shape[axis] - 1)
        cube = np.bincount(flat[inside], minlength=int(np.prod(shape)))
        return cube.reshape(shape).astype(np.float64)

    def add_barcodes(self, barcodes: pandas.DataFrame) -> None:
        """One FOV's barcodes into the blank / coding histograms
        (filterbarcodes.py:311-324)."""
        if self._intensityBins is None:
            raise ValueError('intensity bins are not set')
        if len(barcodes) == 0:
            return
        ids = barcodes['barcode_id'].to_numpy()
        feats = self._features(barcodes)
        blank = np.isin(ids, self._codebook.get_blank_indexes())
        coding = np.isin(ids, self._codebook.get_coding_indexes())
        self._blankCounts += self._count_cube(feats[blank])
        self._codingCounts += self._count_cube(feats[coding])

    def all_reduce(self, group=None, device=None) -> None:
        """Sum the count histograms over the ranks of torch.distributed (each
        rank has added its own FOVs).  Counts are integers held in float64, so
        the sum is exact in any order."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        both = torch.from_numpy(np.stack([self._blankCounts,
                                          self._codingCounts]))
        if device is not None:
            both = both.to(device)
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
        both = both.cpu().numpy()
        self._blankCounts, self._codingCounts = both[0].copy(), both[1].copy()

    def get_blank_count_histogram(self) -> np.ndarray:
        return self._blankCounts

    def get_coding_count_histogram(self) -> np.ndarray:
        return self._codingCounts

    def get_total_count_histogram(self) -> np.ndarray:
        return self._blankCounts + self._codingCounts

    # ---------------------------------------------------------- thresholds --

    def _class_sizes(self):
        return (len(self._codebook.get_blank_indexes()),
                len(self._codebook.get_coding_indexes()))

    def get_blank_fraction_histogram(self) -> np.ndarray:
        """Blank count over CODING count per bin (float max where no coding
        barcode fell), divided by the blank share of the codebook
        (filterbarcodes.py:113-140)."""
        nBlank, nCoding = self._class_sizes()
        coding = self._codingCounts
        ratio = np.full(coding.shape, np.finfo(np.float64).max)
        np.divide(self._blankCounts, coding, out=ratio, where=coding != 0)
        with np.errstate(over='ignore'):
            return ratio / (nBlank / (nBlank + nCoding))

    def _selected_counts(self, threshold: float):
        keep = self.get_blank_fraction_histogram() < threshold
        return (np.sum(self._blankCounts[keep]),
                np.sum(self._codingCounts[keep]))

    def calculate_misidentification_rate_for_threshold(
            self, threshold: float) -> float:
        """Blank barcodes per blank codeword over coding barcodes per coding
        codeword among the bins below the threshold (filterbarcodes.py:142-168)."""
        nBlank, nCoding = self._class_sizes()
        blank, coding = self._selected_counts(threshold)
        with np.errstate(divide='ignore', invalid='ignore'):
            return (blank / nBlank) / (coding / nCoding)

    def calculate_threshold_for_misidentification_rate(
            self, targetMisidentificationRate: float) -> float:
        """Secant search started at 0.2 / 0.3 like the reference's
        scipy.optimize.newton call (filterbarcodes.py:170-188)."""
        return optimize.newton(
            lambda x, target:
                self.calculate_misidentification_rate_for_threshold(x) - target,
            0.2, args=[targetMisidentificationRate],
            tol=self.parameters['tolerance'], x1=0.3, disp=False)

    def calculate_barcode_count_for_threshold(self, threshold: float) -> float:
        """filterbarcodes.py:190-203"""
        blank, coding = self._selected_counts(threshold)
        return blank + coding

    def extract_barcodes_with_threshold(self, blankThreshold: float,
                                        barcodeSet: pandas.DataFrame
                                        ) -> pandas.DataFrame:
        """Keep the barcodes whose (intensity, distance, area) bin has a blank
        fraction below the threshold; values outside the bins clip to the
        nearest bin; intensity and distance bins are closed on the right, area
        bins on the left (np.digitize right=True / False,
        filterbarcodes.py:205-227)."""
        feats = self._features(barcodeSet)
        fraction = self.get_blank_fraction_histogram()
        sides = ('left', 'left', 'right')
        index = []
        for axis, (edges, side) in enumerate(zip(self._edges(), sides)):
            k = np.searchsorted(edges, feats[:, axis], side=side) - 1
            index.append(np.clip(k, 0, fraction.shape[axis] - 1))
        keep = fraction[index[0], index[1], index[2]] < blankThreshold
        return barcodeSet[keep]

    def adaptive_filter(self, barcodes: pandas.DataFrame,
                        misidentification_rate: float = 0.05,
                        threshold: Optional[float] = None) -> pandas.DataFrame:
        """AdaptiveFilterBarcodes._run_analysis (filterbarcodes.py:363-377)."""
        if threshold is None:
            threshold = self.calculate_threshold_for_misidentification_rate(
                misidentification_rate)
        return self.extract_barcodes_with_threshold(threshold, barcodes)
