"""First-order statistics: interface and feature-name surface of the reference's radiomics/firstorder.py
(RadiomicsFirstOrder).  The reference evaluates these with nan-aware numpy reductions over the ROI intensities
(segment mode, firstorder.py:96-101) or over every kernel window (voxel mode, :104-118); here the reductions run on
the MI355X -- prad_firstorder_dev (compaction, rocPRIM sort, block reductions) and prad_voxel_firstorder_dev (one
wave per kernel) -- and this class only derives the 19 feature values from the returned statistics.

Settings as in the reference: voxelArrayShift [0] is added to the intensities in Energy, TotalEnergy and
RootMeanSquared (:151,:166,:364)."""
from __future__ import annotations

import numpy as np

from .base import RadiomicsFeaturesBase, deprecated, _ENQUEUE_DEFAULT
from .image import as_array


class RadiomicsFirstOrder(RadiomicsFeaturesBase):
    def __init__(self, inputImage, inputMask, **kwargs):
        super().__init__(inputImage, inputMask, **kwargs)
        self.pixelSpacing = self.inputImage.GetSpacing()
        self.voxelArrayShift = kwargs.get("voxelArrayShift", 0)
        self.rawImageArray = self.imageArray
        self.discretizedImageArray = self._applyBinning(self.imageArray)
        self.st = None

    def _initVoxelBasedCalculation(self):
        super()._initVoxelBasedCalculation()
        kernelRadius = self.settings.get("kernelRadius", 1)
        # firstorder.py:45-58: kernel offsets are limited by the ROI (masked kernels) or image extent
        if self.masked:
            size = np.max(self.allLabelledVoxelCoordinates, 1) - np.min(self.allLabelledVoxelCoordinates, 1) + 1
        else:
            size = np.array(self.imageArray.shape)
        self.boundingBoxSize = np.minimum(size, kernelRadius * 2 + 1)

    def _calculateFeatures(self, voxelCoordinates=None):
        if not self.voxelBased:
            yield from super()._calculateFeatures(voxelCoordinates)
            return
        # voxel mode: the operator backend evaluates the enabled features kernel by kernel
        names = [n for n, on in self.enabledFeatures.items() if on]
        vals = self.cMatrices.voxel_firstorder(
            self.rawImageArray, self.maskArray, self.discretizedImageArray, voxelCoordinates,
            self.settings.get("kernelRadius", 1), self.boundingBoxSize, self.settings.get("force2D", False),
            self.settings.get("force2Ddimension", 0), self.voxelArrayShift, float(np.multiply.reduce(self.pixelSpacing)),
            names)
        for n in names:
            yield True, n, vals[n]

    def enqueue(self):
        """case pipeline (base.enqueue): the statistics' passes are queued now, _initCalculation collects them"""
        self._queuedStats = None
        fn = getattr(self.cMatrices, "firstorder_stats_enqueue", None)
        if (self.voxelBased or not self.deviceResident or fn is None
                or not self.settings.get("enqueueSegment", _ENQUEUE_DEFAULT) or "Ns" not in self.coefficients):
            return False
        try:
            self._queuedStats = fn(self.rawImageArray, self.maskArray, self.coefficients["Ns"], self.voxelArrayShift)
        except NotImplementedError:
            return False
        return True

    def dropEnqueued(self):
        self._queuedStats = None

    def imageRequest(self):
        if (self.voxelBased or not self.deviceResident or not hasattr(self.cMatrices, "segment_image_enqueue")
                or not self.settings.get("enqueueSegment", _ENQUEUE_DEFAULT) or "Ns" not in self.coefficients):
            return None
        import torch
        raw = self.rawImageArray
        if raw.dtype in (torch.int16, torch.int32) or self.coefficients["Ns"] < (1 << 20):
            return None         # (exact-histogram / full-sort routes of prad_firstorder_dev: synchronous)
        return "firstorder", {"raw": raw, "shift": self.voxelArrayShift, "features": None}

    def takeEnqueued(self, finish):
        self._queuedStats = finish

    def _initCalculation(self, voxelCoordinates=None):
        queued, self._queuedStats = getattr(self, "_queuedStats", None), None
        st = queued() if queued is not None else self.cMatrices.firstorder_stats(self.rawImageArray, self.maskArray,
                                                                                  self.voxelArrayShift)
        self.st = {k: np.array([v], dtype=np.float64) for k, v in st.items()}
        counts = self.coefficients["levelCounts"]           # ROI voxels per present grey level, ascending (:99-101)
        p_i = counts.reshape((1, -1)).astype("float")
        total = np.sum(p_i, 1, keepdims=True)
        total[total == 0] = 1
        self.coefficients["p_i"] = p_i / total

    # -- the 19 features (firstorder.py:147-474) -----------------------------------------------------------
    def getEnergyFeatureValue(self):
        """Σ (X(i) + c)²  with c = voxelArrayShift  (firstorder.py:135)"""
        return self.st["Energy"]

    def getTotalEnergyFeatureValue(self):
        """Vvoxel · Σ (X(i) + c)²  (voxel volume in mm³ times Energy)  (firstorder.py:157)"""
        return self.st["Energy"] * np.multiply.reduce(self.pixelSpacing)

    def getEntropyFeatureValue(self):
        """−Σ p(i) log2(p(i) + ε) over the discretised intensity histogram (binWidth / binCount)  (firstorder.py:181)"""
        p_i = self.coefficients["p_i"]
        return -1.0 * np.sum(p_i * np.log2(p_i + np.spacing(1)), 1)

    def getMinimumFeatureValue(self):
        """min(X)  (firstorder.py:201)"""
        return self.st["Minimum"]

    def get10PercentileFeatureValue(self):
        """10th percentile of X (numpy's linear interpolation between order statistics)  (firstorder.py:211)"""
        return self.st["P10"]

    def get90PercentileFeatureValue(self):
        """90th percentile of X (numpy's linear interpolation between order statistics)  (firstorder.py:219)"""
        return self.st["P90"]

    def getMaximumFeatureValue(self):
        """max(X)  (firstorder.py:228)"""
        return self.st["Maximum"]

    def getMeanFeatureValue(self):
        """1/Np · Σ X(i)  (firstorder.py:240)"""
        return self.st["Mean"]

    def getMedianFeatureValue(self):
        """median of X  (firstorder.py:252)"""
        return self.st["Median"]

    def getInterquartileRangeFeatureValue(self):
        """P75 − P25  (firstorder.py:261)"""
        return self.st["P75"] - self.st["P25"]

    def getRangeFeatureValue(self):
        """max(X) − min(X)  (firstorder.py:276)"""
        return self.st["Maximum"] - self.st["Minimum"]

    def getMeanAbsoluteDeviationFeatureValue(self):
        """1/Np · Σ |X(i) − mean(X)|  (firstorder.py:288)"""
        return self.st["MAD"]

    def getRobustMeanAbsoluteDeviationFeatureValue(self):
        """mean absolute deviation of the voxels with P10 ≤ X(i) ≤ P90 from their own mean  (firstorder.py:301)"""
        return self.st["rMAD"]

    def getRootMeanSquaredFeatureValue(self):
        """sqrt(1/Np · Σ (X(i) + c)²)  (firstorder.py:334)"""
        if self.st["Np"][0] == 0:       # firstorder.py:360-362
            return 0
        return np.sqrt(self.st["Energy"] / self.st["Np"])

    @deprecated
    def getStandardDeviationFeatureValue(self):
        """sqrt(Variance) (population standard deviation; not enabled by default: correlated with Variance)  (firstorder.py:359)"""
        return np.sqrt(self.st["m2"])

    def _m2_safe(self):
        m2 = self.st["m2"].copy()
        m2[m2 == 0] = 1                 # flat region: the moment ratio is returned as 0 (:403-405, :441-443)
        return m2

    def getSkewnessFeatureValue(self):
        """μ3 / σ³ with central moments μk = 1/Np · Σ (X(i) − mean)^k; 0 for a flat region  (firstorder.py:379)"""
        return self.st["m3"] / self._m2_safe() ** 1.5

    def getKurtosisFeatureValue(self):
        """μ4 / σ⁴ (not excess kurtosis: a normal distribution gives 3); 0 for a flat region  (firstorder.py:410)"""
        return self.st["m4"] / self._m2_safe() ** 2.0

    def getVarianceFeatureValue(self):
        """1/Np · Σ (X(i) − mean)²  (firstorder.py:446)"""
        return np.sqrt(self.st["m2"]) ** 2      # np.nanstd(x) ** 2 (:462)

    def getUniformityFeatureValue(self):
        """Σ p(i)² over the discretised intensity histogram  (firstorder.py:459)"""
        return np.nansum(self.coefficients["p_i"] ** 2, 1)
