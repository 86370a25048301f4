"""Grey Level Dependence Matrix features: interface and feature-name surface of the reference's radiomics/gldm.py
(RadiomicsGLDM), matrix built on the MI355X through cMatrices.calculate_gldm.

P has shape (Nvox, Ngp, Nd'): absent grey levels dropped, dependence counts that never occur dropped
(gldm.py:103-136); jvector = dependence count + 1; Nz = number of voxels (forced to 1 when empty)."""
from __future__ import annotations

import numpy as np

from .base import deprecated
from .glszm import _ZoneLikeFeatures


class RadiomicsGLDM(_ZoneLikeFeatures):
    def __init__(self, inputImage, inputMask, **kwargs):
        super().__init__(inputImage, inputMask, **kwargs)
        self.gldm_a = kwargs.get("gldm_a", 0)
        self.P_gldm = None
        self.imageArray = self._applyBinning(self.imageArray)

    def _P(self):
        return self.P_gldm

    def _segmentRoute(self):
        return ("gldm", {"alpha": self.gldm_a})

    def _calculateFeatures(self, voxelCoordinates=None):
        fused = self._fusedVoxelFeatures("gldm", voxelCoordinates, self.gldm_a)
        if fused is None:
            fused = self._fusedSegmentFeatures("gldm", alpha=self.gldm_a)
        if fused is not None:
            yield from fused
            return
        yield from super()._calculateFeatures(voxelCoordinates)

    def _initCalculation(self, voxelCoordinates=None):
        self.P_gldm = self._calculateMatrix(voxelCoordinates)

    def _calculateMatrix(self, voxelCoordinates=None):
        Ng = self.coefficients["Ng"]
        args = [self.imageArray, self.maskArray, np.array(self.settings.get("distances", [1])), Ng, self.gldm_a,
                self.settings.get("force2D", False), self.settings.get("force2Ddimension", 0)]
        P = self.cMatrices.calculate_gldm(*(args + self._matrix_tail(voxelCoordinates)))
        P = np.delete(P, self._absent_levels(), 1)
        j = np.arange(1, P.shape[2] + 1, dtype="float64")
        pd = np.sum(P, 1)
        pg = np.sum(P, 2)
        unused = np.where(np.sum(pd, 0) == 0)
        P = np.delete(P, unused, 2)
        pd = np.delete(pd, unused, 1)
        Nz = np.sum(pd, 1)
        Nz[Nz == 0] = 1
        c = self.coefficients
        c["Nz"] = Nz
        c["pd"] = pd
        c["ps"] = pd            # alias used by the shared helpers
        c["pg"] = pg
        c["ivector"] = c["grayLevels"].astype(float)
        c["jvector"] = np.delete(j, unused)
        return P

    def getSmallDependenceEmphasisFeatureValue(self):
        """Σij P(i,j) / j² / Nz  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:138)"""
        c = self.coefficients
        return np.sum(c["pd"] / (c["jvector"][None, :] ** 2), 1) / c["Nz"]

    def getLargeDependenceEmphasisFeatureValue(self):
        """Σij P(i,j) j² / Nz  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:154)"""
        return self._over_sizes(self.coefficients["jvector"] ** 2)

    def getGrayLevelNonUniformityFeatureValue(self):
        """Σi (Σj P(i,j))² / Nz  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:170)"""
        c = self.coefficients
        return np.sum(c["pg"] ** 2, 1) / c["Nz"]

    @deprecated
    def getGrayLevelNonUniformityNormalizedFeatureValue(self):
        """deprecated: equal to first order Uniformity  (gldm.py:186)"""
        raise DeprecationWarning("GLDM - Gray Level Non-Uniformity Normalized is mathematically equal to "
                                 "First Order - Uniformity")

    def getDependenceNonUniformityFeatureValue(self):
        """Σj (Σi P(i,j))² / Nz  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:207)"""
        c = self.coefficients
        return np.sum(c["pd"] ** 2, 1) / c["Nz"]

    def getDependenceNonUniformityNormalizedFeatureValue(self):
        """Σj (Σi P(i,j))² / Nz²  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:222)"""
        c = self.coefficients
        return np.sum(c["pd"] ** 2, 1) / c["Nz"] ** 2

    def getGrayLevelVarianceFeatureValue(self):
        """Σij p(i,j) (i − μ)² with μ = Σij p(i,j) i  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:237)"""
        return self._level_variance()

    def getDependenceVarianceFeatureValue(self):
        """Σij p(i,j) (j − μ)² with μ = Σij p(i,j) j  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:256)"""
        return self._size_variance()

    def getDependenceEntropyFeatureValue(self):
        """−Σij p(i,j) log2(p(i,j) + ε)  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:275)"""
        return self._entropy()

    @deprecated
    def getDependencePercentageFeatureValue(self):
        """deprecated: always 1 (every ROI voxel has a dependence zone)  (gldm.py:291)"""
        raise DeprecationWarning("GLDM - Dependence Percentage always computes 1")

    def getLowGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) / i² / Nz  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:310)"""
        c = self.coefficients
        return np.sum(c["pg"] / (c["ivector"][None, :] ** 2), 1) / c["Nz"]

    def getHighGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) i² / Nz  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:326)"""
        return self._over_levels(self.coefficients["ivector"] ** 2)

    def getSmallDependenceLowGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) / (i² j²) / Nz  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:342)"""
        return np.sum(self.P_gldm / ((self._iw() ** 2) * (self._jw() ** 2)), (1, 2)) / self.coefficients["Nz"]

    def getSmallDependenceHighGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) i² / j² / Nz  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:364)"""
        return np.sum(self.P_gldm * (self._iw() ** 2) / (self._jw() ** 2), (1, 2)) / self.coefficients["Nz"]

    def getLargeDependenceLowGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) j² / i² / Nz  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:387)"""
        return np.sum(self.P_gldm * (self._jw() ** 2) / (self._iw() ** 2), (1, 2)) / self.coefficients["Nz"]

    def getLargeDependenceHighGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) i² j² / Nz  [j = dependence size; Nz = Σ P(i,j) = Np]  (gldm.py:410)"""
        return np.sum(self.P_gldm * ((self._jw() ** 2) * (self._iw() ** 2)), (1, 2)) / self.coefficients["Nz"]
