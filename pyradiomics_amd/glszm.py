"""Grey Level Size Zone Matrix features: interface and feature-name surface of the reference's radiomics/glszm.py
(RadiomicsGLSZM), matrix built on the MI355X through cMatrices.calculate_glszm.

P has shape (Nvox, Ngp, Ns'): absent grey levels dropped (glszm.py:99-106), zone sizes that never occur dropped
(:123-126).  Nz = zones, Np = voxels in zones, both forced to 1 when empty (:117-121).  No angle axis."""
from __future__ import annotations

import numpy as np

from .base import RadiomicsFeaturesBase

_EPS = np.spacing(1)


class _ZoneLikeFeatures(RadiomicsFeaturesBase):
    """the feature pattern shared by GLSZM and GLDM: a (Nvox, level, size) count matrix P with marginals
    pg (per level) and ps (per size / dependence), Nz = sum(P)"""

    def _P(self):
        raise NotImplementedError

    def _iw(self):
        return self.coefficients["ivector"][None, :, None]

    def _jw(self):
        return self.coefficients["jvector"][None, None, :]

    def _over_sizes(self, weight):      # sum_j ps * weight / Nz
        c = self.coefficients
        return np.sum(c["ps"] * weight[None, :], 1) / c["Nz"]

    def _over_levels(self, weight):
        c = self.coefficients
        return np.sum(c["pg"] * weight[None, :], 1) / c["Nz"]

    def _level_variance(self):
        c = self.coefficients
        i = c["ivector"][None, :]
        pg = c["pg"] / c["Nz"][:, None]
        u = np.sum(pg * i, 1, keepdims=True)
        return np.sum(pg * (i - u) ** 2, 1)

    def _size_variance(self):
        c = self.coefficients
        j = c["jvector"][None, :]
        ps = c["ps"] / c["Nz"][:, None]
        u = np.sum(ps * j, 1, keepdims=True)
        return np.sum(ps * (j - u) ** 2, 1)

    def _entropy(self):
        p = self._P() / self.coefficients["Nz"][:, None, None]
        return -np.sum(p * np.log2(p + _EPS), (1, 2))


class RadiomicsGLSZM(_ZoneLikeFeatures):
    def __init__(self, inputImage, inputMask, **kwargs):
        super().__init__(inputImage, inputMask, **kwargs)
        self.P_glszm = None
        self.imageArray = self._applyBinning(self.imageArray)

    def _P(self):
        return self.P_glszm

    def _segmentRoute(self):
        return ("glszm", {})

    def _calculateFeatures(self, voxelCoordinates=None):
        fused = self._fusedVoxelFeatures("glszm", voxelCoordinates)
        if fused is None:
            fused = self._fusedSegmentFeatures("glszm")
        if fused is not None:
            yield from fused
            return
        yield from super()._calculateFeatures(voxelCoordinates)

    def _initCalculation(self, voxelCoordinates=None):
        self.P_glszm = self._calculateMatrix(voxelCoordinates)
        self._calculateCoefficients()

    def _calculateMatrix(self, voxelCoordinates=None):
        Ng = self.coefficients["Ng"]
        Ns = self.coefficients["Ns"] if self.deviceResident else np.sum(self.maskArray)
        args = [self.imageArray, self.maskArray, Ng, Ns, self.settings.get("force2D", False),
                self.settings.get("force2Ddimension", 0)]
        self._sizes = None
        compact = getattr(self.cMatrices, "calculate_glszm_compact", None)
        if compact is not None and not self.voxelBased and self.settings.get("compactGLSZM", True):
            # only the non-empty size columns (what _calculateCoefficients keeps anyway), straight from the device
            P, self._sizes = compact(*args)
        else:
            P = self.cMatrices.calculate_glszm(*(args + self._matrix_tail(voxelCoordinates)))
        return np.delete(P, self._absent_levels(), 1)

    def _calculateCoefficients(self):
        P = self.P_glszm
        ps = np.sum(P, 1)
        pg = np.sum(P, 2)
        j = np.arange(1, P.shape[2] + 1, dtype=np.float64) if self._sizes is None else self._sizes.astype(np.float64)
        Nz = np.sum(P, (1, 2))
        Nz[Nz == 0] = 1
        Np = np.sum(ps * j[None, :], 1)
        Np[Np == 0] = 1
        unused = np.where(np.sum(ps, 0) == 0)
        self.P_glszm = np.delete(P, unused, 2)
        c = self.coefficients
        c["Np"], c["Nz"] = Np, Nz
        c["ps"] = np.delete(ps, unused, 1)
        c["pg"] = pg
        c["ivector"] = c["grayLevels"].astype(float)
        c["jvector"] = np.delete(j, unused)

    def getSmallAreaEmphasisFeatureValue(self):
        """Σij P(i,j) / j² / Nz  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:140)"""
        c = self.coefficients
        return np.sum(c["ps"] / (c["jvector"][None, :] ** 2), 1) / c["Nz"]

    def getLargeAreaEmphasisFeatureValue(self):
        """Σij P(i,j) j² / Nz  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:156)"""
        return self._over_sizes(self.coefficients["jvector"] ** 2)

    def getGrayLevelNonUniformityFeatureValue(self):
        """Σi (Σj P(i,j))² / Nz  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:172)"""
        c = self.coefficients
        return np.sum(c["pg"] ** 2, 1) / c["Nz"]

    def getGrayLevelNonUniformityNormalizedFeatureValue(self):
        """Σi (Σj P(i,j))² / Nz²  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:187)"""
        c = self.coefficients
        return np.sum(c["pg"] ** 2, 1) / c["Nz"] ** 2

    def getSizeZoneNonUniformityFeatureValue(self):
        """Σj (Σi P(i,j))² / Nz  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:202)"""
        c = self.coefficients
        return np.sum(c["ps"] ** 2, 1) / c["Nz"]

    def getSizeZoneNonUniformityNormalizedFeatureValue(self):
        """Σj (Σi P(i,j))² / Nz²  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:217)"""
        c = self.coefficients
        return np.sum(c["ps"] ** 2, 1) / c["Nz"] ** 2

    def getZonePercentageFeatureValue(self):
        """Nz / Np  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:232)"""
        return self.coefficients["Nz"] / self.coefficients["Np"]

    def getGrayLevelVarianceFeatureValue(self):
        """Σij p(i,j) (i − μ)² with μ = Σij p(i,j) i  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:249)"""
        return self._level_variance()

    def getZoneVarianceFeatureValue(self):
        """Σij p(i,j) (j − μ)² with μ = Σij p(i,j) j  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:269)"""
        return self._size_variance()

    def getZoneEntropyFeatureValue(self):
        """−Σij p(i,j) log2(p(i,j) + ε)  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:289)"""
        return self._entropy()

    def getLowGrayLevelZoneEmphasisFeatureValue(self):
        """Σij P(i,j) / i² / Nz  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:309)"""
        c = self.coefficients
        return np.sum(c["pg"] / (c["ivector"][None, :] ** 2), 1) / c["Nz"]

    def getHighGrayLevelZoneEmphasisFeatureValue(self):
        """Σij P(i,j) i² / Nz  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:325)"""
        return self._over_levels(self.coefficients["ivector"] ** 2)

    def getSmallAreaLowGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) / (i² j²) / Nz  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:341)"""
        return np.sum(self.P_glszm / ((self._iw() ** 2) * (self._jw() ** 2)), (1, 2)) / self.coefficients["Nz"]

    def getSmallAreaHighGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) i² / j² / Nz  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:364)"""
        return np.sum(self.P_glszm * (self._iw() ** 2) / (self._jw() ** 2), (1, 2)) / self.coefficients["Nz"]

    def getLargeAreaLowGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) j² / i² / Nz  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:388)"""
        return np.sum(self.P_glszm * (self._jw() ** 2) / (self._iw() ** 2), (1, 2)) / self.coefficients["Nz"]

    def getLargeAreaHighGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) i² j² / Nz  [j = zone size; Nz = Σ P(i,j)]  (glszm.py:412)"""
        return np.sum(self.P_glszm * (self._iw() ** 2) * (self._jw() ** 2), (1, 2)) / self.coefficients["Nz"]
