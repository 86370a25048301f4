"""Neighbouring Grey Tone Difference Matrix features: interface and feature-name surface of the reference's
radiomics/ngtdm.py (RadiomicsNGTDM), matrix built on the MI355X through cMatrices.calculate_ngtdm.

P has shape (Nvox, Ngp, 3) = (n_i, s_i, level) with rows of absent levels dropped (ngtdm.py:112-114);
p_i = n_i / Nvp (:116-132).  Zero-division fallbacks follow :149-150, :187-188, :219-220, :247-249, :284-285."""
from __future__ import annotations

import numpy as np

from .base import RadiomicsFeaturesBase


class RadiomicsNGTDM(RadiomicsFeaturesBase):
    def __init__(self, inputImage, inputMask, **kwargs):
        super().__init__(inputImage, inputMask, **kwargs)
        self.P_ngtdm = None
        self.imageArray = self._applyBinning(self.imageArray)

    def _segmentRoute(self):
        return ("ngtdm", {})

    def _calculateFeatures(self, voxelCoordinates=None):
        fused = self._fusedVoxelFeatures("ngtdm", voxelCoordinates)
        if fused is None:
            fused = self._fusedSegmentFeatures("ngtdm")
        if fused is not None:
            yield from fused
            return
        yield from super()._calculateFeatures(voxelCoordinates)

    def _initCalculation(self, voxelCoordinates=None):
        self.P_ngtdm = self._calculateMatrix(voxelCoordinates)
        self._calculateCoefficients()

    def _calculateMatrix(self, voxelCoordinates=None):
        args = [self.imageArray, self.maskArray, np.array(self.settings.get("distances", [1])),
                self.coefficients["Ng"], self.settings.get("force2D", False),
                self.settings.get("force2Ddimension", 0)]
        P = self.cMatrices.calculate_ngtdm(*(args + self._matrix_tail(voxelCoordinates)))
        return np.delete(P, np.where(np.sum(P[:, :, 0], 0) == 0), 1)

    def _calculateCoefficients(self):
        n = self.P_ngtdm[:, :, 0]
        c = self.coefficients
        c["Nvp"] = np.sum(n, 1)
        c["p_i"] = n / c["Nvp"][:, None]
        c["s_i"] = self.P_ngtdm[:, :, 1]
        c["ivector"] = self.P_ngtdm[:, :, 2]
        c["Ngp"] = np.sum(n > 0, 1)
        c["p_zero"] = np.where(c["p_i"] == 0)

    def _pairs(self, f):
        """f(a_i, a_j) evaluated on all level pairs, zeroed where either level is absent in that kernel"""
        z = self.coefficients["p_zero"]
        f[z[0], :, z[1]] = 0
        f[z[0], z[1], :] = 0
        return f

    def getCoarsenessFeatureValue(self):
        """1 / Σi p(i) s(i); 10⁶ where the sum is 0 (completely homogeneous region)  (ngtdm.py:133)"""
        c = self.coefficients
        s = np.sum(c["p_i"] * c["s_i"], 1)
        s[s != 0] = 1 / s[s != 0]
        s[s == 0] = 1e6
        return s

    def getContrastFeatureValue(self):
        """(1 / (Ngp (Ngp − 1)) · Σij p(i) p(j) (i − j)²) · (1 / Nvp · Σi s(i)); 0 for one grey level  (ngtdm.py:153)"""
        c = self.coefficients
        p, i = c["p_i"], c["ivector"]
        div = c["Ngp"] * (c["Ngp"] - 1)
        val = (np.sum(p[:, :, None] * p[:, None, :] * (i[:, :, None] - i[:, None, :]) ** 2, (1, 2))
               * np.sum(c["s_i"], 1) / c["Nvp"])
        val[div != 0] /= div[div != 0]
        val[div == 0] = 0
        return val

    def getBusynessFeatureValue(self):
        """Σi p(i) s(i) / Σij |i p(i) − j p(j)| over levels with p ≠ 0; 0 for one grey level  (ngtdm.py:192)"""
        c = self.coefficients
        ip = c["ivector"] * c["p_i"]
        absdiff = np.sum(self._pairs(np.abs(ip[:, :, None] - ip[:, None, :])), (1, 2))
        val = np.sum(c["p_i"] * c["s_i"], 1)
        val[absdiff != 0] = val[absdiff != 0] / absdiff[absdiff != 0]
        val[absdiff == 0] = 0
        return val

    def getComplexityFeatureValue(self):
        """1 / Nvp · Σij |i − j| (p(i) s(i) + p(j) s(j)) / (p(i) + p(j)) over levels with p ≠ 0  (ngtdm.py:223)"""
        c = self.coefficients
        p, i = c["p_i"], c["ivector"]
        ps = p * c["s_i"]
        num = self._pairs(ps[:, :, None] + ps[:, None, :])
        den = p[:, :, None] + p[:, None, :]
        den[den == 0] = 1
        return np.sum(np.abs(i[:, :, None] - i[:, None, :]) * num / den, (1, 2)) / c["Nvp"]

    def getStrengthFeatureValue(self):
        """Σij (p(i) + p(j)) (i − j)² / Σi s(i) over levels with p ≠ 0; 0 where Σ s(i) = 0  (ngtdm.py:256)"""
        c = self.coefficients
        p, i = c["p_i"], c["ivector"]
        tot = np.sum(c["s_i"], 1)
        val = np.sum(self._pairs((p[:, :, None] + p[:, None, :]) * (i[:, :, None] - i[:, None, :]) ** 2), (1, 2))
        val[tot != 0] /= tot[tot != 0]
        val[tot == 0] = 0
        return val
