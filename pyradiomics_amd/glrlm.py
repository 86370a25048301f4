"""Grey Level Run Length Matrix features: interface and feature-name surface of the reference's radiomics/glrlm.py
(RadiomicsGLRLM), matrix built on the MI355X through cMatrices.calculate_glrlm.

P has shape (Nvox, Ngp, Nr', Na): grey levels absent from the ROI dropped (glrlm.py:118-125), optional distance
weighting (:127-150), all-empty angles dropped (:152-166), run lengths that never occur dropped (:187-190).
Nr = runs per angle; every feature is computed per angle and averaged with nanmean (:196-523)."""
from __future__ import annotations

import numpy as np

from .base import RadiomicsFeaturesBase
from .glcm import _weights

_EPS = np.spacing(1)


class RadiomicsGLRLM(RadiomicsFeaturesBase):
    def __init__(self, inputImage, inputMask, **kwargs):
        super().__init__(inputImage, inputMask, **kwargs)
        self.weightingNorm = kwargs.get("weightingNorm")
        self.P_glrlm = None
        self.imageArray = self._applyBinning(self.imageArray)

    def _segmentRoute(self):
        return ("glrlm", {}) if self.weightingNorm is None else None

    def _calculateFeatures(self, voxelCoordinates=None):
        fused = self._fusedVoxelFeatures("glrlm", voxelCoordinates) if self.weightingNorm is None else None
        if fused is None:
            fused = self._fusedSegmentFeatures("glrlm") if self.weightingNorm is None else None
        if fused is not None:
            yield from fused
            return
        yield from super()._calculateFeatures(voxelCoordinates)

    def _initCalculation(self, voxelCoordinates=None):
        self.P_glrlm = self._calculateMatrix(voxelCoordinates)
        self._calculateCoefficients()

    def _calculateMatrix(self, voxelCoordinates=None):
        Ng = self.coefficients["Ng"]
        Nr = np.max(self.imageArray.shape)   # glrlm.py:101: the longest axis of the (cropped) array
        args = [self.imageArray, self.maskArray, Ng, Nr, self.settings.get("force2D", False),
                self.settings.get("force2Ddimension", 0)]
        P, angles = self.cMatrices.calculate_glrlm(*(args + self._matrix_tail(voxelCoordinates)))
        P = np.delete(P, self._absent_levels(), 1)
        if self.weightingNorm is not None:
            w = _weights(angles, np.array(self.inputImage.GetSpacing()[::-1]), self.weightingNorm, "glrlm",
                         self.logger)
            P = np.sum(P * w[None, None, None, :], 3, keepdims=True)
        runs = np.sum(P, (1, 2))
        if P.shape[3] > 1:
            empty = np.where(np.sum(runs, 0) == 0)
            if len(empty[0]) > 0:
                P = np.delete(P, empty, 3)
                runs = np.delete(runs, empty, 1)
        runs[runs == 0] = np.nan
        self.coefficients["Nr"] = runs
        return P

    def _calculateCoefficients(self):
        pr = np.sum(self.P_glrlm, 1)
        pg = np.sum(self.P_glrlm, 2)
        j = np.arange(1, self.P_glrlm.shape[2] + 1, dtype=np.float64)
        unused = np.where(np.sum(pr, (0, 2)) == 0)
        self.P_glrlm = np.delete(self.P_glrlm, unused, 2)
        c = self.coefficients
        c["pr"] = np.delete(pr, unused, 1)
        c["pg"] = pg
        c["ivector"] = c["grayLevels"].astype(float)
        c["jvector"] = np.delete(j, unused)

    # weights along the level (i) and run-length (j) axes, broadcast to (Nvox, Ng, Nr, Na)
    def _i(self):
        return self.coefficients["ivector"][None, :, None, None]

    def _j(self):
        return self.coefficients["jvector"][None, None, :, None]

    def _joint(self, weight):
        return np.nanmean(np.sum(self.P_glrlm * weight, (1, 2)) / self.coefficients["Nr"], 1)

    def getShortRunEmphasisFeatureValue(self):
        """Σij P(i,j) / j² / Nr(θ)  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:196)"""
        c = self.coefficients
        return np.nanmean(np.sum(c["pr"] / (c["jvector"][None, :, None] ** 2), 1) / c["Nr"], 1)

    def getLongRunEmphasisFeatureValue(self):
        """Σij P(i,j) j² / Nr(θ)  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:213)"""
        c = self.coefficients
        return np.nanmean(np.sum(c["pr"] * (c["jvector"][None, :, None] ** 2), 1) / c["Nr"], 1)

    def getGrayLevelNonUniformityFeatureValue(self):
        """Σi (Σj P(i,j))² / Nr(θ)  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:230)"""
        c = self.coefficients
        return np.nanmean(np.sum(c["pg"] ** 2, 1) / c["Nr"], 1)

    def getGrayLevelNonUniformityNormalizedFeatureValue(self):
        """Σi (Σj P(i,j))² / Nr(θ)²  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:246)"""
        c = self.coefficients
        return np.nanmean(np.sum(c["pg"] ** 2, 1) / (c["Nr"] ** 2), 1)

    def getRunLengthNonUniformityFeatureValue(self):
        """Σj (Σi P(i,j))² / Nr(θ)  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:262)"""
        c = self.coefficients
        return np.nanmean(np.sum(c["pr"] ** 2, 1) / c["Nr"], 1)

    def getRunLengthNonUniformityNormalizedFeatureValue(self):
        """Σj (Σi P(i,j))² / Nr(θ)²  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:278)"""
        c = self.coefficients
        return np.nanmean(np.sum(c["pr"] ** 2, 1) / c["Nr"] ** 2, 1)

    def getRunPercentageFeatureValue(self):
        """Nr(θ) / Np  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:294)"""
        c = self.coefficients
        Np = np.sum(c["pr"] * c["jvector"][None, :, None], 1)
        return np.nanmean(c["Nr"] / Np, 1)

    def getGrayLevelVarianceFeatureValue(self):
        """Σij p(i,j) (i − μ)² with μ = Σij p(i,j) i  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:319)"""
        c = self.coefficients
        i = c["ivector"][None, :, None]
        pg = c["pg"] / c["Nr"][:, None, :]
        u = np.sum(pg * i, 1, keepdims=True)
        return np.nanmean(np.sum(pg * (i - u) ** 2, 1), 1)

    def getRunVarianceFeatureValue(self):
        """Σij p(i,j) (j − μ)² with μ = Σij p(i,j) j  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:340)"""
        c = self.coefficients
        j = c["jvector"][None, :, None]
        pr = c["pr"] / c["Nr"][:, None, :]
        u = np.sum(pr * j, 1, keepdims=True)
        return np.nanmean(np.sum(pr * (j - u) ** 2, 1), 1)

    def getRunEntropyFeatureValue(self):
        """−Σij p(i,j) log2(p(i,j) + ε)  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:361)"""
        p = self.P_glrlm / self.coefficients["Nr"][:, None, None, :]
        return np.nanmean(-np.sum(p * np.log2(p + _EPS), (1, 2)), 1)

    def getLowGrayLevelRunEmphasisFeatureValue(self):
        """Σij P(i,j) / i² / Nr(θ)  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:383)"""
        c = self.coefficients
        return np.nanmean(np.sum(c["pg"] / (c["ivector"][None, :, None] ** 2), 1) / c["Nr"], 1)

    def getHighGrayLevelRunEmphasisFeatureValue(self):
        """Σij P(i,j) i² / Nr(θ)  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:400)"""
        c = self.coefficients
        return np.nanmean(np.sum(c["pg"] * (c["ivector"][None, :, None] ** 2), 1) / c["Nr"], 1)

    def getShortRunLowGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) / (i² j²) / Nr(θ)  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:417)"""
        return self._joint(1.0 / ((self._i() ** 2) * (self._j() ** 2)))

    def getShortRunHighGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) i² / j² / Nr(θ)  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:445)"""
        return np.nanmean(np.sum(self.P_glrlm * (self._i() ** 2) / (self._j() ** 2), (1, 2)) / self.coefficients["Nr"], 1)

    def getLongRunLowGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) j² / i² / Nr(θ)  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:471)"""
        return np.nanmean(np.sum(self.P_glrlm * (self._j() ** 2) / (self._i() ** 2), (1, 2)) / self.coefficients["Nr"], 1)

    def getLongRunHighGrayLevelEmphasisFeatureValue(self):
        """Σij P(i,j) i² j² / Nr(θ)  [j = run size; Nr(θ) = Σ P(i,j|θ), per angle, mean over angles]  (glrlm.py:497)"""
        return self._joint((self._j() ** 2) * (self._i() ** 2))
