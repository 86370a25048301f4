"""Grey Level Co-occurrence Matrix features: interface and feature-name surface of the reference's
radiomics/glcm.py (RadiomicsGLCM), with the matrix built on the MI355X through cMatrices.calculate_glcm.

Conventions (glcm.py:113-258): P has shape (Nvox, Ngp, Ngp, Na) after dropping grey levels absent from the ROI
(:149-152), adding the transpose when symmetricalGLCM (:155-157), optional distance weighting that collapses the
angle axis (:160-182), dropping all-empty angles (:186-198) and per-angle normalisation (:201-203).  Every feature
is evaluated per angle and averaged with nanmean (:260-887).  i / j are the ACTUAL level values of the kept rows;
the sum / difference axes use the maximum level Ng, not the pruned count (:217-224)."""
from __future__ import annotations

import numpy as np

from .base import RadiomicsFeaturesBase, deprecated

_EPS = np.spacing(1)


def _weights(angles, spacing_zyx, norm, kind, logger):
    """per-angle weights: GLCM uses exp(-d^2) (glcm.py:160-182), GLRLM uses the distance d itself
    (glrlm.py:127-148), with d the infinity / euclidean / manhattan norm of the angle in mm"""
    out = np.empty(len(angles))
    for n, a in enumerate(angles):
        step = np.abs(a) * spacing_zyx
        if norm == "infinity":
            d, d2 = max(step), max(step) ** 2
        elif norm == "euclidean":
            d2 = np.sum(step ** 2)
            d = np.sqrt(d2)
        elif norm == "manhattan":
            d = np.sum(step)
            d2 = d ** 2
        elif norm == "no_weighting":
            out[n] = 1
            continue
        else:
            logger.warning('weigthing norm "%s" is unknown, W is set to 1', norm)
            out[n] = 1
            continue
        out[n] = np.exp(-d2) if kind == "glcm" else d
    return out


class RadiomicsGLCM(RadiomicsFeaturesBase):
    def __init__(self, inputImage, inputMask, **kwargs):
        super().__init__(inputImage, inputMask, **kwargs)
        self.symmetricalGLCM = kwargs.get("symmetricalGLCM", True)
        self.weightingNorm = kwargs.get("weightingNorm")
        self.P_glcm = None
        self.imageArray = self._applyBinning(self.imageArray)

    def _initCalculation(self, voxelCoordinates=None):
        self.P_glcm = self._calculateMatrix(voxelCoordinates)
        self._calculateCoefficients()

    def _initHostOnly(self, voxelCoordinates=None):
        """what MCC needs when every other feature came from the device: the normalised matrix and its marginals"""
        self.P_glcm = self._calculateMatrix(voxelCoordinates)
        self.coefficients["px"] = self.P_glcm.sum(2, keepdims=True)
        self.coefficients["py"] = self.P_glcm.sum(1, keepdims=True)

    def _segmentRoute(self):
        return ("glcm", {"symmetrical": self.symmetricalGLCM}) if self.weightingNorm is None else None

    def _calculateFeatures(self, voxelCoordinates=None):
        """voxel mode: when the operator backend offers the fused kernel and it covers the request, feature maps
        come straight from the device (no (Nvox, Ng, Ng, Na) intermediate); otherwise the reference's route
        (matrix + numpy formulas, base.py:253-273) is taken.  `fusedVoxel: False` in the settings forces the latter."""
        if self.weightingNorm is None:
            seg = self._fusedSegmentFeatures("glcm", symmetrical=self.symmetricalGLCM)
            if seg is not None:
                for ok, n, v in seg:     # glcm.py:702-703: a ROI of one grey level has 1 x 1 matrices -> MCC = 1
                    yield ok, n, (np.array(1.0) if n == "MCC" and ok and len(self.coefficients["grayLevels"]) < 2 else v)
                return
        fused = getattr(self.cMatrices, "voxel_glcm_features", None)
        names = [n for n, on in self.enabledFeatures.items() if on]
        if (self.voxelBased and voxelCoordinates is not None and fused is not None and names
                and self.settings.get("fusedVoxel", True) and self.weightingNorm is None):
            covered = list(getattr(self.cMatrices, "VOXEL_GLCM_FEATURES", names))
            if hasattr(self.cMatrices, "voxel_glcm_features") and len(self.coefficients["grayLevels"]) >= 2:
                covered.append("MCC")                              # (a one-level ROI takes the reference's "-> 1" rule)
            dev_names = [n for n in names if n in covered]
            rest = [n for n in names if n not in covered]          # MCC (and deprecated names, which only raise)
            try:
                vals = fused(self.imageArray, self.maskArray, np.array(self.settings.get("distances", [1])),
                             self.coefficients["Ng"], self.settings.get("force2D", False),
                             self.settings.get("force2Ddimension", 0), self.settings.get("kernelRadius", 1),
                             voxelCoordinates, dev_names, self.symmetricalGLCM) if dev_names else {}
            except NotImplementedError:
                vals = None
                if "MCC" in dev_names:                             # e.g. more than 64 levels: everything but MCC may still fuse
                    dev_names = [n for n in dev_names if n != "MCC"]
                    rest = [n for n in names if n not in dev_names]
                    try:
                        vals = fused(self.imageArray, self.maskArray, np.array(self.settings.get("distances", [1])),
                                     self.coefficients["Ng"], self.settings.get("force2D", False),
                                     self.settings.get("force2Ddimension", 0), self.settings.get("kernelRadius", 1),
                                     voxelCoordinates, dev_names, self.symmetricalGLCM) if dev_names else {}
                    except NotImplementedError:
                        vals = None
            if vals is not None:
                if rest:                                           # only these take the per-kernel matrix route
                    self._initHostOnly(voxelCoordinates)
                for n in names:
                    if n in vals:
                        yield True, n, vals[n]
                    else:
                        try:
                            yield True, n, getattr(self, "get%sFeatureValue" % n)()
                        except DeprecationWarning as dw:
                            self.logger.warning("Feature %s is deprecated: %s", n, dw)
                            yield False, n, np.nan
                return
        yield from super()._calculateFeatures(voxelCoordinates)

    def _calculateMatrix(self, voxelCoordinates=None):
        Ng = self.coefficients["Ng"]
        args = [self.imageArray, self.maskArray, np.array(self.settings.get("distances", [1])), Ng,
                self.settings.get("force2D", False), self.settings.get("force2Ddimension", 0)]
        P, angles = self.cMatrices.calculate_glcm(*(args + self._matrix_tail(voxelCoordinates)))
        keep = self.coefficients["grayLevels"] - 1
        P = P[:, keep][:, :, keep]
        if self.symmetricalGLCM:
            P = P + P.transpose((0, 2, 1, 3))
        if self.weightingNorm is not None:
            w = _weights(angles, np.array(self.inputImage.GetSpacing()[::-1]), self.weightingNorm, "glcm",
                         self.logger)
            P = np.sum(P * w[None, None, None, :], 3, keepdims=True)
        total = np.sum(P, (1, 2))
        if P.shape[3] > 1:
            empty = np.where(np.sum(total, 0) == 0)
            if len(empty[0]) > 0:
                P = np.delete(P, empty, 3)
                total = np.delete(total, empty, 1)
        total[total == 0] = np.nan
        P /= total[:, None, None, :]
        return P

    def _calculateCoefficients(self):
        P = self.P_glcm
        Ng = self.coefficients["Ng"]
        levels = self.coefficients["grayLevels"].astype("float")
        i, j = np.meshgrid(levels, levels, indexing="ij", sparse=True)
        kSum = np.arange(2, 2 * Ng + 1, dtype="float")
        kDiff = np.arange(0, Ng, dtype="float")
        c = self.coefficients
        c["eps"] = _EPS
        c["i"], c["j"] = i, j
        c["kValuesSum"], c["kValuesDiff"] = kSum, kDiff
        c["px"] = P.sum(2, keepdims=True)
        c["py"] = P.sum(1, keepdims=True)
        c["ux"] = np.sum(i[None, :, :, None] * P, (1, 2), keepdims=True)
        c["uy"] = np.sum(j[None, :, :, None] * P, (1, 2), keepdims=True)
        c["pxAddy"] = np.array([np.sum(P[:, i + j == k, :], 1) for k in kSum]).transpose((1, 0, 2))
        c["pxSuby"] = np.array([np.sum(P[:, np.abs(i - j) == k, :], 1) for k in kDiff]).transpose((1, 0, 2))
        c["HXY"] = (-1) * np.sum(P * np.log2(P + _EPS), (1, 2))

    # -- helpers ---------------------------------------------------------------------------------------
    def _centred(self, power):
        c = self.coefficients
        dev = (c["i"] + c["j"])[None, :, :, None] - c["ux"] - c["uy"]
        return np.nanmean(np.sum(self.P_glcm * dev ** power, (1, 2)), 1)

    def _diff_weighted(self, denom):
        return np.nanmean(np.sum(self.coefficients["pxSuby"] / denom[None, :, None], 1), 1)

    # -- features (glcm.py:260-887) --------------------------------------------------------------------
    def getAutocorrelationFeatureValue(self):
        """Σij p(i,j) · i · j  (glcm.py:260)"""
        c = self.coefficients
        return np.nanmean(np.sum(self.P_glcm * (c["i"] * c["j"])[None, :, :, None], (1, 2)), 1)

    def getJointAverageFeatureValue(self):
        """μx = Σij p(i,j) · i  (glcm.py:274)"""
        return self.coefficients["ux"].mean((1, 2, 3))

    def getClusterProminenceFeatureValue(self):
        """Σij (i + j − μx − μy)⁴ p(i,j)  (glcm.py:294)"""
        return self._centred(4)

    def getClusterShadeFeatureValue(self):
        """Σij (i + j − μx − μy)³ p(i,j)  (glcm.py:314)"""
        return self._centred(3)

    def getClusterTendencyFeatureValue(self):
        """Σij (i + j − μx − μy)² p(i,j)  (glcm.py:334)"""
        return self._centred(2)

    def getContrastFeatureValue(self):
        """Σij (i − j)² p(i,j)  (glcm.py:353)"""
        c = self.coefficients
        return np.nanmean(np.sum(self.P_glcm * (np.abs(c["i"] - c["j"]))[None, :, :, None] ** 2, (1, 2)), 1)

    def getCorrelationFeatureValue(self):
        """(Σij p(i,j) · i · j − μx μy) / (σx σy); 1 where σx σy = 0 (flat region)  (glcm.py:368)"""
        c = self.coefficients
        P = self.P_glcm
        di = c["i"][None, :, :, None] - c["ux"]
        dj = c["j"][None, :, :, None] - c["uy"]
        sigx = np.sum(P * di ** 2, (1, 2), keepdims=True) ** 0.5
        sigy = np.sum(P * dj ** 2, (1, 2), keepdims=True) ** 0.5
        corr = np.sum(P * di * dj, (1, 2), keepdims=True) / (sigx * sigy + _EPS)
        corr[sigx * sigy == 0] = 1
        return np.nanmean(corr, (1, 2, 3))

    def getDifferenceAverageFeatureValue(self):
        """Σk k · p(x−y)(k)  (glcm.py:412)"""
        c = self.coefficients
        return np.nanmean(np.sum(c["kValuesDiff"][None, :, None] * c["pxSuby"], 1), 1)

    def getDifferenceEntropyFeatureValue(self):
        """−Σk p(x−y)(k) log2(p(x−y)(k) + ε)  (glcm.py:428)"""
        p = self.coefficients["pxSuby"]
        return np.nanmean((-1) * np.sum(p * np.log2(p + _EPS), 1), 1)

    def getDifferenceVarianceFeatureValue(self):
        """Σk (k − DifferenceAverage)² p(x−y)(k)  (glcm.py:443)"""
        c = self.coefficients
        k = c["kValuesDiff"][None, :, None]
        mean = np.sum(k * c["pxSuby"], 1, keepdims=True)
        return np.nanmean(np.sum(c["pxSuby"] * (k - mean) ** 2, 1), 1)

    @deprecated
    def getDissimilarityFeatureValue(self):
        """deprecated: equal to DifferenceAverage  (glcm.py:460)"""
        raise DeprecationWarning("GLCM - Dissimilarity is mathematically equal to GLCM - Difference Average")

    def getJointEnergyFeatureValue(self):
        """Σij p(i,j)²  (glcm.py:480)"""
        return np.nanmean(np.sum(self.P_glcm ** 2, (1, 2)), 1)

    def getJointEntropyFeatureValue(self):
        """HXY = −Σij p(i,j) log2(p(i,j) + ε)  (glcm.py:498)"""
        return np.nanmean(self.coefficients["HXY"], 1)

    @deprecated
    def getHomogeneity1FeatureValue(self):
        """deprecated: equal to Id  (glcm.py:516)"""
        raise DeprecationWarning("GLCM - Homogeneity 1 is mathematically equal to GLCM - Inverse Difference")

    @deprecated
    def getHomogeneity2FeatureValue(self):
        """deprecated: equal to Idm  (glcm.py:536)"""
        raise DeprecationWarning("GLCM - Homogeneity 2 is mathematically equal to GLCM - Inverse Difference Moment")

    def getImc1FeatureValue(self):
        """(HXY − HXY1) / max(HX, HY); 0 where both marginal entropies are 0  (glcm.py:555)"""
        c = self.coefficients
        px, py = c["px"], c["py"]
        HX = (-1) * np.sum(px * np.log2(px + _EPS), (1, 2))
        HY = (-1) * np.sum(py * np.log2(py + _EPS), (1, 2))
        HXY1 = (-1) * np.sum(self.P_glcm * np.log2(px * py + _EPS), (1, 2))
        div = np.fmax(HX, HY)
        imc1 = c["HXY"] - HXY1
        imc1[div != 0] /= div[div != 0]
        imc1[div == 0] = 0
        return np.nanmean(imc1, 1)

    def getImc2FeatureValue(self):
        """sqrt(1 − exp(−2 (HXY2 − HXY))); 0 where HXY > HXY2 (numerical noise on flat regions)  (glcm.py:614)"""
        c = self.coefficients
        pxy = c["px"] * c["py"]
        HXY2 = (-1) * np.sum(pxy * np.log2(pxy + _EPS), (1, 2))
        imc2 = (1 - np.e ** (-2 * (HXY2 - c["HXY"]))) ** 0.5
        imc2[HXY2 == c["HXY"]] = 0
        return np.nanmean(imc2, 1)

    def getIdmFeatureValue(self):
        """Σk p(x−y)(k) / (1 + k²)  (glcm.py:649)"""
        return self._diff_weighted(1 + self.coefficients["kValuesDiff"] ** 2)

    def getMCCFeatureValue(self):
        """sqrt of the second largest eigenvalue of Q(i,j) = sum_k P(i,k) P(j,k) / (px(i) py(k)) (glcm.py:665-707).
        The reference hands the non-symmetric Q to np.linalg.eigvals.  Q = Dx^-1 P Dy^-1 P^T is similar to A A^T with
        A(i,k) = P(i,k) / sqrt(px(i) py(k)), so its eigenvalues are the squared singular values of A: MCC is the second
        largest singular value of A (batched SVD; the eps guard of the reference's denominator is kept in A)."""
        c = self.coefficients
        P, px, py = self.P_glcm, c["px"], c["py"]
        if P.shape[1] < 2:
            return 1                    # glcm.py:702-703
        A = P / np.sqrt(px * py + _EPS)                              # (Nvox, Ng, Ng, Na)
        A = np.where(np.isnan(A), 0.0, A)                            # kernels with an empty angle (P is NaN there)
        sv = np.linalg.svd(A.transpose((0, 3, 1, 2)), compute_uv=False)      # (Nvox, Na, Ng), descending
        mcc = sv[:, :, 1]
        empty = np.isnan(np.sum(P, (1, 2)))                          # (Nvox, Na): angle absent in this kernel
        mcc = np.where(empty, np.nan, mcc)
        return np.nanmean(mcc, 1)

    def getIdmnFeatureValue(self):
        """Σk p(x−y)(k) / (1 + k² / Ng²)  (glcm.py:709)"""
        c = self.coefficients
        return self._diff_weighted(1 + (c["kValuesDiff"] ** 2) / (c["Ng"] ** 2))

    def getIdFeatureValue(self):
        """Σk p(x−y)(k) / (1 + k)  (glcm.py:729)"""
        return self._diff_weighted(1 + self.coefficients["kValuesDiff"])

    def getIdnFeatureValue(self):
        """Σk p(x−y)(k) / (1 + k / Ng)  (glcm.py:744)"""
        c = self.coefficients
        return self._diff_weighted(1 + c["kValuesDiff"] / c["Ng"])

    def getInverseVarianceFeatureValue(self):
        """Σ(k>0) p(x−y)(k) / k²  (glcm.py:762)"""
        c = self.coefficients
        return np.nanmean(np.sum(c["pxSuby"][:, 1:, :] / c["kValuesDiff"][None, 1:, None] ** 2, 1), 1)

    def getMaximumProbabilityFeatureValue(self):
        """max p(i,j)  (glcm.py:778)"""
        return np.nanmean(np.amax(self.P_glcm, (1, 2)), 1)

    def getSumAverageFeatureValue(self):
        """Σk k · p(x+y)(k), k = 2 .. 2Ng  (glcm.py:795)"""
        c = self.coefficients
        return np.nanmean(np.sum(c["kValuesSum"][None, :, None] * c["pxAddy"], 1), 1)

    @deprecated
    def getSumVarianceFeatureValue(self):
        """deprecated: equal to ClusterTendency  (glcm.py:825)"""
        raise DeprecationWarning("GLCM - Sum Variance is mathematically equal to GLCM - Cluster Tendency")

    def getSumEntropyFeatureValue(self):
        """−Σk p(x+y)(k) log2(p(x+y)(k) + ε)  (glcm.py:844)"""
        p = self.coefficients["pxAddy"]
        return np.nanmean((-1) * np.sum(p * np.log2(p + _EPS), 1), 1)

    def getSumSquaresFeatureValue(self):
        """Σij (i − μx)² p(i,j)  (glcm.py:859)"""
        c = self.coefficients
        return np.nanmean(np.sum(self.P_glcm * (c["i"][None, :, :, None] - c["ux"]) ** 2, (1, 2)), 1)
