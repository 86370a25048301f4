"""Feature-class base: the interface of the reference's radiomics/base.py (RadiomicsFeaturesBase) on numpy /
pyradiomics_amd.image.Image inputs.  Same constructor signature, same settings keys, same feature discovery by
`get<Name>FeatureValue` reflection (base.py:163-179), same segment / voxel-based execution (base.py:181-273)."""
from __future__ import annotations

import sys as _sys

import inspect
import os
import logging
import traceback

import numpy as np

from . import backend, imageoperations
from .image import Image, as_array, as_image


def _engine():
    """the device engine (imports torch), loaded on first use: a function-level `from . import engine` goes through
    importlib's locked lookup on every call (11 us each, 54 of them per 256^3 case); sys.modules is a dict lookup"""
    m = _sys.modules.get("pyradiomics_amd.engine")
    if m is None:
        from . import engine as m
    return m


_ENQUEUE_DEFAULT = os.environ.get("PRAD_ENQUEUE_SEGMENT", "1") != "0"     # (A/B switch of the case pipeline, see enqueue())


def deprecated(func):
    """marks a feature function as deprecated (radiomics/__init__.py:18-21)"""
    func._is_deprecated = True
    return func


class RadiomicsFeaturesBase:
    def __init__(self, inputImage, inputMask, **kwargs):
        self.logger = logging.getLogger(self.__module__)
        if inputImage is None or inputMask is None:
            raise ValueError("Missing input image or mask")
        self.settings = kwargs
        self.label = kwargs.get("label", 1)
        self.voxelBased = kwargs.get("voxelBased", False)
        self.coefficients = {}
        self.enabledFeatures = {}
        self.featureValues = {}
        self.featureNames = self.getFeatureNames()
        self.inputImage = as_image(inputImage)
        self.inputMask = as_image(inputMask)
        # Segment mode keeps the volume in HBM from here on (upload once, bin on the device, matrices through the
        # _dev entry points); only the finished matrices come back.  `deviceResident: False` takes the host-array
        # route of the reference (numpy binning, host pointers handed to the operator module).
        self.deviceResident = (not self.voxelBased and bool(kwargs.get("deviceResident", True))
                               and getattr(self.cMatrices, "DEVICE_TENSORS", False))
        self.imageArray = self.inputImage.device_tensor() if self.deviceResident else as_array(self.inputImage)
        if self.voxelBased:
            self._initVoxelBasedCalculation()
        else:
            self._initSegmentBasedCalculation()

    # -- initialisation (base.py:93-125) -------------------------------------------------------------
    def _initSegmentBasedCalculation(self):
        if self.deviceResident:
            self.maskArray = imageoperations.roiTensor(self.inputMask, self.label)
        else:
            self.maskArray = as_array(self.inputMask) == self.label

    def _initVoxelBasedCalculation(self):
        self.masked = self.settings.get("maskedKernel", True)
        maskArray = as_array(self.inputMask) == self.label
        self.labelledVoxelCoordinates = np.array(np.where(maskArray))
        self.allLabelledVoxelCoordinates = self.labelledVoxelCoordinates
        shard = self.settings.get("voxelShard")
        if shard is not None:
            # multi-GPU voxel maps (pyradiomics_amd.batch.voxel_maps_sharded): this process computes only its
            # contiguous slice of the centre list (raster order => a z-slab); kernels still see the whole volume
            rank, world = shard
            n = self.labelledVoxelCoordinates.shape[1]
            self.labelledVoxelCoordinates = self.labelledVoxelCoordinates[:, (n * rank) // world:(n * (rank + 1)) // world]
        # unmasked kernels discretise (and later count) over the whole image
        self.maskArray = maskArray if self.masked else np.ones(self.imageArray.shape, dtype=bool)

    def _initCalculation(self, voxelCoordinates=None):
        pass

    def _applyBinning(self, matrix):
        if self.deviceResident:
            return self._applyBinningDevice(matrix)
        matrix, _ = imageoperations.binImage(matrix, self.maskArray, **self.settings)
        self.coefficients["grayLevels"], self.coefficients["levelCounts"] = np.unique(matrix[self.maskArray],
                                                                                       return_counts=True)
        self.coefficients["Ng"] = int(np.max(self.coefficients["grayLevels"]))
        return matrix

    def _applyBinningDevice(self, tensor):
        """binImage + grey-level bookkeeping in HBM (prad_roi_minmax_dev, then levels and the level census in one pass:
        prad_digitize_counts_dev).
        All feature classes of one derived image share the result: the reference re-bins per class (base.py:119-125)."""
        engine = _engine()
        key = ("levels", self.settings.get("binWidth", 25), self.settings.get("binCount"), id(self.maskArray))
        memo = self.inputImage._derived
        pending = memo.get(key)
        if pending is None or type(pending) is dict:
            if pending is not None:       # queued by prebinDevice() while the host worked on the image before this one
                del memo[key]
                levels, top, edges, counts = engine.bin_image_collect(pending, with_counts=True)
            else:
                levels, top, edges, counts = engine.bin_image(tensor, self.maskArray, with_counts=True, **self.settings)
            levels._prad_memo = {"mask": self.maskArray}      # lets cMatrices serve GLCM and GLRLM from one sweep
            memo[key] = (levels, np.flatnonzero(counts[1:]) + 1, int(counts[1:].sum()), self.maskArray,
                         counts[1:][counts[1:] > 0])
        levels, grayLevels, Ns, _, levelCounts = memo[key]
        self.coefficients["grayLevels"] = grayLevels
        self.coefficients["levelCounts"] = levelCounts
        self.coefficients["Ng"] = int(grayLevels.max())
        self.coefficients["Ns"] = Ns
        return levels

    @staticmethod
    def prebinDevice(inputImage, inputMask, **settings):
        """Queues the discretisation a feature class built on (inputImage, inputMask, settings) will ask for, without waiting
        for it (engine.bin_image_enqueue); _applyBinningDevice collects it.  The case pipeline calls this for derived image
        i + 1 before it collects image i, so the one host round trip of a binning (ROI min / max -> number of levels) has
        passed by the time the classes of image i + 1 are constructed.  Returns True when something was queued."""
        image, mask = as_image(inputImage), as_image(inputMask)
        maskArray = imageoperations.roiTensor(mask, settings.get("label", 1))
        key = ("levels", settings.get("binWidth", 25), settings.get("binCount"), id(maskArray))
        if key in image._derived:
            return False
        token = _engine().bin_image_enqueue(image.device_tensor(), maskArray, **settings)
        if token is None:
            return False
        image._derived[key] = token
        return True

    @staticmethod
    def dropPrebinned(inputImage):
        """waits for and discards what prebinDevice queued for an image that is not going to be evaluated (never raises)"""
        memo = getattr(inputImage, "_derived", None) or {}
        for key in [k for k, v in memo.items() if type(k) is tuple and k and k[0] == "levels" and type(v) is dict]:
            token = memo.pop(key)
            try:
                _engine().bin_image_collect(token)
            except Exception:      # noqa: BLE001
                pass

    @property
    def cMatrices(self):
        return backend.get()

    def _matrix_tail(self, voxelCoordinates):
        """extra positional arguments of the cMatrices calls in voxel mode (e.g. glcm.py:142-143)"""
        if self.voxelBased:
            return [self.settings.get("kernelRadius", 1), voxelCoordinates]
        return []

    def _absent_levels(self):
        """0-based rows of grey levels 1..Ng that do not occur in the ROI (e.g. glrlm.py:118-125)"""
        Ng = self.coefficients["Ng"]
        present = np.zeros(Ng + 1, dtype=bool)
        present[self.coefficients["grayLevels"]] = True
        return np.where(~present[1:])[0]

    # -- feature switches (base.py:127-179) ----------------------------------------------------------
    def enableFeatureByName(self, featureName, enable=True):
        if featureName not in self.featureNames:
            raise LookupError("Feature not found: " + featureName)
        if self.featureNames[featureName]:
            self.logger.warning("Feature %s is deprecated, use with caution!", featureName)
        self.enabledFeatures[featureName] = enable

    def enableAllFeatures(self):
        for name, is_deprecated in self.featureNames.items():
            if not is_deprecated:
                self.enableFeatureByName(name, True)

    def disableAllFeatures(self):
        self.enabledFeatures = {}
        self.featureValues = {}

    _feature_names_cache: dict = {}

    @classmethod
    def getFeatureNames(cls):
        """{feature name: deprecated?} (base.py:221-231); cached per class -- inspect.getmembers cost 3 ms per case"""
        names = RadiomicsFeaturesBase._feature_names_cache.get(cls)
        if names is None:
            names = {name[3:-12]: getattr(fn, "_is_deprecated", False)
                     for name, fn in inspect.getmembers(cls)
                     if name.startswith("get") and name.endswith("FeatureValue")}
            RadiomicsFeaturesBase._feature_names_cache[cls] = names
        return dict(names)

    # -- execution (base.py:181-273) -----------------------------------------------------------------
    def execute(self):
        if len(self.enabledFeatures) == 0:
            self.enableAllFeatures()
        if self.voxelBased:
            self._calculateVoxels()
        else:
            self._calculateSegment()
        return self.featureValues

    def _calculateSegment(self):
        for _ok, name, value in self._calculateFeatures():
            # (numpy.squeeze of the reference, base.py:209; a 0-d array -- what the device formulas hand over -- is its own squeeze)
            self.featureValues[name] = value if type(value) is np.ndarray and value.ndim == 0 else np.squeeze(value)

    def _calculateVoxels(self):
        initValue = self.settings.get("initValue", 0)
        voxelBatch = self.settings.get("voxelBatch", -1)
        shape = self.imageArray.shape
        for name, enabled in self.enabledFeatures.items():
            if enabled:
                self.featureValues[name] = np.full(shape, initValue, dtype="float")
        total = self.labelledVoxelCoordinates.shape[1]
        if voxelBatch < 0:
            voxelBatch = total
        start = 0
        while start < total:
            coords = self.labelledVoxelCoordinates[:, start:start + voxelBatch]
            for ok, name, value in self._calculateFeatures(coords):
                if ok:
                    self.featureValues[name][tuple(coords)] = value
            start += voxelBatch
        for name, enabled in self.enabledFeatures.items():
            if enabled:
                ref = self.inputImage
                self.featureValues[name] = ref.like(self.featureValues[name]) if isinstance(ref, Image) \
                    else Image(self.featureValues[name])

    def _fusedVoxelFeatures(self, cls, voxelCoordinates, alpha=0):
        """voxel mode: feature maps of class `cls` straight from the device when the operator backend offers the
        fused kernels and they cover the request (no (Nvox, Ng, ...) intermediate); None otherwise.
        `fusedVoxel: False` in the settings forces the reference's route (matrix + numpy formulas, base.py:253-273)."""
        fused = getattr(self.cMatrices, "voxel_texture_features", None)
        names = [n for n, on in self.enabledFeatures.items() if on]
        if not (self.voxelBased and voxelCoordinates is not None and fused is not None and names
                and self.settings.get("fusedVoxel", True)):
            return None
        if getattr(self, "_fusedInputs", None) is None:           # upload the discretised volume once, not per batch
            up = getattr(self.cMatrices, "_to_device", lambda a, **kw: a)
            self._fusedInputs = (up(self.imageArray, integer=True), up(self.maskArray))
        try:
            vals = fused(cls, self._fusedInputs[0], self._fusedInputs[1], np.array(self.settings.get("distances", [1])),
                         self.coefficients["Ng"], self.settings.get("force2D", False),
                         self.settings.get("force2Ddimension", 0), self.settings.get("kernelRadius", 1),
                         voxelCoordinates, names, alpha)
        except NotImplementedError:
            return None
        return [(True, n, vals[n]) for n in names]

    # -- case pipeline: enqueue now, collect in execute() (no reference analogue) ---------------------------
    def _segmentRoute(self):
        """(class key, extra keyword arguments) of the fused segment route this class takes with its current settings,
        None when it has none"""
        return None

    def enqueue(self):
        """Queues this class's segment-mode device work -- matrix and feature kernels -- on the current stream without
        waiting for it; execute() then only collects the values.  The caller synchronises in between through
        cMatrices.segment_sync() and calls dropEnqueued() when that reports voided values.  Returns True when work was
        enqueued (False: execute() computes as usual)."""
        self._enqueued = None
        fused = getattr(self.cMatrices, "segment_features_enqueue", None)
        route = None if (self.voxelBased or not self.deviceResident or fused is None
                         or not self.settings.get("fusedSegment", True)
                         or not self.settings.get("enqueueSegment", _ENQUEUE_DEFAULT)) else self._segmentRoute()
        if route is None or route[0] not in getattr(self.cMatrices, "ENQUEUE_CLASSES", ()):
            return False
        if len(self.enabledFeatures) == 0:
            self.enableAllFeatures()
        names = [n for n, on in self.enabledFeatures.items() if on]
        if not names:
            return False
        try:
            finish = fused(route[0], self.imageArray, self.maskArray, self.coefficients["Ng"], names,
                           distances=self.settings.get("distances", [1]), force2D=self.settings.get("force2D", False),
                           force2Ddimension=self.settings.get("force2Ddimension", 0), Ns=self.coefficients.get("Ns"),
                           **route[1])
        except NotImplementedError:
            return False
        self._enqueued = (tuple(names), finish)
        return True

    def dropEnqueued(self):
        self._enqueued = None
        # matrices queued with deferred=True were memoised on the shared levels tensor before their verdict was known: a
        # voided image (level outside [1, Ng]) must not serve them to the synchronous route that raises the reference's error
        memo = getattr(self.imageArray, "_prad_memo", None)
        if isinstance(memo, dict):
            for k in [k for k in memo if k != "mask"]:
                del memo[k]

    def imageRequest(self):
        """(class key, request dict) for cMatrices.segment_image_enqueue -- this class's share of the one-call-per-image
        enqueue -- or None when it has to queue (or compute) on its own"""
        fused = getattr(self.cMatrices, "segment_image_enqueue", None)
        if (self.voxelBased or not self.deviceResident or fused is None or not self.settings.get("fusedSegment", True)
                or not self.settings.get("enqueueSegment", _ENQUEUE_DEFAULT)):
            return None
        route = self._segmentRoute()
        if route is None or [int(d) for d in np.asarray(self.settings.get("distances", [1])).ravel()] != [1]:
            return None
        if len(self.enabledFeatures) == 0:
            self.enableAllFeatures()
        names = [n for n, on in self.enabledFeatures.items() if on]
        if not names:
            return None
        return route[0], dict(route[1], features=names)

    def takeEnqueued(self, finish):
        """hands this class the finish() of a queue somebody else filled (featureextractor, one call per image)"""
        self._enqueued = (tuple(n for n, on in self.enabledFeatures.items() if on), finish)

    def _fusedSegmentFeatures(self, cls, host_only=(), **extra):
        """segment mode on the device-resident route: matrix AND feature formulas on the device when the operator
        backend offers it (only the feature values come back); None otherwise.  Features named in `host_only` (GLCM's
        MCC) are evaluated from the host matrix in addition.  `fusedSegment: False` forces the reference's route
        (matrix to the host, numpy formulas)."""
        fused = getattr(self.cMatrices, "segment_features", None)
        names = [n for n, on in self.enabledFeatures.items() if on]
        if (self.voxelBased or not self.deviceResident or fused is None or not names
                or not self.settings.get("fusedSegment", True)):
            return None
        dev_names = [n for n in names if n not in host_only]
        queued, self._enqueued = getattr(self, "_enqueued", None), None
        try:
            if queued is not None and queued[0] == tuple(dev_names):
                vals = queued[1]()
            else:
                vals = fused(cls, self.imageArray, self.maskArray, self.coefficients["Ng"], dev_names,
                             distances=self.settings.get("distances", [1]), force2D=self.settings.get("force2D", False),
                             force2Ddimension=self.settings.get("force2Ddimension", 0), Ns=self.coefficients.get("Ns"),
                             **extra) if dev_names else {}
        except NotImplementedError:
            return None
        out = []
        rest = [n for n in names if n not in vals]      # host_only names and whatever the device declined
        if rest:
            getattr(self, "_initHostOnly", self._initCalculation)(None)
        for n in names:
            if n in vals:
                out.append((True, n, np.array(vals[n])))
            else:
                try:
                    out.append((True, n, getattr(self, "get%sFeatureValue" % n)()))
                except Exception:
                    self.logger.error("FAILED: %s", traceback.format_exc())
                    out.append((False, n, np.nan))
        return out

    def _calculateFeatures(self, voxelCoordinates=None):
        self._initCalculation(voxelCoordinates)
        for name, enabled in self.enabledFeatures.items():
            if not enabled:
                continue
            try:
                yield True, name, getattr(self, "get%sFeatureValue" % name)()
            except DeprecationWarning as dw:
                self.logger.warning("Feature %s is deprecated: %s", name, dw)
                yield False, name, np.nan
            except Exception:
                self.logger.error("FAILED: %s", traceback.format_exc())
                yield False, name, np.nan
