"""`RadiomicsFeatureExtractor`: the orchestration API of the reference's radiomics/featureextractor.py for the
path this package accelerates -- image types Original / Wavelet / LoG / Square / SquareRoot / Logarithm / Exponential, feature classes firstorder / glcm / glrlm /
glszm / gldm / ngtdm, segment-based and voxel-based extraction -- without SimpleITK / pykwalify (not installed on the build and
GPU hosts).  Same constructor forms (parameter file, dict, or keyword settings), the same enable*/disable* methods,
the same `execute(image, mask, label=None, voxelBased=False)` returning an OrderedDict keyed
`<imageType>_<featureClass>_<featureName>` (featureextractor.py:241-396, 560-604).

Not provided (out of scope): the shape classes,
the remaining image types, parameter-file schema validation.  Enabling them raises/ warns instead of silently
computing something else."""
from __future__ import annotations

import collections
import contextlib
import json
import logging
import os
from itertools import chain

import numpy as np

from . import __version__, backend, filters, imageoperations
from .image import Image, as_array, as_image, read_image

logger = logging.getLogger(__name__)

_FEATURE_CLASSES = ("firstorder", "glcm", "gldm", "glrlm", "glszm", "ngtdm")     # the reference's (alphabetical) order
_IMAGE_TYPES = {"Original": filters.getOriginalImage, "Wavelet": filters.getWaveletImage, "LoG": filters.getLoGImage,
                "Square": filters.getSquareImage, "SquareRoot": filters.getSquareRootImage,
                "Logarithm": filters.getLogarithmImage, "Exponential": filters.getExponentialImage,
                "Gradient": filters.getGradientImage}


_feature_classes = None


def getFeatureClasses():
    """name -> class (radiomics/__init__.py:64-117), restricted to the classes on the accelerated path.  Built once:
    the reference caches its dictionary too (`_featureClasses`), and six importlib lookups per derived image were 0.6 ms
    of a 10 ms case."""
    global _feature_classes
    if _feature_classes is None:
        import importlib
        names = {"firstorder": "RadiomicsFirstOrder"}
        _feature_classes = {n: getattr(importlib.import_module("pyradiomics_amd." + n), names.get(n, "Radiomics" + n.upper()))
                            for n in _FEATURE_CLASSES}
    return dict(_feature_classes)


def getImageTypes():
    return list(_IMAGE_TYPES)


class RadiomicsFeatureExtractor:
    def __init__(self, *args, **kwargs):
        self.settings = {}
        self.enabledImagetypes = {}
        self.enabledFeatures = {}
        self.featureClassNames = list(_FEATURE_CLASSES)
        if len(args) == 1 and isinstance(args[0], dict):
            self._applyParams(paramsDict=args[0])
            self.settings.update(kwargs)          # keyword settings override the parameter set (featureextractor.py:117-137)
        elif len(args) == 1 and isinstance(args[0], (str, os.PathLike)):
            self._applyParams(paramsFile=args[0])
            self.settings.update(kwargs)
        else:
            self.settings = self._getDefaultSettings()
            self.settings.update(kwargs)
            self.enabledImagetypes = {"Original": {}}
            self.enabledFeatures = {n: [] for n in self.featureClassNames}

    @staticmethod
    def _getDefaultSettings():
        # featureextractor.py:139-163 (keys that reach the accelerated path keep their defaults)
        return {"minimumROIDimensions": 2, "minimumROISize": None, "normalize": False, "normalizeScale": 1,
                "removeOutliers": None, "resampledPixelSpacing": None, "interpolator": "sitkBSpline", "preCrop": False,
                "padDistance": 5, "distances": [1], "force2D": False, "force2Ddimension": 0, "resegmentRange": None,
                "label": 1, "additionalInfo": True}

    # -- configuration (featureextractor.py:165-239, 606-763) ----------------------------------------------
    def loadParams(self, paramsFile):
        self._applyParams(paramsFile=paramsFile)

    def loadJSONParams(self, jsonString):
        self._applyParams(paramsDict=json.loads(jsonString))

    def _applyParams(self, paramsFile=None, paramsDict=None):
        if paramsFile is not None:
            import yaml
            with open(paramsFile) as f:
                params = yaml.safe_load(f)
        else:
            params = paramsDict
        params = params or {}
        # the reference validates against radiomics/schemas/paramSchema.yaml here (pykwalify)
        from . import paramcheck
        paramcheck.validate(params, list(_IMAGE_TYPES), {n: list(c.getFeatureNames()) for n, c in getFeatureClasses().items()})
        self.settings = self._getDefaultSettings()
        self.settings.update(params.get("setting") or {})
        self.settings.update(params.get("voxelSetting") or {})
        types = params.get("imageType")
        self.enabledImagetypes = {"Original": {}} if not types else {k: (v or {}) for k, v in types.items()}
        for t in self.enabledImagetypes:
            if t not in _IMAGE_TYPES:
                raise NotImplementedError("image type %r is outside the accelerated path (available: %s)"
                                          % (t, ", ".join(_IMAGE_TYPES)))
        classes = params.get("featureClass")
        if not classes:
            self.enabledFeatures = {n: [] for n in self.featureClassNames}
        else:
            self.enabledFeatures = {}
            for k, v in classes.items():
                if k not in self.featureClassNames:
                    logger.warning("feature class %r is outside the accelerated path and is skipped", k)
                    continue
                self.enabledFeatures[k] = list(v) if v else []

    def addProvenance(self, provenance_on=True):
        self.settings["additionalInfo"] = provenance_on

    def enableAllImageTypes(self):
        self.enabledImagetypes = {t: {} for t in _IMAGE_TYPES}

    def disableAllImageTypes(self):
        self.enabledImagetypes = {}

    def enableImageTypeByName(self, imageType, enabled=True, customArgs=None):
        if imageType not in _IMAGE_TYPES:
            raise NotImplementedError("image type %r is outside the accelerated path" % imageType)
        if enabled:
            self.enabledImagetypes[imageType] = customArgs or {}
        else:
            self.enabledImagetypes.pop(imageType, None)

    def enableImageTypes(self, **enabledImagetypes):
        for k, v in enabledImagetypes.items():
            self.enableImageTypeByName(k, True, v)

    def enableAllFeatures(self):
        self.enabledFeatures = {n: [] for n in self.featureClassNames}

    def disableAllFeatures(self):
        self.enabledFeatures = {}

    def enableFeatureClassByName(self, featureClass, enabled=True):
        if featureClass not in self.featureClassNames:
            logger.warning("Feature class %s is not recognized", featureClass)
            return
        if enabled:
            self.enabledFeatures[featureClass] = []
        else:
            self.enabledFeatures.pop(featureClass, None)

    def enableFeaturesByName(self, **enabledFeatures):
        for k, v in enabledFeatures.items():
            if k in self.featureClassNames:
                self.enabledFeatures[k] = list(v) if v else []
            else:
                logger.warning("Feature class %s is not recognized", k)

    # -- execution ---------------------------------------------------------------------------------------
    @staticmethod
    def loadImage(imageFilepath, maskFilepath, **kwargs):
        """paths to NRRD / NIfTI-1 / MetaImage files, pyradiomics_amd.image.Image objects, or numpy arrays (z, y, x)"""
        def load(x):
            if isinstance(x, (str, os.PathLike)):
                return read_image(os.fspath(x))
            return as_image(x)
        image, mask = load(imageFilepath), load(maskFilepath)
        if getattr(mask, "components", None) is None and len(mask.shape) == len(image.shape) + 1:
            # an array with a trailing component axis (a vector mask given without file geometry)
            mask = Image(mask.array, image.GetSpacing(), image.GetOrigin(), image.GetDirection())
        if getattr(mask, "components", None) is not None or len(mask.shape) == len(mask.GetSpacing()) + 1:
            mask = imageoperations.getMask(mask, **kwargs)      # the channel of a vector mask (label_channel)
        # (a scalar mask is not scanned here: execute() finds a missing label when it takes the bounding box, on the
        # device, and raises the same "Label (..) not present in mask")
        if len(image.shape) != len(mask.shape):
            raise ValueError("Image/Mask datatype or size mismatch: %s vs %s" % (image.shape, mask.shape))
        # imageoperations.checkMask step 1 (:241-287): same grid within geometryTolerance, or correctMask resamples
        mask = imageoperations.checkMaskGeometry(image, mask, **kwargs)
        return image, mask

    def execute(self, imageFilepath, maskFilepath, label=None, label_channel=None, voxelBased=False):
        """featureextractor.py:241-343: every enabled feature of every enabled image type for one (image, mask) pair"""
        steps = self._executeSteps(imageFilepath, maskFilepath, label, label_channel, voxelBased)
        try:
            while True:
                next(steps)
        except StopIteration as done:
            return done.value

    def executeMany(self, cases, label=None, label_channel=None, voxelBased=False):
        """Generator over the results of `cases` -- (image, mask) pairs, or (image, mask, label) -- in their order, with ONE CASE OF
        OVERLAP on this thread: case i + 1 is loaded, uploaded, filtered and its first derived images queued before the values of
        case i's last derived image are collected, so the GPU works through case i's tail while the host does case i + 1's head
        (the upload of a 256^3 case alone is 1.3 ms in which nothing else of that case can run).  Results are those of
        execute(), bit for bit (the same calls in the same order per case; only the collection of the last image moves).  The
        reference's batch mode is a process pool over cases (scripts/__init__.py:387-416); this is its one-process form for a
        GPU.  A case that raises is reported by re-raising at its position; the case in flight behind it is abandoned."""
        prev, cur = None, None
        try:
            for case in cases:
                lab = case[2] if len(case) > 2 else label
                steps = self._executeSteps(case[0], case[1], lab, label_channel, voxelBased)
                try:
                    next(steps)                      # head of this case: loaded, uploaded, filtered, its FIRST derived image queued
                    cur = (steps, None)
                except StopIteration as done:        # (nothing was left pending: e.g. no derived image)
                    cur = (None, done.value)
                if prev is not None:
                    last, prev = prev, None
                    yield self._finishSteps(last)    # the previous case's last image: its kernels ran under this case's head
                if cur[0] is not None:
                    try:
                        next(cur[0])                 # the body: every image but the last collected, the last one left queued
                    except StopIteration as done:
                        cur = (None, done.value)
                prev, cur = cur, None
            if prev is not None:
                last, prev = prev, None
                yield self._finishSteps(last)
        finally:
            for state in (prev, cur):
                if state is not None and state[0] is not None:
                    state[0].close()                 # (abandons what the unfinished case queued)

    @staticmethod
    def _finishSteps(state):
        steps, value = state
        if steps is None:
            return value
        try:
            while True:
                next(steps)
        except StopIteration as done:
            return done.value

    def _executeSteps(self, imageFilepath, maskFilepath, label=None, label_channel=None, voxelBased=False):
        """execute() as a generator that yields TWICE -- when the case's first derived image is queued (executeMany collects the
        previous case's last image there: at most two images of two cases in flight, the library's tickets are a ring of four) and
        when everything is queued or collected except the values of the last derived image -- and returns what execute() returns"""
        s = self.settings.copy()
        if label is not None:
            s["label"] = label
        if label_channel is not None:
            s["label_channel"] = label_channel
        label = s.get("label", 1)

        kernelRadius = 0
        if voxelBased:
            s["voxelBased"] = True
            kernelRadius = s.get("kernelRadius", 1)
        out = collections.OrderedDict()
        image, mask = self.loadImage(imageFilepath, maskFilepath, **s)
        # Segment-based extraction keeps the case in HBM: one upload of image + mask, then filters, crop,
        # discretisation and the matrices all work on device tensors (`deviceResident: False` restores the
        # host-array route of the reference; voxel-based extraction batches host coordinates and uses it too).
        on_dev = (bool(s.get("deviceResident", True)) and not voxelBased
                  and getattr(backend.get(), "DEVICE_TENSORS", False))
        s["deviceResident"] = on_dev
        info = collections.OrderedDict()

        def describe(stage, img, msk):
            """diagnostics_Image-<stage>_* / diagnostics_Mask-<stage>_* (generalinfo.py:97-190, the subset that needs
            no SimpleITK label statistics)"""
            on = msk.on_device or (on_dev and stage != "original")
            r = imageoperations.roiTensor(msk, label) if on else (msk.array == label)
            try:
                blo, bhi = imageoperations.boundingBox(r)
            except ValueError:
                raise ValueError("Label (%g) not present in mask" % label)
            info["diagnostics_Image-%s_Spacing" % stage] = img.GetSpacing()
            info["diagnostics_Image-%s_Size" % stage] = img.GetSize()
            info["diagnostics_Mask-%s_Spacing" % stage] = msk.GetSpacing()
            info["diagnostics_Mask-%s_Size" % stage] = msk.GetSize()
            info["diagnostics_Mask-%s_BoundingBox" % stage] = tuple(int(v) for v in blo[::-1]) + \
                tuple(int(v) for v in (bhi - blo + 1)[::-1])
            info["diagnostics_Mask-%s_VoxelNum" % stage] = int(r.sum())

        if s.get("additionalInfo", True):
            describe("original", image, mask)
        if s.get("normalize", False):                  # featureextractor.py:432-433: before anything else sees the image
            image = imageoperations.normalizeImage(image, **s)
        # :436-440 (the reference's loadImage only resamples when BOTH the spacing and an interpolator are given)
        if s.get("resampledPixelSpacing") is not None and s.get("interpolator") is not None:
            if not np.any(mask.array == label):
                raise ValueError("Label (%g) not present in mask" % label)
            image, mask = imageoperations.resampleImage(image, mask, **s)
            if s.get("additionalInfo", True):
                describe("interpolated", image, mask)
        elif s.get("preCrop", False):
            # featureextractor.py:471-482: work on the ROI's bounding box grown by padDistance from here on (a speed-up
            # the user opts into; filters then see the crop's borders instead of the image's)
            image, mask = imageoperations.cropToTumorMask(image, mask, label, padDistance=s.get("padDistance", 5),
                                                          deviceResident=on_dev, alignRows=False)
        if s.get("resegmentRange") is not None:
            if not np.any(mask.array == label):
                raise ValueError("Label (%g) not present in mask" % label)
            mask = imageoperations.resegmentMask(image, mask, **s)
            if s.get("additionalInfo", True):
                describe("resegmented", image, mask)
        roi = imageoperations.roiTensor(mask, label) if on_dev else (mask.array == label)
        try:
            lo, hi = imageoperations.boundingBox(roi)
        except ValueError:
            raise ValueError("Label (%g) not present in mask" % label)
        mask._derived[("bbox", label)] = (lo, hi)
        nroi = int(roi.sum())
        ndims = int(np.sum(hi - lo + 1 > 1))
        if ndims < s.get("minimumROIDimensions", 2):
            raise ValueError("mask has too few dimensions (number of dimensions %d, minimum required %d)"
                             % (ndims, s.get("minimumROIDimensions", 2)))
        if s.get("minimumROISize") is not None and nroi <= s["minimumROISize"]:
            raise ValueError("Size of the ROI is too small (minimum size: %g)" % s["minimumROISize"])
        if s.get("additionalInfo", True):
            out["diagnostics_Versions_PyRadiomicsAMD"] = __version__
            out["diagnostics_Versions_Numpy"] = np.__version__
            out["diagnostics_Configuration_Settings"] = {k: v for k, v in s.items()}
            out["diagnostics_Configuration_EnabledImageTypes"] = dict(self.enabledImagetypes)
            out.update(info)
        gens = []
        for imageType, custom in self.enabledImagetypes.items():
            args = s.copy()
            args.update(custom)
            gens = chain(gens, _IMAGE_TYPES[imageType](image, mask, **args))
        # Look-ahead: image i + 1 is cropped and its discretisation QUEUED (prebinDevice: no round trip) right behind the device
        # classes of image i, and the host collects image i - 1 after that -- the GPU works through a queue while the host does
        # its Python, and the one round trip of a binning (ROI min / max -> number of levels) has passed by the time the classes
        # of image i + 1 are constructed; results keep the reference's order (featureextractor.py:371-396: image after image).
        from .base import RadiomicsFeaturesBase
        bins = on_dev and bool(self.enabledFeatures)      # (every class discretises: first order for its entropy / uniformity)
        it = iter(gens)

        def fetch():
            try:
                derived, typeName, kw = next(it)
            except StopIteration:
                return None
            cimg, cmask = imageoperations.cropToTumorMask(derived, mask, label, padDistance=kernelRadius, deviceResident=on_dev)
            if bins:
                RadiomicsFeaturesBase.prebinDevice(cimg, cmask, **kw)
            return cimg, cmask, typeName, kw

        pending, nxt, head_done = None, None, False
        try:
            nxt = fetch()
            while nxt is not None:
                cur, nxt = nxt, None
                try:
                    started = self._startFeatures(cur[0], cur[1], cur[2], **cur[3])
                except BaseException:
                    RadiomicsFeaturesBase.dropPrebinned(cur[0])
                    raise
                prev, pending = pending, started
                if prev is None and not head_done:
                    head_done = True
                    yield                          # (executeMany: the previous case's last image is collected here)
                try:
                    nxt = fetch()
                except BaseException:
                    if prev is not None:
                        self._abandonFeatures(prev)
                    raise
                if prev is not None:
                    out.update(self._finishFeatures(prev))      # (abandons `prev` itself when it fails)
            if pending is not None:
                yield                              # (executeMany: the next case's head runs here; GeneratorExit abandons `pending`)
                last, pending = pending, None
                out.update(self._finishFeatures(last))
        except BaseException:
            # an exception between "queued" and "collected" (the next image's filter / crop / binning, a host class): the
            # queued image still holds a ticket of the library's in-flight table (4 per thread) and its side streams hold
            # tensors -- retire them, or every later case on this (persistent batch worker) thread fails
            if pending is not None:
                self._abandonFeatures(pending)
            if nxt is not None:
                RadiomicsFeaturesBase.dropPrebinned(nxt[0])
            raise
        return out

    @staticmethod
    def _abandonFeatures(started):
        """waits for and discards everything _startFeatures queued for one derived image (never raises; idempotent: a token
        that was already waited for -- `tokens[i] is None` -- is not waited for a second time)"""
        fcs, queued, tokens, _name = started
        cm = queued[0].cMatrices if queued else (fcs[0][1].cMatrices if fcs else None)
        for i, wait in ((0, "segment_image_wait"), (1, "segment_wait")):
            tok, tokens[i] = tokens[i], None
            if tok is not None and cm is not None:
                try:
                    getattr(cm, wait)(tok)
                except Exception:          # noqa: BLE001 -- the original exception is the one to report
                    pass
        for fc in queued:
            try:
                fc.dropEnqueued()
            except Exception:              # noqa: BLE001
                pass

    def computeFeatures(self, image, mask, imageTypeName, **kwargs):
        """featureextractor.py:560-604"""
        return self._finishFeatures(self._startFeatures(image, mask, imageTypeName, **kwargs))

    def _startFeatures(self, image, mask, imageTypeName, **kwargs):
        """first half of computeFeatures: the feature classes of one derived image constructed (binning), the device
        classes' kernels queued; -> state for _finishFeatures"""
        classes = getFeatureClasses()
        fcs = []
        for cname, fnames in self.enabledFeatures.items():
            if cname not in classes:
                continue
            fc = classes[cname](image, mask, **kwargs)
            for f in fnames or []:
                fc.enableFeatureByName(f)
            fcs.append((cname, fc))
        # Case pipeline: every class whose work runs on the device without a host round trip (GLCM, GLRLM, GLSZM, GLDM, NGTDM,
        # first order of float images) queues all of its kernels first, on side streams; the rest (first order of the
        # int16 original: exact histogram) runs on the main stream while those queues drain; then ONE wait per image.  The reference evaluates class after
        # class (featureextractor.py:560-604), each with its own round trips.
        cm = fcs[0][1].cMatrices if fcs else None
        side = hasattr(cm, "segment_queue") and fcs[0][1].deviceResident
        queued, names, ticket = [], [], [None]
        try:
            return self._queueFeatures(fcs, cm, side, queued, names, ticket, imageTypeName, kwargs)
        except BaseException:
            # a class that fails to queue after the image call went through: retire that call's ticket before re-raising
            self._abandonFeatures((fcs, queued, [ticket[0], None], imageTypeName))
            raise

    def _queueFeatures(self, fcs, cm, side, queued, names, ticket, imageTypeName, kwargs):
        # one library call queues every class the fused kernels cover (cMatrices.segment_image_enqueue) ...
        image_token = None
        if side and hasattr(cm, "segment_image_enqueue"):
            reqs = {}
            for cname, fc in fcs:
                rq = fc.imageRequest() if hasattr(fc, "imageRequest") else None
                if rq is not None and rq[0] not in reqs:
                    reqs[rq[0]] = (fc, rq[1])
            if reqs:
                any_fc = next(iter(reqs.values()))[0]
                levels = getattr(any_fc, "discretizedImageArray", any_fc.imageArray)
                image_token, finishes = cm.segment_image_enqueue(
                    levels, any_fc.maskArray, any_fc.coefficients["Ng"], any_fc.coefficients["Ns"],
                    {k: rq for k, (fc, rq) in reqs.items()}, force2D=kwargs.get("force2D", False),
                    force2Ddimension=kwargs.get("force2Ddimension", 0))
                ticket[0] = image_token
                for k, fin in finishes.items():
                    reqs[k][0].takeEnqueued(fin)
                    queued.append(reqs[k][0])
        # ... the others queue on their own (or compute in _finishFeatures)
        for cname, fc in fcs:
            if fc in queued:
                continue
            with (cm.segment_queue(cname) if side else contextlib.nullcontext()):
                if fc.enqueue():
                    queued.append(fc)
                    names.append(cname)
        token = cm.segment_mark(names) if names else None
        return fcs, queued, [image_token, token], imageTypeName

    def _finishFeatures(self, started):
        """second half of computeFeatures: the host-side classes evaluated, the queued ones waited for and collected"""
        fcs, queued, tokens, imageTypeName = started
        out = collections.OrderedDict()
        try:
            # (everything below runs under the guard: a failing host class, a failing wait -- the library's generation check, a
            # HIP error -- or a failing collect leaves no ticket un-retired and no queued class holding its results: ADVICE r4)
            values = {cname: fc.execute() for cname, fc in fcs if fc not in queued}
            if queued:
                cm = queued[0].cMatrices
                ok = True
                for i, wait in ((0, cm.segment_image_wait), (1, cm.segment_wait)):
                    tok, tokens[i] = tokens[i], None          # (marked as waited first: the abandon path must not wait again)
                    if tok is not None:
                        ok = wait(tok) and ok
                if not ok:
                    for fc in queued:      # a level outside [1, Ng]: the synchronous route raises what the reference raises
                        fc.dropEnqueued()
            for cname, fc in fcs:
                for fname, value in (values[cname] if cname in values else fc.execute()).items():
                    out["%s_%s_%s" % (imageTypeName, cname, fname)] = value
        except BaseException:
            self._abandonFeatures(started)
            raise
        return out
