"""Validation of a parameter file / dictionary: the rules of the reference's schema (radiomics/schemas/paramSchema.yaml
checked through pykwalify with radiomics/schemas/schemaFuncs.py, featureextractor.py `_applyParams`) restated as a
table, so that a typo or a value of the wrong type is reported when the extractor is configured and not in the middle
of a batch.  A violation raises ValueError (the reference raises pykwalify's SchemaError); keys the schema does not
know are reported with a warning and passed on -- this package has a few settings of its own."""
from __future__ import annotations

import logging
import numbers

logger = logging.getLogger(__name__)

_INTERPOLATORS = {"sitkNearestNeighbor", "sitkLinear", "sitkBSpline", "sitkGaussian", "sitkLabelGaussian",
                  "sitkHammingWindowedSinc", "sitkCosineWindowedSinc", "sitkWelchWindowedSinc",
                  "sitkLanczosWindowedSinc", "sitkBlackmanWindowedSinc"}          # schemaFuncs.py:22-47
_WEIGHTINGS = ("euclidean", "manhattan", "infinity", "no_weighting")              # schemaFuncs.py:50-62


def _is_int(v):
    return isinstance(v, numbers.Integral) and not isinstance(v, bool)


def _is_float(v):        # pykwalify's "float": a real number that is not a bool, or a string float() accepts ("nan")
    if isinstance(v, bool):
        return False
    if isinstance(v, numbers.Real):
        return True
    if isinstance(v, str):
        try:
            float(v)
            return True
        except ValueError:
            return False
    return False


def _num(kind, lo=None, lo_ex=None, hi=None):
    def check(name, v):
        ok = _is_int(v) if kind == "int" else _is_float(v)
        if not ok:
            raise ValueError("setting %s: %r is not of type %s" % (name, v, kind))
        if isinstance(v, str):
            v = float(v)
        if lo is not None and v < lo:
            raise ValueError("setting %s: %r is below the minimum %r" % (name, v, lo))
        if lo_ex is not None and not v > lo_ex:
            raise ValueError("setting %s: %r must be greater than %r" % (name, v, lo_ex))
        if hi is not None and v > hi:
            raise ValueError("setting %s: %r is above the maximum %r" % (name, v, hi))
    return check


def _bool(name, v):
    if not isinstance(v, bool):
        raise ValueError("setting %s: %r is not of type bool" % (name, v))


def _seq(item):
    def check(name, v):
        if not isinstance(v, (list, tuple)):
            raise ValueError("setting %s: %r is not a sequence" % (name, v))
        for x in v:
            item(name, x)
    return check


def _enum(*values):
    def check(name, v):
        if not isinstance(v, str) or v not in values:
            raise ValueError("setting %s: %r is not one of %s" % (name, v, list(values)))
    return check


def _interpolator(name, v):
    if v is None:
        return
    if isinstance(v, str):
        if v not in _INTERPOLATORS:
            raise ValueError('Interpolator value "%s" not valid, possible values: %s' % (v, sorted(_INTERPOLATORS)))
    elif _is_int(v):
        if v < 1 or v > 10:
            raise ValueError("Intepolator value %d, must be in range of [1-10]" % v)
    else:
        raise ValueError("Interpolator not expected type (str or int)")


def _weighting(name, v):
    if v is None:
        return
    if not isinstance(v, str):
        raise ValueError("WeightingNorm not expected type (str or None)")
    if v not in _WEIGHTINGS:
        raise ValueError('WeightingNorm value "%s" not valid, possible values: %s' % (v, list(_WEIGHTINGS)))


def _wavelet(name, v):
    """schemaFuncs.py:11-19 asks PyWavelets; here: a tabulated name, a (dec_lo, dec_hi) pair or a Wavelet-like object"""
    if isinstance(v, str):
        from .filters import WAVELETS
        if v not in WAVELETS:
            raise ValueError('Wavelet "%s" is not tabulated here %s; pass a (dec_lo, dec_hi) pair instead'
                             % (v, sorted(WAVELETS)))
    elif not (hasattr(v, "dec_lo") or (isinstance(v, (list, tuple)) and len(v) == 2)):
        raise ValueError("Wavelet not expected type (str or (dec_lo, dec_hi))")


# paramSchema.yaml:6-141 (`setting`, also the value type of every imageType entry)
SETTINGS = {
    "minimumROIDimensions": _num("int", lo=1, hi=3), "minimumROISize": _num("int", lo_ex=0),
    "geometryTolerance": _num("float", lo_ex=0), "correctMask": _bool, "additionalInfo": _bool,
    "label": _num("int", lo_ex=0), "label_channel": _num("int", lo=0), "binWidth": _num("float", lo_ex=0),
    "binCount": _num("int", lo_ex=0), "normalize": _bool, "normalizeScale": _num("float", lo_ex=0),
    "removeOutliers": _num("float", lo_ex=0), "resampledPixelSpacing": _seq(_num("float", lo=0)),
    "interpolator": _interpolator, "padDistance": _num("int", lo=0), "distances": _seq(_num("int", lo_ex=0)),
    "force2D": _bool, "force2Ddimension": _num("int", lo=0, hi=2), "resegmentRange": _seq(_num("float")),
    "resegmentMode": _enum("absolute", "relative", "sigma"), "resegmentShape": _bool, "preCrop": _bool,
    "sigma": _seq(_num("float", lo_ex=0)), "start_level": _num("int", lo=0), "level": _num("int", lo_ex=0),
    "wavelet": _wavelet, "gradientUseSpacing": _bool, "lbp2DRadius": _num("float", lo_ex=0),
    "lbp2DSamples": _num("int", lo=1), "lbp2DMethod": _enum("default", "ror", "uniform", "var"),
    "lbp3DLevels": _num("int", lo=1), "lbp3DIcosphereRadius": _num("float", lo_ex=0),
    "lbp3DIcosphereSubdivision": _num("int", lo=0), "voxelArrayShift": _num("int"), "symmetricalGLCM": _bool,
    "weightingNorm": _weighting, "gldm_a": _num("int", lo=0),
}
# paramSchema.yaml:143-160 (`voxelSetting`)
VOXEL_SETTINGS = {"kernelRadius": _num("int", lo_ex=0), "maskedKernel": _bool, "initValue": _num("float"),
                  "voxelBatch": _num("int", lo_ex=0)}
# settings of this package (no reference analogue)
OWN_SETTINGS = {"deviceResident": _bool, "fusedVoxel": _bool, "fusedSegment": _bool, "enqueueSegment": _bool,
                "compactGLSZM": _bool}


def _check_map(where, values, *tables):
    if values is None:
        return
    if not isinstance(values, dict):
        raise ValueError("%s must be a mapping, got %r" % (where, type(values).__name__))
    for name, v in values.items():
        for table in tables:
            if name in table:
                if v is not None or table[name] in (_interpolator, _weighting):
                    table[name](name, v)
                break
        else:
            logger.warning("%s: %r is not a setting of the parameter schema; it is passed on unchecked", where, name)


def validate(params, image_types, feature_classes):
    """`params`: the parsed parameter file; `image_types`: names this package can produce; `feature_classes`:
    {class name: [feature names]}.  Raises ValueError on the first violation (paramSchema.yaml + schemaFuncs.py)."""
    if params is None:
        return
    if not isinstance(params, dict):
        raise ValueError("the parameter file must hold a mapping at its top level")
    unknown = set(params) - {"setting", "voxelSetting", "imageType", "featureClass"}
    if unknown:
        raise ValueError("unknown top-level parameter key(s): %s" % ", ".join(sorted(str(k) for k in unknown)))
    _check_map("setting", params.get("setting"), SETTINGS, OWN_SETTINGS, VOXEL_SETTINGS)
    _check_map("voxelSetting", params.get("voxelSetting"), VOXEL_SETTINGS)
    if "imageType" in params:
        types = params["imageType"]
        if types is None:
            raise ValueError("imageType dictionary cannot be None value")               # schemaFuncs.py:87-89
        for t, custom in types.items():
            _check_map("imageType %s" % t, custom, SETTINGS, OWN_SETTINGS)
    if "featureClass" in params:
        classes = params["featureClass"]
        if classes is None:
            raise ValueError("featureClass dictionary cannot be None value")            # schemaFuncs.py:65-67
        for cname, feats in classes.items():
            if feats is None:
                continue
            if not isinstance(feats, (list, tuple)):
                raise ValueError("Value of feature class %s not expected type (list)" % cname)
            if cname in feature_classes:
                bad = set(feats) - set(feature_classes[cname])
                if bad:
                    raise ValueError("Feature Class %s contains unrecognized features: %s" % (cname, sorted(bad)))
