"""pyradiomics_amd -- MI355X-native engine behind pyradiomics' hot path: the `cMatrices` texture matrices, first-order
statistics, discretisation, resampling and the wavelet / LoG filter stack, with the reference's feature-class,
feature-extractor and command-line surface on top.

    csrc/              hand-written HIP kernels (gfx950) + the C ABI of include/pyradiomics_amd.h
    cmatrices          drop-in for `radiomics._cmatrices` (ctypes over the C ABI; numpy arrays or device tensors)
    engine             device-resident entry points (torch tensors in HBM)
    firstorder, glcm, glrlm, glszm, gldm, ngtdm, base     feature classes (radiomics/<same name>.py)
    imageoperations, filters, image                          binning / crop / resample / normalise, image types, file I/O
    featureextractor   RadiomicsFeatureExtractor (radiomics/featureextractor.py)
    scripts, __main__  `python -m pyradiomics_amd` (radiomics/scripts)
    batch              sharding over torch.distributed ranks
Nothing in this package falls back to the CPU for the matrix path; without the built library / a HIP device the
operator calls raise.
"""
__version__ = "0.1.0"


def __getattr__(name):
    # lazy: importing the package must not load torch / the HIP library
    if name == "RadiomicsFeatureExtractor":
        from .featureextractor import RadiomicsFeatureExtractor
        return RadiomicsFeatureExtractor
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
