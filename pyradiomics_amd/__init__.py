"""pyradiomics_amd -- MI355X-native texture-matrix engine behind pyradiomics' `cMatrices` operator API.

Only what the hot path needs lives here:
    csrc/          hand-written HIP kernels (gfx950) + the C ABI of include/pyradiomics_amd.h
    cmatrices      drop-in for `radiomics._cmatrices` (ctypes over the C ABI)
    engine         device-resident entry points (torch tensors in HBM) used by bench / batch / voxel drivers
Nothing in this package falls back to the CPU; importing `cmatrices` without the built library raises.
"""
__version__ = "0.1.0"
