"""Mirror of the model factories of the reference's image_processing_models_GPU module that the stage
after the per-surface statistics uses (image_processing_models_GPU.py:189-212).  The smoothing, normals
and patch models of that module are reached through find_surfaces.get_GPU_models()."""
import numpy as np

from . import find_surfaces as _fs


def _any_context():
    return _fs._context(_fs.height, _fs.width, _fs.standard_kernel_size)


def chi_squared_distances_model_2D(pool_size=(5, 5), strides=(2, 2)):
    """-> callable(histogram [A, B, S]) -> float32 [S, S]: average pooling, then the chi-squared distance
    sum((a - b)^2 / (a + b)) between the pooled histograms of every pair of surfaces
    (image_processing_models_GPU.py:189-200)."""
    return lambda histogram: _any_context().chi_squared_distances(np.asarray(histogram), pool_size, strides)


def chi_squared_distances_model_1D():
    """-> callable(histogram [K, S]) -> float32 [S, S] (image_processing_models_GPU.py:202-212)."""
    return lambda histogram: _any_context().chi_squared_distances(np.asarray(histogram))
