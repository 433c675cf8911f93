"""Mirror of the reference's post_processing module (post_processing.py:4-28): one mean-field
pass that smooths a final segmentation, with the segment labels in the role of the patches."""
import numpy as np

from . import find_surfaces as _fs
from .find_surfaces import *  # noqa: F401,F403  (the reference star-imports find_surfaces here)


def apply_CRF(data, CRF):
    """data["final_surfaces"] <- argmax of one CRF update seeded with it (post_processing.py:4-19).
    CRF is the object get_postprocessing_model() builds; whole frames go through fs_apply_crf."""
    labels = np.asarray(data["final_surfaces"])
    h, w = labels.shape
    ctx = _fs._context(h, w, CRF.kernel_size, params=CRF.params)
    data["final_surfaces"] = ctx.apply_crf(labels, data["depth"], data["rgb"], data["angles"])
    return data


def get_postprocessing_model(kernel_size=7, do_post_processing=True):
    """-> callable(data) -> data (post_processing.py:21-28)."""
    if not do_post_processing:
        return lambda x: x
    size_x, size_y = int(_fs.width / _fs.div_x), int(_fs.height / _fs.div_y)
    CRF = _fs.conv_crf_Gauss(*_fs.load_Gauss_parameters(), kernel_size, size_y, size_x)
    return lambda x: apply_CRF(x, CRF)
