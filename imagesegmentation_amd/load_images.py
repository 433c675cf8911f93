"""Image and surface-data loaders with the reference's file conventions
(load_images.py:8-20, 30-47).

`data_root` is the directory holding ``disparity/``, ``RGB/`` and ``annotation/``.
The reference reads ``data/`` relative to the working directory; that is tried
first, then ``$FS_DATA_ROOT``, then the OSD copy shipped with this repository.
"""
import os
import re

import numpy as np

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data_root = None      # set to override the search below
out_root = "out"      # find_surfaces writes out/{index}/*.npy relative to the CWD (find_surfaces.py:537-545)


def _data_root():
    if data_root is not None:
        return data_root
    for cand in ("data", os.environ.get("FS_DATA_ROOT"), os.path.join(_REPO, "data", "osd")):
        if cand and os.path.isdir(os.path.join(cand, "disparity")):
            return cand
    raise FileNotFoundError("no data directory with disparity/, RGB/ and annotation/ (tried ./data, $FS_DATA_ROOT, "
                            f"{os.path.join(_REPO, 'data', 'osd')})")


def sorted_alphanumeric(data):
    """Natural sort: digit runs compare as numbers (load_images.py:8-11)."""
    def key(name):
        return [int(tok) if tok.isdigit() else tok.lower() for tok in re.split("([0-9]+)", name)]
    return sorted(data, key=key)


def load_image(index):
    """-> (depth uint16 [H,W], rgb uint8 [H,W,3], annotation uint8 [H,W]); `index`
    selects from the naturally sorted listing of disparity/ (load_images.py:13-20)."""
    from PIL import Image
    root = _data_root()
    name = sorted_alphanumeric(os.listdir(os.path.join(root, "disparity")))[index]
    return tuple(np.asarray(Image.open(os.path.join(root, sub, name))) for sub in ("disparity", "RGB", "annotation"))


def save_surface_information(index, data):
    """The seven files find_surfaces() writes per image (find_surfaces.py:537-545).
    `Q.npy` holds the surface labels, `depth.npy` the raw depth."""
    d = os.path.join(out_root, str(index))
    os.makedirs(d, exist_ok=True)
    for fname, key in (("Q", "surfaces"), ("depth", "depth"), ("angles", "angles"), ("vectors", "vectors"),
                       ("patches", "patches"), ("points", "points_3d"), ("edges", "depth_edges")):
        np.save(os.path.join(d, fname + ".npy"), data[key])


def load_image_and_surface_information(index, with_lab=True):
    """Reads back what find_surfaces() saved (load_images.py:30-43); num_surfaces is
    recomputed as max(surfaces)+1 as the reference does.  Like the reference it
    prints a hint and exits when the files are missing."""
    try:
        data = {}
        data["depth"], data["rgb"], data["annotation"] = load_image(index)
        if with_lab:
            from .utils import rgb_to_Lab
            data["lab"] = rgb_to_Lab(data["rgb"])
        d = os.path.join(out_root, str(index))
        for fname, key in (("Q", "surfaces"), ("depth", "depth"), ("angles", "angles"), ("patches", "patches"),
                           ("points", "points_3d"), ("edges", "depth_edges"), ("vectors", "vectors")):
            data[key] = np.load(os.path.join(d, fname + ".npy"))
        data["num_surfaces"] = int(np.max(data["surfaces"]) + 1)
        return data
    except (OSError, IndexError, ValueError):
        print("The index is not available or the surfaces for this image have not been detected yet. "
              "Run find_surfaces_for_indices(..., save_data=True) first so that the surface data can be loaded.")
        raise SystemExit
