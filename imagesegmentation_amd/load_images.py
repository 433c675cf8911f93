"""Image and surface-data loaders with the reference's file conventions
(load_images.py:8-20, 30-47).

`data_root` is the directory holding ``disparity/``, ``RGB/`` and ``annotation/``.
The reference reads ``data/`` relative to the working directory; that is tried
first, then ``$FS_DATA_ROOT``, then the OSD copy shipped with this repository.
"""
import os
import re
import struct
import zlib

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


def _png_plain(path):
    """Decodes the plain kind of PNG the data set consists of (grey or RGB, 8 or 16 bits, not interlaced, rows filtered with
    None / Sub / Up) with zlib and numpy, which release the interpreter lock for the bulk of the work, so a pool of loader
    threads scales (PIL holds the lock for most of a decode).  Returns None for any other file, 16-bit colour (which PIL
    returns as 8 bits) and damaged streams included: the caller then uses PIL, which decides what such a file is."""
    with open(path, "rb") as f:
        b = f.read()
    if b[:8] != b"\x89PNG\r\n\x1a\n":
        return None
    pos, header, parts = 8, None, []
    while pos + 12 <= len(b):
        (length,), kind = struct.unpack(">I", b[pos:pos + 4]), b[pos + 4:pos + 8]
        if kind == b"IHDR":
            header = struct.unpack(">IIBBBBB", b[pos + 8:pos + 21])
        elif kind == b"IDAT":
            parts.append(b[pos + 8:pos + 8 + length])
        elif kind == b"IEND":
            break
        pos += 12 + length
    if header is None:
        return None
    w, h, bits, colour, compression, filtering, interlace = header
    if bits not in (8, 16) or colour not in (0, 2) or compression or filtering or interlace or (bits == 16 and colour == 2):
        return None
    channels = 3 if colour == 2 else 1
    bpp = channels * bits // 8                      # bytes per pixel: the distance the Sub filter looks back
    try:
        raw = np.frombuffer(zlib.decompress(b"".join(parts)), np.uint8)
    except zlib.error:
        return None
    if raw.size != h * (1 + w * bpp):
        return None
    raw = raw.reshape(h, 1 + w * bpp)
    filters = raw[:, 0]
    if filters.max(initial=0) > 2:
        return None                                 # Average / Paeth rows: a per-byte recurrence, left to PIL
    x = raw[:, 1:].reshape(h, w, bpp)
    sub = filters == 1
    if sub.all():
        x = np.cumsum(x, axis=1, dtype=np.uint8)    # Sub: byte + the same byte of the pixel to the left, modulo 256
    else:
        x = x.copy()
        if sub.any():
            x[sub] = np.cumsum(x[sub], axis=1, dtype=np.uint8)
        for r in np.nonzero(filters == 2)[0]:       # Up: byte + the byte above (the finished row), in row order
            if r > 0:
                x[r] += x[r - 1]
    if bits == 16:
        x = x.reshape(h, w, channels, 2)
        x = (x[..., 0].astype(np.uint16) << 8) | x[..., 1]
    return x.reshape(h, w) if channels == 1 else x.reshape(h, w, channels)


def read_png(path):
    """numpy array of a PNG file: what np.asarray(PIL.Image.open(path)) gives."""
    img = _png_plain(path)
    if img is None:
        from PIL import Image
        img = np.asarray(Image.open(path))
    return img


def load_image(index):
    """-> (depth uint16 [H,W], rgb uint8 [H,W,3], annotation uint8 [H,W]); `index`
    selects from the naturally sorted listing of disparity/ (load_images.py:13-20)."""
    root = _data_root()
    name = sorted_alphanumeric(os.listdir(os.path.join(root, "disparity")))[index]
    return tuple(read_png(os.path.join(root, sub, name)) for sub in ("disparity", "RGB", "annotation"))


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
