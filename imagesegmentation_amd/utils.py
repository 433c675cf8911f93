"""Small host helpers feeding the per-surface statistics (utils.py:5-23 of the reference).

The reference delegates the colour conversion to skimage.color.rgb2lab (version
unpinned, not installed here).  rgb_to_Lab below restates that function's published
definition: sRGB (IEC 61966-2-1) -> linear -> CIE XYZ -> CIE L*a*b* under the D65
illuminant and 2-degree observer, skimage's defaults.  No output of the reference
pins it.
"""
import numpy as np

_RGB2XYZ = np.array([[0.412453, 0.357580, 0.180423],
                     [0.212671, 0.715160, 0.072169],
                     [0.019334, 0.119193, 0.950227]])
_D65_WHITE = np.array([0.95047, 1.0, 1.08883])


def normed(x):
    return x / np.linalg.norm(x)


def rgb_to_Lab(image):
    """uint8 (or float in [0,1]) RGB [..., 3] -> float64 L*a*b* [..., 3]."""
    rgb = np.asarray(image)
    rgb = rgb.astype(np.float64) / 255.0 if rgb.dtype == np.uint8 else rgb.astype(np.float64)
    lin = np.where(rgb > 0.04045, np.power((rgb + 0.055) / 1.055, 2.4), rgb / 12.92)
    xyz = lin @ _RGB2XYZ.T / _D65_WHITE
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    L = 116.0 * f[..., 1] - 16.0
    a = 500.0 * (f[..., 0] - f[..., 1])
    b = 200.0 * (f[..., 1] - f[..., 2])
    return np.stack([L, a, b], axis=-1)


def angles_to_normals(angles):
    """Rotate (0,0,-1) by angles[:,0] about x, then by angles[:,1] about y (utils.py:10-23),
    as closed-form rotations instead of one scipy Rotation per row."""
    a = np.asarray(angles, np.float64)
    a0, a1 = a[:, 0], a[:, 1]
    # Rx(a0) (0,0,-1) = (0, sin a0, -cos a0);  Ry(a1) (x,y,z) = (x cos a1 + z sin a1, y, -x sin a1 + z cos a1)
    y = np.sin(a0)
    z = -np.cos(a0)
    return np.stack([z * np.sin(a1), y, z * np.cos(a1)], axis=-1)
