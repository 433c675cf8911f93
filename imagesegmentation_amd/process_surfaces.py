"""Mirror of the per-surface statistics entry point of the reference's process_surfaces module
(process_surfaces.py:672-689).  The pixel loops (numba in the reference) run in fs_surface_stats on
the GPU; what is left here is arithmetic on S x S or S-sized arrays, in numpy like the reference."""
import numpy as np

from . import find_surfaces as _fs
from . import utils


def determine_even_planes(data):
    """1 for surfaces whose normalised angle histogram is peaked (process_surfaces.py:464-478)."""
    hist = np.swapaxes(data["hist_angle"], 2, 0)                   # [S, 40, 40]
    total = hist.sum(axis=(-1, -2), keepdims=True)
    total[total == 0] = 1
    p = hist / total
    entropy = -np.sum(p * np.log(np.where(p == 0, 1.0, p)), axis=(-1, -2))
    peak = hist.max(axis=(-1, -2))
    planes = np.zeros(data["num_surfaces"])
    planes[peak >= 1.5] = 1.0
    planes[entropy > 2] = 0.0
    planes[peak >= 3] = 1.0
    return planes


def calc_depth_extend_distance(i, j, depth_extends):
    """Largest gap between the (norm, x, y, z) intervals of two surfaces, 0 where they overlap
    (process_surfaces.py:317-329)."""
    d1, d2 = depth_extends[i], depth_extends[j]
    overlap = ((d2[:, 0] <= d1[:, 0]) & (d1[:, 0] <= d2[:, 1])) | ((d2[:, 0] <= d1[:, 1]) & (d1[:, 1] <= d2[:, 1])) | \
              ((d1[:, 0] <= d2[:, 0]) & (d2[:, 1] <= d1[:, 1]))
    gap = np.where(d1[:, 0] > d2[:, 1], d1[:, 0] - d2[:, 1], d2[:, 0] - d1[:, 1])
    gap[overlap] = 0
    return np.max(gap)


def create_depth_extend_distance_matrix(data):
    """Symmetric matrix of calc_depth_extend_distance (process_surfaces.py:552-560): entry (i, j), i < j, is the distance
    of the pair in that order, mirrored below the diagonal; all pairs at once instead of a Python double loop."""
    de = np.asarray(data["depth_extend"])
    d1, d2 = de[:, None], de[None, :]                       # [S,1,4,2], [1,S,4,2]
    with np.errstate(invalid="ignore"):                      # labels without pixels carry (+inf, -inf) extents
        overlap = ((d2[..., 0] <= d1[..., 0]) & (d1[..., 0] <= d2[..., 1])) | ((d2[..., 0] <= d1[..., 1]) & (d1[..., 1] <= d2[..., 1])) | \
                  ((d1[..., 0] <= d2[..., 0]) & (d2[..., 1] <= d1[..., 1]))
        gap = np.where(d1[..., 0] > d2[..., 1], d1[..., 0] - d2[..., 1], d2[..., 0] - d1[..., 1])
    gap = np.where(overlap, 0, gap).astype(de.dtype)
    upper = np.triu(np.max(gap, axis=-1).astype(np.float64), 1)
    return upper + upper.T


def calc_depth_extend_to_depth_extend_distance_ratio(data):
    """Distance over the mean interval length of the pair (process_surfaces.py:664-670)."""
    size = np.sum(np.abs(data["depth_extend"][:, :, 0] - data["depth_extend"][:, :, 1]), axis=-1, keepdims=True) / 4
    pair = (size + size.T) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = data["depth_extend_distances"] / pair
    ratio[np.isnan(ratio)] = 0
    return ratio


def neighbor_lists(neighbor_counts):
    """Per surface the labels j >= 1 with more than 3 shared border pixels (process_surfaces.py:70-77)."""
    linked = np.asarray(neighbor_counts) > 3
    linked[:, 0] = False
    return [np.flatnonzero(row).tolist() for row in linked]


def calculate_nearest_points(point_sets, target_points):
    """For every surface i and every target point of its list, the pixel of the surface's patch closest to
    it (process_surfaces.py:638-654) -> list of int32 [m_i, 2] arrays."""
    return _fs._context(_fs.height, _fs.width, _fs.standard_kernel_size).nearest_points(point_sets, target_points)


def prepare_border_centers(neighbors, border_centers):
    """Per surface the border centres towards its neighbours, (0, 0) if it has none
    (process_surfaces.py:737-746)."""
    result = []
    for i in range(len(neighbors)):
        rows = [border_centers[i][j] for j in neighbors[i]] or [np.asarray([0, 0])]
        result.append(np.asarray(rows, dtype="float32"))
    return result


def extract_information_from_surface_data_and_preprocess_surfaces(data, texture_model):
    """Adds the per-surface statistics to `data` (process_surfaces.py:672-689): patch_points, surfaces
    (relabelled, disconnected parts removed), avg_pos, counts, centroids, texture, lab, hist_color,
    hist_angle, hist_texture, avg_normals, planes, neighbors, border_centers, norm, depth_extend,
    depth_extend_distances, depth_extend_distance_ratio.  texture_model(rgb) -> [H/4, W/4, T] float
    array (the reference's ResNet feature extractor; not part of this build) or None."""
    h, w = np.shape(data["patches"])
    S = int(data["num_surfaces"])
    data["texture"] = None if texture_model is None else texture_model(data["rgb"])
    ctx = _fs._context(h, w, _fs.standard_kernel_size)
    # without a Lab image in the dict the colour conversion (utils.rgb_to_Lab in the reference) runs on the GPU
    st = ctx.surface_stats(data["surfaces"], data["patches"], data["depth_edges"], data["points_3d"], data["angles"], data.get("lab"),
                           data["texture"], S, rgb=data["rgb"])
    if "lab" not in data:
        data["lab"] = st["lab"]
    off = st["patch_point_offsets"]
    data["patch_points"] = [st["patch_points"][off[i]:off[i + 1]] for i in range(S)]
    data["surfaces"] = st["surfaces"]
    data["avg_pos"], data["counts"] = st["avg_pos"], st["counts"]
    data["centroids"] = st["centroids"]
    data["hist_color"], data["hist_angle"], data["hist_texture"] = st["hist_color"], st["hist_angle"], st["hist_texture"]
    data["avg_normals"] = utils.angles_to_normals(st["avg_angles"])
    data["planes"] = determine_even_planes(data)
    data["neighbors"], data["border_centers"] = neighbor_lists(st["neighbor_counts"]), st["border_centers"]
    data["norm"] = st["norm"]
    data["depth_extend"] = st["depth_extend"]
    data["depth_extend_distances"] = create_depth_extend_distance_matrix(data) / 1000
    data["depth_extend_distance_ratio"] = calc_depth_extend_to_depth_extend_distance_ratio(data)
    return data


_pools = {}


def extract_information_for_frames(frames, texture_model, workers=2):
    """The per-frame function above over a list of frames' dicts, from `workers` threads (the loops of the reference's
    callers over a data set's images reach it one frame after the other: process_surfaces.py:748-751,
    assemble_objects_rules.py:129-131).  A frame's results are about twice its inputs in bytes and one call sends the
    inputs up before the results come down; two calls in flight, each thread with a library context of its own, use
    both directions of the link at once.  The threads stay (a context costs more to set up than dozens of frames to
    process).  Fills the dicts in place and returns the list."""
    frames = list(frames)
    if workers <= 1 or len(frames) <= 1:
        for data in frames:
            extract_information_from_surface_data_and_preprocess_surfaces(data, texture_model)
        return frames
    pool = _pools.get(workers)
    if pool is None:
        from concurrent.futures import ThreadPoolExecutor
        pool = _pools[workers] = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="fs-stats")
    list(pool.map(lambda data: extract_information_from_surface_data_and_preprocess_surfaces(data, texture_model), frames))
    return frames
