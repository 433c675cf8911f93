"""Mirror of the IoU evaluation of the reference's evaluate_models module (evaluate_models.py:15-68).
The per-pixel confusion counts come from fs_overlap_counts; the rest is arithmetic on the small
[detections, annotations] matrix, as in the reference."""
import numpy as np

from . import find_surfaces as _fs

min_annotation_overlap_threshold = 0.2


def extract_IoU_information(annotation, detected_objects):
    """-> (num_valid_annotations, num_valid_detections, IoU_max_annotations [num_annotations - 1]):
    for every ground-truth object (label >= 1) the largest IoU with any detected segment
    (evaluate_models.py:19-58)."""
    annotation = np.asarray(annotation, np.int64)
    detected_objects = np.asarray(detected_objects, np.int64)
    num_detections = int(detected_objects.max()) + 1
    num_annotations = int(annotation.max()) + 1
    h, w = annotation.shape
    overlap = _fs._context(h, w, _fs.standard_kernel_size).overlap_counts(annotation, detected_objects, num_annotations, num_detections)
    count_detections = overlap.sum(axis=1)
    count_annotations = overlap.sum(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        IoU = overlap / (count_annotations[None, :] + count_detections[:, None] - overlap)
        detection_annotation_overlap = overlap.sum(axis=-1) / count_detections
    # the reference keeps the running maximum with `>` from 0, so NaN entries (empty label pairs) never win
    IoU_max_annotations = np.zeros(num_annotations - 1)
    if num_annotations > 1:
        IoU_max_annotations = np.max(np.where(np.isnan(IoU[:, 1:]), 0.0, IoU[:, 1:]), axis=0, initial=0.0)
    detection_annotation_overlap[np.isnan(detection_annotation_overlap)] = 0
    num_valid_detections = float(np.sum(detection_annotation_overlap > min_annotation_overlap_threshold))
    num_valid_annotations = float(np.sum(count_annotations != 0))
    return num_valid_annotations, num_valid_detections, IoU_max_annotations


def evaluate_Avg_max_IoU(evaluation_inputs):
    """Mean over images of the mean maximal IoU per ground-truth object (evaluate_models.py:61-68)."""
    total = 0
    for annotation, detected_objects in evaluation_inputs:
        _, _, values = extract_IoU_information(np.asarray(annotation).astype("int64"), detected_objects)
        sample = sum(values) / len(values)
        print(sample)
        total += sample
    return total / len(evaluation_inputs)
