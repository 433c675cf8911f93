"""The task menu's entry into this stage (main.py:25-26 of the reference).  The other tasks of the reference's menu
(object assembly, training, evaluation) are downstream of the files this one writes and are not part of this package.

    python -m imagesegmentation_amd.main          # out/{0..110}/{Q,depth,angles,vectors,patches,points,edges}.npy
"""
from . import find_surfaces


def find_surfaces_for_dataset():
    """All 111 frames of the data set through find_surfaces, results saved under out/ (main.py:25-26)."""
    find_surfaces.find_surfaces_for_indices(range(111), save_data=True, plot_result=False)


if __name__ == "__main__":
    find_surfaces_for_dataset()
