"""``utils.system_utils`` of the 3DGS python layer: only ``mkdir_p`` is used by GSWorld
(/root/reference/gsworld/mani_skill/utils/wrappers/semantic_3dgs_wrapper.py:21, gsworld/utils/gaussian_merger.py)."""
import os
from errno import EEXIST


def mkdir_p(folder_path):
    try:
        os.makedirs(folder_path)
    except OSError as exc:
        if exc.errno == EEXIST and os.path.isdir(folder_path):
            return
        raise


def searchForMaxIteration(folder):
    saved_iters = [int(fname.split("_")[-1]) for fname in os.listdir(folder)]
    return max(saved_iters)
