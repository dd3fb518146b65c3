"""``utils.graphics_utils`` of the 3DGS python layer, backed by gsworld_amd.camera."""
import math
from typing import NamedTuple

import numpy as np

from gsworld_amd.camera import get_projection_matrix as getProjectionMatrix  # noqa: F401
from gsworld_amd.camera import get_world2view2 as getWorld2View2  # noqa: F401


class BasicPointCloud(NamedTuple):
    points: np.array
    colors: np.array
    normals: np.array


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))
