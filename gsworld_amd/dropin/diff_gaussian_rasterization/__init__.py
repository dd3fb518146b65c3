"""Drop-in ``diff_gaussian_rasterization``: put ``gsworld_amd/dropin`` on ``sys.path`` and
GSWorld's imports (/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:22-26 and, through the
3DGS python layer, ``gaussian_renderer``) resolve to the MI355X rasterizer.  ``SparseGaussianAdam`` is
intentionally absent (see gsworld_amd/rasterizer.py)."""
from gsworld_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, cpu_deep_copy_tuple,  # noqa: F401
                                    rasterize_gaussians)
from gsworld_amd import _C  # noqa: F401
