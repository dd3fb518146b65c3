"""``arguments`` of the 3DGS python layer.  GSWorld imports only the ``PipelineParams`` name as a type annotation
(/root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:13,37) and builds its own parameter groups
(/root/reference/gsworld/utils/gs_utils.py:22-106); the four pipeline flags below are the rasterizer's inputs."""


class GroupParams:
    pass


class PipelineParams:
    def __init__(self, parser=None, sentinel=False):
        self.convert_SHs_python = False
        self.compute_cov3D_python = False
        self.debug = False
        self.antialiasing = False
        if parser is not None:
            group = parser.add_argument_group("Pipeline Parameters")
            for key in ("convert_SHs_python", "compute_cov3D_python", "debug", "antialiasing"):
                group.add_argument("--" + key, default=False, action="store_true")

    def extract(self, args):
        group = GroupParams()
        for key in ("convert_SHs_python", "compute_cov3D_python", "debug", "antialiasing"):
            setattr(group, key, getattr(args, key, False))
        return group
