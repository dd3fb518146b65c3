"""Drop-in ``fused_ssim`` package (``from fused_ssim import fused_ssim``)."""
from gsworld_amd.ssim import FusedSSIMMap, allowed_padding, fused_ssim, fusedssim, fusedssim_backward  # noqa: F401
