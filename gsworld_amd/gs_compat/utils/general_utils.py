"""``utils.general_utils`` helpers of the 3DGS python layer that GaussianModel needs."""
import torch


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def build_rotation(r):
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)


def build_scaling_rotation(s, r):
    return build_rotation(r) * s[:, None, :]


def strip_symmetric(L):
    return torch.stack((L[:, 0, 0], L[:, 0, 1], L[:, 0, 2], L[:, 1, 1], L[:, 1, 2], L[:, 2, 2]), 1)


def PILtoTorch(pil_image, resolution):
    import numpy as np

    resized = pil_image.resize(resolution)
    img = torch.from_numpy(np.array(resized)) / 255.0
    return img.permute(2, 0, 1) if img.dim() == 3 else img.unsqueeze(-1).permute(2, 0, 1)
