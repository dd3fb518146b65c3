"""CPU restatement of fused-ssim (rahul-goel/fused-ssim ssim.cu + fused_ssim/__init__.py) with
plain ``torch.nn.functional.conv2d``: 11x11 window = outer product of the normalised 11-tap sigma-1.5 Gaussian,
zero padding 5 ("same") or cropped 5 px per side ("valid"), C1 = 0.01^2, C2 = 0.03^2, mean over all elements.
TEST INFRASTRUCTURE ONLY; the gradient reference is autograd of this definition."""
import torch
import torch.nn.functional as F


def gaussian_window(dtype=torch.float64):
    k = torch.arange(11, dtype=torch.float64) - 5
    g = torch.exp(-(k * k) / (2 * 1.5 * 1.5))
    g = g / g.sum()
    return g.to(dtype)


def ssim_map(img1, img2, C1=0.01 ** 2, C2=0.03 ** 2):
    B, CH, H, W = img1.shape
    g = gaussian_window(img1.dtype)
    w2d = (g[:, None] * g[None, :])[None, None].repeat(CH, 1, 1, 1)

    def conv(x):
        return F.conv2d(x, w2d, padding=5, groups=CH)

    mu1, mu2 = conv(img1), conv(img2)
    s11 = conv(img1 * img1) - mu1 * mu1
    s22 = conv(img2 * img2) - mu2 * mu2
    s12 = conv(img1 * img2) - mu1 * mu2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))


def ssim(img1, img2, padding="same"):
    m = ssim_map(img1, img2)
    if padding == "valid":
        m = m[:, :, 5:-5, 5:-5]
    return m.mean()
