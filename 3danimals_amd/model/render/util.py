"""Small math helpers on the hot path -- the subset of /root/reference/model/render/util.py that the
replaced modules (and their callers) use.  Plain torch; fused into kernels where it matters."""
import numpy as np
import torch


def dot(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return torch.sum(x * y, -1, keepdim=True)


def reflect(x: torch.Tensor, n: torch.Tensor) -> torch.Tensor:
    return 2 * dot(x, n) * n - x


def length(x: torch.Tensor, eps: float = 1e-20) -> torch.Tensor:
    """sqrt(max(x.x, eps)): the clamp keeps the gradient finite at 0 (reference util.py:28-29)."""
    return torch.sqrt(torch.clamp(dot(x, x), min=eps))


def safe_normalize(x: torch.Tensor, eps: float = 1e-20) -> torch.Tensor:
    return x / length(x, eps)


def to_hvec(x: torch.Tensor, w: float) -> torch.Tensor:
    return torch.nn.functional.pad(x, pad=(0, 1), mode="constant", value=w)


def perspective(fovy=0.7854, aspect=1.0, n=0.1, f=1000.0, device=None):
    """gluPerspective with y flipped so images come out upright (reference util.py:189-194)."""
    y = np.tan(fovy / 2)
    return torch.tensor([[1 / (y * aspect), 0, 0, 0], [0, 1 / -y, 0, 0], [0, 0, -(f + n) / (f - n), -(2 * f * n) / (f - n)], [0, 0, -1, 0]],
                        dtype=torch.float32, device=device)


def fovx_to_fovy(fovx, aspect):
    return np.arctan(np.tan(fovx / 2) / aspect) * 2.0


def focal_length_to_fovy(focal_length, sensor_height):
    return 2 * np.arctan(0.5 * sensor_height / focal_length)


def scale_img_nhwc(x: torch.Tensor, size, mag="bilinear", min="area") -> torch.Tensor:
    """Resize an NHWC image (reference util.py:97-108); only reached when spp > 1."""
    assert (x.shape[1] >= size[0] and x.shape[2] >= size[1]) or (x.shape[1] < size[0] and x.shape[2] < size[1])
    y = x.permute(0, 3, 1, 2)
    if x.shape[1] > size[0] and x.shape[2] > size[1]:
        y = torch.nn.functional.interpolate(y, size, mode=min)
    elif mag in ("bilinear", "bicubic"):
        y = torch.nn.functional.interpolate(y, size, mode=mag, align_corners=True)
    else:
        y = torch.nn.functional.interpolate(y, size, mode=mag)
    return y.permute(0, 2, 3, 1).contiguous()


def avg_pool_nhwc(x: torch.Tensor, size) -> torch.Tensor:
    y = torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), size)
    return y.permute(0, 2, 3, 1).contiguous()


def checkerboard(res, checker_size) -> np.ndarray:
    tiles_y, tiles_x = (res[0] + checker_size * 2 - 1) // (checker_size * 2), (res[1] + checker_size * 2 - 1) // (checker_size * 2)
    check = np.kron([[1, 0] * tiles_x, [0, 1] * tiles_x] * tiles_y, np.ones((checker_size, checker_size))) * 0.33 + 0.33
    check = check[: res[0], : res[1]]
    return np.stack((check, check, check), axis=-1)
