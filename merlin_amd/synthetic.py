"""Synthetic MERFISH image stacks generated on the GPU with torch (inputs for
bench.py and the full-size tests; not part of the decode path).

Parameters follow merlin/util/simulator.py:18-28: Gaussian PSF sigma 1.2 px,
Poisson background 100; spot amplitudes 1000-5000 on the on-bits of a
uniformly drawn codeword."""
import numpy as np
import torch


def to_uint16(x: torch.Tensor) -> torch.Tensor:
    """float/int tensor with values in [0, 65535] -> torch.uint16."""
    xi = x.to(torch.int32)
    xi = torch.where(xi >= 32768, xi - 65536, xi).to(torch.int16)
    return xi.view(torch.uint16)


def synthetic_fov(barcodes: np.ndarray, Z: int, H: int, W: int,
                  spotsPerPlane: int, seed: int, device,
                  background: float = 100.0, psfSigma: float = 1.2,
                  amplitude=(1000.0, 5000.0)) -> torch.Tensor:
    """uint16 [B, Z, H, W] on ``device``."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    cb = torch.as_tensor(np.asarray(barcodes), device=dev)
    N, B = cb.shape
    img = torch.empty((B, Z, H, W), dtype=torch.float32, device=dev)
    for b in range(B):
        img[b] = torch.poisson(torch.full((Z, H, W), background, device=dev),
                               generator=gen)
    r = 4
    oy, ox = torch.meshgrid(torch.arange(-r, r + 1, device=dev),
                            torch.arange(-r, r + 1, device=dev), indexing='ij')
    oy, ox = oy.reshape(1, -1), ox.reshape(1, -1)
    flat = img.view(B, Z, H * W)
    for z in range(Z):
        n = spotsPerPlane
        code = torch.randint(0, N, (n,), device=dev, generator=gen)
        cy = torch.rand(n, device=dev, generator=gen) * (H - 2 * r - 1) + r
        cx = torch.rand(n, device=dev, generator=gen) * (W - 2 * r - 1) + r
        amp = torch.rand(n, device=dev, generator=gen) \
            * (amplitude[1] - amplitude[0]) + amplitude[0]
        iy, ix = cy.floor().long(), cx.floor().long()
        py = iy[:, None] + oy
        px = ix[:, None] + ox
        psf = torch.exp(-((py - cy[:, None]) ** 2 + (px - cx[:, None]) ** 2)
                        / (2 * psfSigma ** 2)) * amp[:, None]
        lin = (py * W + px).reshape(-1)
        on = cb[code].to(torch.float32)                       # [n, B]
        jitter = torch.rand((n, B), device=dev, generator=gen) * 0.6 + 0.7
        w = on * jitter
        for b in range(B):
            flat[b, z].index_put_(
                (lin,), (psf * w[:, b:b + 1]).reshape(-1), accumulate=True)
    return to_uint16(img.round_().clamp_(0, 65535))
