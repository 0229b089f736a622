"""The image dump of the training loop (train_celeba.py:162-168): 100 samples of the fixed latents as a 10 x 10 grid,
un-normalised to uint8 and written as PNG.  The reference calls torchvision.io.write_png; torchvision is not in this
image, so the encoder is the few lines PNG needs (zlib + CRC32 from the standard library, filter type 0).  Host-side,
once per `log_img_interval` iterations."""
from __future__ import annotations

import struct
import zlib

import torch
from torch import Tensor

from .utils import unnormalize_img


def sample_grid(fakes: Tensor, rows: int = 10, cols: int = 10) -> Tensor:
    """(rows*cols, C, H, W) in [-1, 1] -> uint8 (C, rows*H, cols*W): train_celeba.py:166-167 as written there."""
    fakes = fakes.float().cpu().unflatten(0, (rows, cols)).permute(2, 0, 3, 1, 4).flatten(1, 2).flatten(-2, -1)
    return unnormalize_img(fakes)


def encode_png(img: Tensor) -> bytes:
    """uint8 (C, H, W), C = 1 (grey) or 3 (RGB) -> PNG file contents."""
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[0] not in (1, 3):
        raise ValueError(f"encode_png expects a uint8 (1 | 3, H, W) tensor, got {img.dtype} {tuple(img.shape)}")
    c, h, w = img.shape
    rows = img.permute(1, 2, 0).contiguous().cpu().numpy().reshape(h, w * c)
    raw = b"".join(b"\x00" + rows[y].tobytes() for y in range(h))  # filter type 0 in front of every scanline

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    ihdr = struct.pack(">IIBBBBB", w, h, 8, 0 if c == 1 else 2, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def write_png(img: Tensor, path: str) -> None:
    with open(path, "wb") as f:
        f.write(encode_png(img))
