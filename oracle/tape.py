"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Record / replay of the global-RNG draws the hot path makes (`torch.randn` for z and Noise,
`torch.rand` for the WGAN-GP alpha), so the reference, the oracle and the HIP product can be
driven with identical random inputs regardless of device or RNG implementation.
"""
from __future__ import annotations

from typing import List, Optional

import torch


def _shape_of(size) -> tuple:
    if len(size) == 1 and not isinstance(size[0], int):
        return tuple(size[0])
    return tuple(int(s) for s in size)


class RandTape:
    """Context manager.  ``RandTape()`` records every torch.randn/torch.rand result (as fp32 CPU);
    ``RandTape(tape)`` replays them in order, checking shapes, on the requested device/dtype."""

    def __init__(self, tape: Optional[List[torch.Tensor]] = None):
        self.replay = tape is not None
        self.tape: List[torch.Tensor] = list(tape) if tape is not None else []
        self.pos = 0

    def _wrap(self, fn):
        def wrapped(*size, **kw):
            if self.replay:
                kw.pop("generator", None)
                t = self.tape[self.pos]
                self.pos += 1
                want = _shape_of(size)
                if tuple(t.shape) != want:
                    raise RuntimeError(f"tape entry {self.pos - 1} has shape {tuple(t.shape)}, draw wants {want}")
                return t.to(device=kw.get("device"), dtype=kw.get("dtype") or t.dtype)
            out = fn(*size, **kw)
            self.tape.append(out.detach().to("cpu", torch.float32).clone())
            return out

        return wrapped

    def __enter__(self):
        self._randn, self._rand = torch.randn, torch.rand
        torch.randn = self._wrap(self._randn)
        torch.rand = self._wrap(self._rand)
        return self

    def __exit__(self, *exc):
        torch.randn, torch.rand = self._randn, self._rand
        return False
