"""Explicit randomness for the oracle (test infrastructure, see oracle/__init__.py).

The reference draws from torch's *global* CPU generator in a fixed order
(SURVEY.md section 8(a) row A1).  Three primitives cover every draw on the path:

  exponential(like)  - the V Exp(1) variates ``torch.multinomial(p, 1)`` consumes:
                       ATen's single-sample path is argmax(p / q), q = empty_like(p).exponential_(1)
                       (reference utils.py:221; checked against torch 2.10.0 in
                       tests/golden/make_golden.py, fixture ``multinomial_equiv``)
  uniform()          - ``torch.rand(1)`` (reference speculative_sampling.py:1978)
  reseed(seed)       - ``torch.manual_seed(random_seed)`` (speculative_sampling.py:1976-1977)
"""
from __future__ import annotations

from typing import List, Tuple

import torch


class TorchGlobalNoise:
    """Draws live from torch's default CPU generator, in the reference's order."""

    def exponential(self, like: torch.Tensor) -> torch.Tensor:
        return torch.empty_like(like, device="cpu").exponential_(1)

    def uniform(self) -> torch.Tensor:
        return torch.rand(1)

    def uniform64(self) -> torch.Tensor:
        """One double uniform: what ATen's with-replacement multinomial draws per sample (oracle/beam_ref.py)."""
        return torch.rand(1, dtype=torch.float64)

    def reseed(self, seed: int) -> None:
        torch.manual_seed(seed)


class RecordingNoise:
    """Wraps a source and keeps every draw, in order, for replay into the HIP path."""

    def __init__(self, inner=None):
        self.inner = inner or TorchGlobalNoise()
        self.events: List[Tuple[str, object]] = []

    def exponential(self, like):
        e = self.inner.exponential(like)
        self.events.append(("exp", e.clone()))
        return e

    def uniform(self):
        r = self.inner.uniform()
        self.events.append(("uni", r.clone()))
        return r

    def uniform64(self):
        r = self.inner.uniform64()
        self.events.append(("uni64", r.clone()))
        return r

    def reseed(self, seed):
        self.inner.reseed(seed)
        self.events.append(("seed", int(seed)))


class RecordedNoise:
    """Replays a recorded stream; a kind mismatch means the RNG contract broke."""

    def __init__(self, events):
        self.events = list(events)
        self.pos = 0

    def _next(self, kind):
        if self.pos >= len(self.events):
            raise RuntimeError(f"noise stream exhausted at draw {self.pos} (wanted {kind})")
        k, v = self.events[self.pos]
        if k != kind:
            raise RuntimeError(f"noise order mismatch at draw {self.pos}: recorded {k}, asked {kind}")
        self.pos += 1
        return v

    def exponential(self, like):
        e = self._next("exp")
        e = torch.as_tensor(e)
        assert e.numel() == like.numel(), (e.shape, like.shape)
        return e.reshape(like.shape).to(like.dtype)

    def uniform(self):
        return torch.as_tensor(self._next("uni"), dtype=torch.float32).reshape(1)

    def uniform64(self):
        return torch.as_tensor(self._next("uni64"), dtype=torch.float64).reshape(1)

    def reseed(self, seed):
        s = self._next("seed")
        assert int(s) == int(seed)

    def exhausted(self) -> bool:
        return self.pos == len(self.events)
