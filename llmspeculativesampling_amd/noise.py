"""Randomness providers for the decode loop.

The reference draws from torch's global generator in a fixed order:
gamma draft samples, one discarded target sample, up to gamma uniforms (each preceded by
``torch.manual_seed(random_seed)`` when that is truthy), one residual/bonus sample.

* ``HostTorchNoise``  parity mode: the variates come from torch's *CPU* generator in exactly that
                      order (``torch.multinomial(p, 1)`` on CPU is argmax(p / Exp(1)^V)) and are
                      uploaded; token ids then match the reference's CPU path under the same seed.
* ``ReplayNoise``     a recorded stream (the golden fixtures), same semantics.
* ``DeviceNoise``     throughput mode: counter-based Philox on the device, nothing crosses PCIe.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


class DeviceNoise:
    """Philox4x32-10 on the GPU, keyed by (seed, draw index)."""
    on_device = True

    def __init__(self, seed: int = 0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.draw = 0

    def next_draws(self, n: int = 1) -> int:
        d = self.draw
        self.draw += n
        return d

    def reseed(self, seed: int) -> None:       # random_seed quirk: every uniform restarts the stream
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.draw = 0


class HostTorchNoise:
    """Live torch CPU generator, uploaded per draw (parity with the reference CPU path)."""
    on_device = False

    def __init__(self, device, generator: Optional[torch.Generator] = None):
        self.device = device
        self.gen = generator            # None = torch's global default generator, like the reference

    # ``dtype``: the dtype of the probability row being sampled.  torch.multinomial draws its noise as
    # empty_like(probs).exponential_(1): a 16-bit row consumes the generator differently from an fp32 one and its
    # variates are rounded to that dtype (they are uploaded as fp32 holding those values).
    def exponential(self, V: int, dtype=torch.float32) -> torch.Tensor:
        e = torch.empty(V, dtype=dtype).exponential_(1, generator=self.gen)
        return e.float().to(self.device, non_blocking=False)

    def skip_exponential(self, V: int, dtype=torch.float32) -> None:
        torch.empty(V, dtype=dtype).exponential_(1, generator=self.gen)

    def exponential_rows(self, rows: int, V: int, dtype=torch.float32) -> torch.Tensor:
        """One draw of shape (rows, V): what torch.multinomial consumes for a (rows, V) input
        (multi_speculative_sampling's width-w samples, kvcache_model.py:283 with multi > 1)."""
        e = torch.empty((rows, V), dtype=dtype).exponential_(1, generator=self.gen)
        return e.float().to(self.device, non_blocking=False)

    def skip_exponential_rows(self, rows: int, V: int, dtype=torch.float32) -> None:
        torch.empty((rows, V), dtype=dtype).exponential_(1, generator=self.gen)

    def uniforms(self, gamma: int, random_seed) -> Tuple[torch.Tensor, object]:
        """gamma uniforms as the reference would draw them if nothing were rejected, plus a token to
        re-align the generator once the accepted count is known."""
        if random_seed:
            # speculative_sampling.py:1976-1978: reseed before every r -> all r equal, state independent of l
            self._seed(random_seed)
            r = torch.rand(1, generator=self.gen)
            return r.repeat(gamma).to(self.device), None
        state = self._get_state()
        r = torch.cat([torch.rand(1, generator=self.gen) for _ in range(gamma)])
        return r.to(self.device), state

    def uniform_one(self) -> torch.Tensor:
        """One ``torch.rand(1)`` (the beam variant draws its uniforms one at a time, speculative_sampling.py:288)."""
        return torch.rand(1, generator=self.gen)

    def uniform64_one(self) -> torch.Tensor:
        """One double uniform: ATen's with-replacement multinomial draws one per sample (sampling/beam.py)."""
        return torch.rand(1, dtype=torch.float64, generator=self.gen)

    def realign(self, token, consumed: int) -> None:
        """The reference stops drawing at the first reject: rewind and draw exactly `consumed`."""
        if token is None:
            return
        self._set_state(token)
        for _ in range(consumed):
            torch.rand(1, generator=self.gen)

    def _seed(self, s):
        if self.gen is None:
            torch.manual_seed(s)
        else:
            self.gen.manual_seed(s)

    def _get_state(self):
        return torch.get_rng_state() if self.gen is None else self.gen.get_state()

    def _set_state(self, st):
        if self.gen is None:
            torch.set_rng_state(st)
        else:
            self.gen.set_state(st)


class ReplayNoise:
    """Replays [("exp", tensor) | ("uni", tensor) | ("seed", int)] recorded from the reference."""
    on_device = False

    def __init__(self, events: List[Tuple[str, object]], device):
        self.events = list(events)
        self.pos = 0
        self.device = device

    def _take(self, kind):
        if self.pos >= len(self.events):
            raise RuntimeError(f"noise stream exhausted at draw {self.pos} (wanted {kind})")
        k, v = self.events[self.pos]
        if k != kind:
            raise RuntimeError(f"noise order mismatch at draw {self.pos}: recorded {k}, asked {kind}")
        self.pos += 1
        return v

    def exponential(self, V: int, dtype=None) -> torch.Tensor:
        e = torch.as_tensor(self._take("exp")).to(torch.float32).reshape(-1)
        assert e.numel() == V
        return e.to(self.device)

    def skip_exponential(self, V: int, dtype=None) -> None:
        self._take("exp")

    def exponential_rows(self, rows: int, V: int, dtype=None) -> torch.Tensor:
        e = torch.as_tensor(self._take("exp")).to(torch.float32)
        assert e.numel() == rows * V, (tuple(e.shape), rows, V)
        return e.reshape(rows, V).contiguous().to(self.device)

    def skip_exponential_rows(self, rows: int, V: int, dtype=None) -> None:
        e = torch.as_tensor(self._take("exp"))
        assert e.numel() == rows * V

    def uniform_one(self) -> torch.Tensor:
        return torch.as_tensor(self._take("uni"), dtype=torch.float32).reshape(1)

    def uniform64_one(self) -> torch.Tensor:
        return torch.as_tensor(self._take("uni64"), dtype=torch.float64).reshape(1)

    def uniforms(self, gamma: int, random_seed):
        # the recording holds only the uniforms the reference actually consumed (it stops at the first
        # reject); pad with 2.0 (always "reject") - a padded slot can only be reached after a reject
        vals = []
        start = self.pos
        while len(vals) < gamma and self.pos < len(self.events):
            k, v = self.events[self.pos]
            if k == "seed":
                assert random_seed and int(v) == int(random_seed)
                self.pos += 1
                continue
            if k != "uni":
                break
            vals.append(float(torch.as_tensor(v).reshape(-1)[0]))
            self.pos += 1
        consumed = len(vals)
        vals += [2.0] * (gamma - len(vals))
        return torch.tensor(vals, dtype=torch.float32).to(self.device), ("replay", start, consumed)

    def realign(self, token, consumed: int) -> None:
        if token is None:
            return
        _, _start, recorded = token
        if recorded != consumed:
            raise RuntimeError(f"accept scan consumed {consumed} uniforms, the recording holds {recorded}")

    def exhausted(self) -> bool:
        return self.pos == len(self.events)
