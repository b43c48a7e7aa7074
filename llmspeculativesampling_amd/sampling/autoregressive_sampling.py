"""Drop-in for reference sampling/autoregressive_sampling.py:8-61 (``autoregressive_sampling``)."""
from __future__ import annotations

import torch

from .._lib import lib, check
from ..engine import as_specdec_model, _stream, check_token_ids
from ..noise import DeviceNoise, HostTorchNoise
from .kvcache_model import KVCacheModel


@torch.no_grad()
def autoregressive_sampling(x: torch.Tensor, model, N: int, eos_token_id: int, temperature: float = 1,
                            top_k: int = 0, top_p: float = 0, pad_token_id=None, *, rng=None):
    """reference autoregressive_sampling.py:9-61: exactly N tokens unless EOS is drawn (the EOS is kept).
    RNG contract: one ``sample`` per token."""
    assert x.shape[0] == 1
    m = as_specdec_model(model)
    dev = m.device
    V = m.cfg.vocab_size
    L0 = x.shape[1]
    check_token_ids(x, V)
    if rng is None or rng == "host":
        noise = HostTorchNoise(dev)
    elif rng == "device":
        noise = DeviceNoise(seed=int(torch.initial_seed()))
    else:
        noise = rng
    kv = KVCacheModel(m, temperature, top_k, top_p, max_seq=L0 + N + 1, noise=noise, full_history=False)
    kv._ensure(L0 + N + 1)
    seq32 = torch.zeros(L0 + N + 1, dtype=torch.int32, device=dev)
    seq32[:L0] = x[0].to(device=dev, dtype=torch.int32)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    st = _stream()
    n_out = 0
    # the EOS test needs each token on the host (autoregressive_sampling.py:55): one 4-byte read per step
    for i in range(N):
        S = L0 + i
        kv.forward_sample(seq32, S, noise, err)
        n_out += 1
        tok = int(seq32[S])
        if int(err) != 0:
            raise RuntimeError("prob error")
        kv.check_errors(S - 1, S)
        if tok == eos_token_id:
            break
    return seq32[:L0 + n_out].to(torch.int64).unsqueeze(0).to(x.device)
