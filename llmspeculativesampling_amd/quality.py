"""Caller-side quality check of the reference harness (untimed there): mean target log-prob of the generated
suffix, reference evaluation.py:109-132 ``get_score`` (decoder-only branch).  One full target forward through the
engine; the log-softmax/gather/mean is three torch ops on the device (not part of the decode path)."""
from __future__ import annotations

import torch

from .engine import as_specdec_model, check_token_ids


@torch.no_grad()
def get_score(output: torch.Tensor, target_model, input_len: int) -> torch.Tensor:
    m = as_specdec_model(target_model)
    if m.cfg.is_encoder_decoder:
        raise NotImplementedError("encoder-decoder scoring (reference evaluation.py:126-132) is out of scope")
    assert output.dim() == 2 and output.size(0) == 1
    S = output.size(1)
    check_token_ids(output, m.cfg.vocab_size)
    ses = m.new_session(S + 1)
    ids = output[0].to(device=m.device, dtype=torch.int32)
    logits = torch.empty((S, m.cfg.vocab_size), dtype=torch.float32, device=m.device)
    done = 0
    while done < S:                                   # all S rows of logits, max_rows at a time
        n = min(64, S - done)                         # logit rows per call
        ses.forward(ids[done:done + n], n, logits_out=logits[done:done + n])
        done += n
    logp = torch.log_softmax(logits[:-1], dim=-1)
    picked = torch.gather(logp, -1, output[0, 1:, None].to(m.device))
    return picked[input_len - 1:].mean()
