"""Drop-in for the hot path of the reference's ``sampling`` package (sampling/__init__.py:1-7).

Every name the reference package exports is importable from here, so the harness's import lines
(evaluation.py:13-14) keep working; the variants SURVEY.md section 8 marks out of scope raise NotImplementedError
when called.  ``beam_speculative_sampling_v2`` (section 8(f) rank 4) is built for extra_sample_cnt == 1; its draft side is
"parity unpinned" (sampling/beam.py)."""
from .speculative_sampling import speculative_sampling
from .autoregressive_sampling import autoregressive_sampling
from .batch import speculative_sampling_batch
from .multi import multi_speculative_sampling
from .beam import beam_speculative_sampling_v2
from .kvcache_model import KVCacheModel
from .utils import (norm_logits, sample, max_fn, top_k_top_p_filter, get_seq_att_mask, get_accept_prob, update_large_prob,
                    get_num_acc_prob, get_expect_cnt_by_thres)


def _out_of_scope(name: str, where: str, why: str):
    def stub(*args, **kwargs):
        raise NotImplementedError(f"{name} (reference sampling/{where}) is out of scope of this build: {why}")
    stub.__name__ = name
    stub.__doc__ = f"Out of scope (SURVEY.md section 8): {why}"
    return stub


speculative_sampling_v2 = _out_of_scope("speculative_sampling_v2", "speculative_sampling.py:2079-2194",
                                        "the no-KV-cache variant; use speculative_sampling")
beam_speculative_sampling = _out_of_scope("beam_speculative_sampling", "speculative_sampling.py:585-1115",
                                          "rests on beam_sample_with_kv_cache")
mjsd_speculative_sampling = _out_of_scope("mjsd_speculative_sampling", "speculative_sampling.py:1117-1376",
                                          "joint-probability multi-draft variant")
BiLD_sampling = _out_of_scope("BiLD_sampling", "speculative_sampling.py:1718-1872", "fallback / rollback policy variant")
random_width_beam_sampling = _out_of_scope("random_width_beam_sampling", "autoregressive_sampling.py:63-207",
                                           "target-only stochastic beam baseline")

__all__ = ["speculative_sampling", "speculative_sampling_v2", "autoregressive_sampling", "multi_speculative_sampling",
           "beam_speculative_sampling", "BiLD_sampling", "mjsd_speculative_sampling", "random_width_beam_sampling",
           "beam_speculative_sampling_v2",
           "speculative_sampling_batch", "KVCacheModel", "norm_logits", "sample", "max_fn", "top_k_top_p_filter"]
