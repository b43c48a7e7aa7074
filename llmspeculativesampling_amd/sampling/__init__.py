"""Drop-in for the hot path of the reference's ``sampling`` package (sampling/__init__.py:1-7)."""
from .speculative_sampling import speculative_sampling
from .autoregressive_sampling import autoregressive_sampling
from .batch import speculative_sampling_batch
from .multi import multi_speculative_sampling
from .kvcache_model import KVCacheModel
from .utils import norm_logits, sample, max_fn, top_k_top_p_filter

__all__ = ["speculative_sampling", "autoregressive_sampling", "speculative_sampling_batch", "multi_speculative_sampling",
           "KVCacheModel",
           "norm_logits", "sample", "max_fn", "top_k_top_p_filter"]
