"""CPU oracle: a torch-CPU restatement of the reference's speculative-sampling path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``llmspeculativesampling_amd`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may.  It is the checker, never the thing measured or shipped.

Parity status: the reference (ZongyueQin/LLMSpeculativeSampling @ 2025-02-04)
ships no tests and no golden vectors, so upstream pins
nothing.  The oracle is instead pinned by fixtures generated in the build
container by importing the reference itself (``tests/golden/make_golden.py``,
torch 2.10.0 CPU) and committed under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function here against them.

Each function cites the reference file:line it restates.  Randomness is made
explicit through ``oracle.noise``: the reference's draws (``torch.multinomial`` ==
argmax(p / Exp(1)-noise), ``torch.rand(1)``, ``torch.manual_seed``) are routed
through a NoiseSource so the same stream can be replayed into the HIP kernels.
"""
from .noise import TorchGlobalNoise, RecordedNoise, RecordingNoise  # noqa: F401
from .sampling_ref import norm_logits, top_k_top_p_filter, sample, max_fn  # noqa: F401
from .models_ref import RefCausalLM  # noqa: F401
from .kvcache_ref import RefKVCacheModel  # noqa: F401
from .specdec_ref import speculative_sampling, autoregressive_sampling  # noqa: F401
from .multi_ref import multi_speculative_sampling  # noqa: F401
from . import tree_ref  # noqa: F401
