"""MI355X-native speculative-sampling decode engine (drop-in for the hot path of
ZongyueQin/LLMSpeculativeSampling).  Importing the sampling API requires the built
libspecdec.so; ``config`` and ``synth`` are importable without it."""
__version__ = "0.1.0"
