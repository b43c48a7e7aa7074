"""Device-resident decoder models and sessions on top of libspecdec.so.

``SpecDecModel`` holds one model's weights in HBM in the layout the HIP kernels
stream (bf16 GEMM matrices tile-packed) plus the ``sd_model`` handle.
``Session`` is one sequence's KV arena + scratch: the state a reference
``KVCacheModel`` keeps in ``_past_key_values`` (kvcache_model.py:24-36).

PyTorch is used here for device memory, streams and weight loading only; every
kernel on the path is in csrc/.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Dict, List, Optional

import torch

from . import _lib
from ._lib import lib, check
from .config import ModelConfig, config_from_hf

MAX_ROWS_PER_FORWARD = 80       # rows of one stream-batched pass (SD_MAX_ROWS: 8 streams x (gamma + 1 = 9) verify rows = 72)
MAX_LOGIT_ROWS = 64             # logit rows of one single-sequence call (the lm_head's row gather: streaming kernel), tree nodes
MAX_PREFILL_ROWS = 256          # rows of one single-sequence sd_session_forward call; longer prompts are chunked


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def check_token_ids(ids, vocab: int) -> None:
    """nn.Embedding raises IndexError on an id outside [0, vocab) (reference modeling_llama.py:672, modeling_opt.py:
    669); the HIP embedding gather has no such check, so every Python entry validates the ids it is handed
    (a tokenizer's added pad / special token beyond the table is the usual way to get one)."""
    if isinstance(ids, torch.Tensor):
        if ids.numel() == 0:
            return
        lo, hi = int(ids.min()), int(ids.max())
    else:
        if not ids:
            return
        lo, hi = min(ids), max(ids)
    if lo < 0 or hi >= vocab:
        raise IndexError(f"index out of range in self: token id {lo if lo < 0 else hi} outside [0, {vocab})")


def same_device(draft: "SpecDecModel", target: "SpecDecModel") -> None:
    """The reference tolerates a draft and a target on different devices (x.to(device), speculative_sampling.py:
    1949); here both models' kernels run on one stream, so raw pointers must belong to one GPU."""
    if draft.device != target.device and (draft.device.index or 0) != (target.device.index or 0):
        raise NotImplementedError(f"draft on {draft.device} and target on {target.device}: cross-device speculative "
                                  "decoding is out of scope (place both models on one GPU)")


class SpecDecModel:
    """Weights in HBM + sd_model handle.  Exposes what the reference reads off a model object:
    ``.config.is_encoder_decoder`` (speculative_sampling.py:1942) and ``.device`` (:1909)."""

    def __init__(self, cfg: ModelConfig, get_tensor: Callable[[str], torch.Tensor],
                 dtype: torch.dtype = torch.bfloat16, device: str = "cuda", max_pos: Optional[int] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("SpecDecModel needs a GPU: the HIP path has no CPU fallback")
        assert dtype in (torch.float32, torch.bfloat16, torch.float16)
        self.config = cfg
        self.cfg = cfg
        self.dtype = dtype
        self.device = torch.device(device)
        self.max_pos = int(max_pos or cfg.max_position_embeddings)
        self.kv_dtype = None                         # None: KV arenas in the model dtype; "fp8": OCP e4m3 (config 5)
        self.tp_group = None                         # tp.TPGroup when this model is one shard of a tensor-parallel target
        self._keep: List[torch.Tensor] = []          # owns every device tensor the handle points into
        self._arrays = []
        self.weight_bytes = 0                        # bytes the forward streams (embedding tables excluded)
        self.fused = dtype != torch.float32 and cfg.intermediate_size % 8 == 0 and cfg.head_dim % 4 == 0
        # dtype of the probability rows the reference would keep for this model (kvcache_model.py:167-168 on the model's
        # logits): OPT's logits stay in the weight dtype (modeling_opt.py:974), Llama's are cast to fp32 (:870)
        self.probs_dtype = dtype if cfg.arch == "opt" else torch.float32
        self.norm_mode = {torch.bfloat16: _lib.SD_NORM_DT_BF16, torch.float16: _lib.SD_NORM_DT_F16}.get(self.probs_dtype, 0)
        self._build(get_tensor)

    # -- weight staging ---------------------------------------------------------------------
    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        t = t.to(device=self.device, dtype=self.dtype).contiguous()
        self._keep.append(t)
        return t

    def _gemm_weight(self, t: torch.Tensor) -> torch.Tensor:
        """[N][K] matrix -> device layout for the GEMM kernels."""
        t = t.to(device=self.device, dtype=self.dtype).contiguous()
        self.weight_bytes += t.numel() * t.element_size()
        if self.dtype != torch.float32:                       # bf16 / fp16: the tile layout is the same for both
            N, K = t.shape
            out = torch.empty_like(t)
            check(lib.sd_pack_weight_bf16(t.data_ptr(), out.data_ptr(), N, K, _stream()), "sd_pack_weight_bf16")
            torch.cuda.current_stream().synchronize()
            t = out
        self._keep.append(t)
        return t

    def _ptr_array(self, tensors: List[Optional[torch.Tensor]]):
        arr = (C.c_void_p * len(tensors))(*[_ptr(t) for t in tensors])
        self._arrays.append(arr)
        return C.cast(arr, C.POINTER(C.c_void_p))

    def _build(self, get: Callable[[str], torch.Tensor]):
        cfg = self.cfg
        L = cfg.num_hidden_layers
        w = _lib.SdModelWeights()
        wqkv, bqkv, wo, bo, wgu, bfc1, wdn, bfc2, n1w, n1b, n2w, n2b = ([] for _ in range(12))
        with torch.no_grad():
            if cfg.arch == "llama":
                w.embed = _ptr(self._dev(get("model.embed_tokens.weight")))
                for i in range(L):
                    p = f"model.layers.{i}."
                    q, k, v = (get(p + f"self_attn.{n}_proj.weight").to(self.device) for n in "qkv")
                    if self.fused:
                        # pair-interleave the rows of every q / k head: d0, d0+D/2, d1, d1+D/2, ... so that one
                        # accumulator quad of the GEMM holds two complete RoPE pairs (specdec.h, fused_layout)
                        D, K = cfg.head_dim, q.shape[1]
                        q = q.view(-1, 2, D // 2, K).transpose(1, 2).reshape(-1, K)
                        k = k.view(-1, 2, D // 2, K).transpose(1, 2).reshape(-1, K)
                    wqkv.append(self._gemm_weight(torch.cat([q, k, v], 0)))
                    del q, k, v
                    wo.append(self._gemm_weight(get(p + "self_attn.o_proj.weight")))
                    g, u = get(p + "mlp.gate_proj.weight").to(self.device), get(p + "mlp.up_proj.weight").to(self.device)
                    if self.fused:
                        K = g.shape[1]                        # 8 gate rows, the same 8 up rows, ...
                        wgu.append(self._gemm_weight(torch.stack([g.view(-1, 8, K), u.view(-1, 8, K)], 1).reshape(-1, K)))
                    else:
                        wgu.append(self._gemm_weight(torch.cat([g, u], 0)))
                    del g, u
                    wdn.append(self._gemm_weight(get(p + "mlp.down_proj.weight")))
                    n1w.append(self._dev(get(p + "input_layernorm.weight")))
                    n2w.append(self._dev(get(p + "post_attention_layernorm.weight")))
                    for lst in (bqkv, bo, bfc1, bfc2, n1b, n2b):
                        lst.append(None)
                w.final_norm_w = _ptr(self._dev(get("model.norm.weight")))
                w.lm_head = _ptr(self._gemm_weight(get("lm_head.weight")))
                # rope table exactly as the reference builds it (modeling_llama.py:107-125)
                D = cfg.head_dim
                inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2).float() / D))
                ang = torch.outer(torch.arange(self.max_pos, dtype=torch.float32), inv)
                w.rope_cos = _ptr(self._dev(ang.cos()))
                w.rope_sin = _ptr(self._dev(ang.sin()))
                self.weight_bytes += (2 * L + 1) * cfg.hidden_size * (4 if self.dtype == torch.float32 else 2)
            else:
                d = "model.decoder."
                emb = get(d + "embed_tokens.weight")
                w.embed = _ptr(self._dev(emb))
                w.lm_head = _ptr(self._gemm_weight(emb))          # tied head (modeling_opt.py:833,840)
                del emb
                w.pos_embed = _ptr(self._dev(get(d + "embed_positions.weight")))
                if cfg.word_embed_proj_dim != cfg.hidden_size:
                    w.project_in = _ptr(self._gemm_weight(get(d + "project_in.weight")))
                    w.project_out = _ptr(self._gemm_weight(get(d + "project_out.weight")))
                if cfg.do_layer_norm_before:
                    w.final_norm_w = _ptr(self._dev(get(d + "final_layer_norm.weight")))
                    w.final_norm_b = _ptr(self._dev(get(d + "final_layer_norm.bias")))
                for i in range(L):
                    p = d + f"layers.{i}."
                    q, k, v = (get(p + f"self_attn.{n}_proj.weight") for n in "qkv")
                    wqkv.append(self._gemm_weight(torch.cat([q.to(self.device), k.to(self.device), v.to(self.device)], 0)))
                    bqkv.append(self._dev(torch.cat([get(p + f"self_attn.{n}_proj.bias").to(self.device) for n in "qkv"], 0)))
                    wo.append(self._gemm_weight(get(p + "self_attn.out_proj.weight")))
                    bo.append(self._dev(get(p + "self_attn.out_proj.bias")))
                    wgu.append(self._gemm_weight(get(p + "fc1.weight")))
                    bfc1.append(self._dev(get(p + "fc1.bias")))
                    wdn.append(self._gemm_weight(get(p + "fc2.weight")))
                    bfc2.append(self._dev(get(p + "fc2.bias")))
                    n1w.append(self._dev(get(p + "self_attn_layer_norm.weight")))
                    n1b.append(self._dev(get(p + "self_attn_layer_norm.bias")))
                    n2w.append(self._dev(get(p + "final_layer_norm.weight")))
                    n2b.append(self._dev(get(p + "final_layer_norm.bias")))
        w.wqkv, w.bqkv = self._ptr_array(wqkv), self._ptr_array(bqkv)
        w.wo, w.bo = self._ptr_array(wo), self._ptr_array(bo)
        w.w_gate_up, w.b_fc1 = self._ptr_array(wgu), self._ptr_array(bfc1)
        w.w_down, w.b_fc2 = self._ptr_array(wdn), self._ptr_array(bfc2)
        w.norm1_w, w.norm1_b = self._ptr_array(n1w), self._ptr_array(n1b)
        w.norm2_w, w.norm2_b = self._ptr_array(n2w), self._ptr_array(n2b)

        c = _lib.SdModelConfig(
            arch=cfg.arch_id, dtype={torch.bfloat16: _lib.SD_BF16, torch.float16: _lib.SD_F16}.get(self.dtype, _lib.SD_F32),
            vocab=cfg.vocab_size, hidden=cfg.hidden_size, inter=cfg.intermediate_size, n_layers=L,
            n_heads=cfg.num_attention_heads, n_kv_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
            max_pos=self.max_pos, opt_pre_ln=int(cfg.do_layer_norm_before), opt_proj_dim=cfg.word_embed_proj_dim or cfg.hidden_size,
            norm_eps=cfg.rms_norm_eps if cfg.arch == "llama" else cfg.layer_norm_eps,
            logits_bf16_round=int(self.dtype != torch.float32), fused_layout=int(self.fused))
        h = C.c_void_p()
        check(lib.sd_model_create(C.byref(c), C.byref(w), C.byref(h)), "sd_model_create")
        self.handle = h
        self._w, self._c = w, c

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and lib is not None:              # (module globals are gone at interpreter shutdown)
            lib.sd_model_destroy(h)
            self.handle = None

    # -- constructors -----------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, cfg: ModelConfig, sd: Dict[str, torch.Tensor], dtype=torch.bfloat16,
                        device="cuda", max_pos=None) -> "SpecDecModel":
        return cls(cfg, lambda n: sd[n], dtype=dtype, device=device, max_pos=max_pos)

    @classmethod
    def from_hf(cls, module, dtype: Optional[torch.dtype] = None, device="cuda", max_pos=None) -> "SpecDecModel":
        """Extract weights from a transformers LlamaForCausalLM / OPTForCausalLM (reference
        evaluation.py:183-253 hands such modules to speculative_sampling)."""
        cfg = config_from_hf(module.config)
        sd = module.state_dict()
        if dtype is None:                                 # the module's own dtype (the reference harness loads fp16)
            dtype = next(iter(sd.values())).dtype
            if dtype not in (torch.float32, torch.bfloat16, torch.float16):
                dtype = torch.float32
        return cls(cfg, lambda n: sd[n], dtype=dtype, device=device, max_pos=max_pos)

    @classmethod
    def synthetic(cls, cfg: ModelConfig, seed: int, dtype=torch.bfloat16, device="cuda", max_pos=None,
                  method: str = "torch", gain: float = 1.0, head_gain: float = 4.0, transform=None,
                  seed_of=None) -> "SpecDecModel":
        """Random-init weights generated tensor by tensor (never the whole model at once on the host).
        ``transform(name, tensor) -> tensor`` post-processes each generated tensor (synth.acceptance_dial_pair)."""
        from .synth import param_shapes, _scale
        import numpy as np
        shapes = {n: (i, s, k) for i, (n, s, k) in enumerate(param_shapes(cfg))}
        gen = torch.Generator(device=device)

        def raw(name: str) -> torch.Tensor:
            idx, shape, kind = shapes[name]
            mean, std = _scale(kind, shape, gain, head_gain)
            if method == "numpy":
                rng = np.random.default_rng([int(seed), idx])
                a = rng.standard_normal(size=shape, dtype=np.float32) * np.float32(std) + np.float32(mean)
                return torch.from_numpy(a)
            gen.manual_seed((int(seed_of(name)) if seed_of is not None else int(seed)) * 100003 + idx)
            t = torch.empty(shape, dtype=dtype, device=device)
            return t.normal_(mean, std, generator=gen)

        def get(name: str) -> torch.Tensor:
            t = raw(name)
            return transform(name, t) if transform is not None else t
        m = cls(cfg, get, dtype=dtype, device=device, max_pos=max_pos)
        m._synth_get, m._synth_names = get, list(shapes)     # lets a host baseline regenerate the same tensors
        return m

    @classmethod
    def from_pretrained_dir(cls, path: str, dtype=torch.bfloat16, device="cuda", max_pos=None) -> "SpecDecModel":
        """A LOCAL HF checkpoint directory (config.json + *.safetensors or pytorch_model*.bin), read tensor by tensor with
        loaders that execute nothing from the files (safetensors; torch.load(weights_only=True)) - never the hub
        (the reference loads by hub name, evaluation.py:183-253; SURVEY.md 8(d): SPECDEC_MODEL_DIR)."""
        cfg, get, names = checkpoint_getter(path)
        m = cls(cfg, get, dtype=dtype, device=device, max_pos=max_pos)
        m._synth_get, m._synth_names = get, names           # lets a host baseline read the same tensors
        m.checkpoint = path
        return m

    def new_session(self, max_seq: int, max_rows: int = MAX_PREFILL_ROWS, kv_dtype: Optional[str] = None) -> "Session":
        ses = Session(self, max_seq, max_rows, kv_dtype=kv_dtype or self.kv_dtype)
        if self.tp_group is not None:
            self.tp_group.bind(ses)
        return ses


def checkpoint_getter(path: str):
    """(ModelConfig, get(name) -> CPU tensor, names) for a local HF checkpoint directory.  Tied OPT heads are resolved the
    way the reference's classes tie them (modeling_opt.py:833,840)."""
    import glob
    import json
    import os
    from types import SimpleNamespace
    from .synth import param_shapes
    with open(os.path.join(path, "config.json")) as f:
        raw = json.load(f)
    hf = SimpleNamespace(**raw)
    if not hasattr(hf, "_name_or_path"):
        hf._name_or_path = os.path.basename(os.path.normpath(path))
    if getattr(hf, "model_type", "") == "opt":
        for k, v in (("do_layer_norm_before", True), ("word_embed_proj_dim", raw.get("hidden_size"))):
            if not hasattr(hf, k):
                setattr(hf, k, v)
    cfg = config_from_hf(hf)
    where: Dict[str, tuple] = {}
    st_files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if st_files:
        from safetensors import safe_open
        for fn in st_files:
            with safe_open(fn, framework="pt", device="cpu") as f:
                for k in f.keys():
                    where[k] = ("st", fn)
    else:
        for fn in sorted(glob.glob(os.path.join(path, "pytorch_model*.bin"))):
            sd = torch.load(fn, map_location="cpu", weights_only=True, mmap=True)
            for k in sd:
                where[k] = ("pt", fn)
            del sd
    if not where:
        raise FileNotFoundError(f"{path}: no *.safetensors or pytorch_model*.bin")
    cache: Dict[str, dict] = {}

    def get(name: str) -> torch.Tensor:
        key = name
        if key not in where:
            alts = [name.replace("model.decoder.", "decoder."), "model." + name]
            if name == "lm_head.weight":
                alts += ["model.decoder.embed_tokens.weight", "decoder.embed_tokens.weight", "model.embed_tokens.weight"]
            key = next((a for a in alts if a in where), None)
            if key is None:
                raise KeyError(f"{path}: tensor {name!r} not in the checkpoint")
        kind, fn = where[key]
        if kind == "st":
            from safetensors import safe_open
            with safe_open(fn, framework="pt", device="cpu") as f:
                return f.get_tensor(key)
        if fn not in cache:
            cache.clear()
            cache[fn] = torch.load(fn, map_location="cpu", weights_only=True, mmap=True)
        return cache[fn][key]
    names = [n for n, _, _ in param_shapes(cfg)]
    if cfg.arch == "opt":
        names.append("lm_head.weight")
    return cfg, get, names


class Session:
    """KV arena [L][2][H_kv][max_seq][D] + scratch for one sequence; ``cache_len`` is the number of
    positions held, so rollback is an assignment (reference kvcache_model.py:359-436)."""

    def __init__(self, model: SpecDecModel, max_seq: int, max_rows: int = MAX_PREFILL_ROWS, kv_dtype: Optional[str] = None):
        cfg = model.cfg
        self.model = model
        self.max_seq = int(min(max_seq, model.max_pos))
        self.max_rows = int(min(max_rows, MAX_PREFILL_ROWS, lib.sd_model_max_rows(model.handle)))
        dev = model.device
        self.kv_fp8 = kv_dtype == "fp8"
        assert kv_dtype in (None, "fp8"), kv_dtype
        shape = (cfg.num_hidden_layers, 2, cfg.num_key_value_heads, self.max_seq, cfg.head_dim)
        if self.kv_fp8:                              # 1 byte per element; x is stored as fp8(x / scale[layer, k|v, head])
            self.kv = torch.zeros(shape, dtype=torch.uint8, device=dev)
            self.kv_scale = torch.ones((cfg.num_hidden_layers, 2, cfg.num_key_value_heads), dtype=torch.float32, device=dev)
        else:
            self.kv = torch.zeros(shape, dtype=model.dtype, device=dev)
            assert self.kv.numel() * self.kv.element_size() == lib.sd_session_kv_bytes(model.handle, self.max_seq)
        nbytes = lib.sd_session_scratch_bytes(model.handle, self.max_rows)
        self.scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.logits = torch.empty((min(self.max_rows, MAX_ROWS_PER_FORWARD), cfg.vocab_size), dtype=torch.float32,
                                  device=dev)
        h = C.c_void_p()
        check(lib.sd_session_create(model.handle, self.max_seq, self.max_rows, self.kv.data_ptr(),
                                    self.scratch.data_ptr(), C.byref(h)), "sd_session_create")
        self.handle = h
        self.cache_len = 0
        if self.kv_fp8:
            check(lib.sd_session_set_kv_fp8(h, self.kv_scale.data_ptr()), "sd_session_set_kv_fp8")

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and lib is not None:
            lib.sd_session_destroy(h)
            self.handle = None

    def forward(self, tokens: torch.Tensor, n_logits: int, logits_out: Optional[torch.Tensor] = None,
                pos0: Optional[int] = None) -> torch.Tensor:
        """Feed ``tokens`` (device int32, 1-D) at positions pos0.. (default: append at cache_len); returns
        fp32 logits of the last ``n_logits`` fed rows (a view of ``logits_out`` or of the session buffer).
        Prompts longer than one launch chain's row budget are chunked."""
        assert tokens.dtype == torch.int32 and tokens.is_cuda and tokens.dim() == 1
        n = tokens.numel()
        if pos0 is None:
            pos0 = self.cache_len
        if logits_out is None:
            logits_out = self.logits
        assert n_logits <= logits_out.shape[0] and logits_out.stride(1) == 1
        ld = logits_out.stride(0)
        st = _stream()
        first_logit_row = n - n_logits
        done = 0
        while done < n:
            m = min(self.max_rows, n - done)
            lo = max(first_logit_row, done)            # rows of this chunk that need logits (at most 64 per call)
            if done + m - lo > MAX_LOGIT_ROWS:
                m = lo + MAX_LOGIT_ROWS - done
            nl = max(0, done + m - lo)
            dst = logits_out.data_ptr() + (lo - first_logit_row) * ld * 4 if nl else None
            check(lib.sd_session_forward(self.handle, tokens.data_ptr() + done * 4, m, pos0 + done, nl, dst, ld, st),
                  "sd_session_forward")
            done += m
        self.cache_len = pos0 + n
        return logits_out[:n_logits]

    def rollback(self, end_pos: int) -> None:
        self.cache_len = min(self.cache_len, int(end_pos))

    def past_key_values(self):
        """The reference's tuple layout: one (k, v) pair of (1, H_kv, S, D) views per layer."""
        S = self.cache_len
        if self.kv_fp8:                              # widened copies (the arena itself stays fp8)
            kv = self.kv.view(torch.float8_e4m3fn)[:, :, :, :S, :].to(self.model.dtype) * \
                self.kv_scale[:, :, :, None, None].to(self.model.dtype)
            return [(kv[l, 0].unsqueeze(0), kv[l, 1].unsqueeze(0)) for l in range(kv.shape[0])]
        return [(self.kv[l, 0, :, :S, :].unsqueeze(0), self.kv[l, 1, :, :S, :].unsqueeze(0))
                for l in range(self.kv.shape[0])]

    # -- per-op-class timing (roofline report) ----------------------------------------------
    def profile(self, on: bool) -> None:
        check(lib.sd_profile_enable(self.handle, int(on)), "sd_profile_enable")

    def profile_read(self):
        ms = (C.c_float * _lib.N_PROFILE_CLASSES)()
        cnt = (C.c_int * _lib.N_PROFILE_CLASSES)()
        check(lib.sd_profile_read(self.handle, ms, cnt), "sd_profile_read")
        return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(_lib.PROFILE_CLASS_NAMES) if cnt[i]}


import weakref

_MODEL_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()   # HF module -> SpecDecModel (dies with the module)


def as_specdec_model(model, dtype: Optional[torch.dtype] = None) -> SpecDecModel:
    """Accept the engine's own model or an HF module (converted once, cached per module object)."""
    if isinstance(model, SpecDecModel):
        return model
    if not hasattr(model, "state_dict") or not hasattr(model, "config"):
        raise TypeError(f"cannot use {type(model).__name__} as a decoder model")
    if getattr(model.config, "is_encoder_decoder", False):
        raise NotImplementedError("encoder-decoder models (reference speculative_sampling.py:1946,1958) are out of scope")
    hit = _MODEL_CACHE.get(model)
    if hit is None:
        hit = SpecDecModel.from_hf(model, dtype=dtype)
        _MODEL_CACHE[model] = hit
    return hit


def batch_forward(sessions: List[Session], seqs: List[torch.Tensor], n_new: List[int], n_logits: List[int],
                  logits_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Stream-batched forward (sd_batch_forward): stream i feeds seqs[i][cache_len : cache_len + n_new[i]] (seqs[i] is
    that stream's whole int32 token buffer on the device, indexed by absolute position).  Returns the packed fp32 logits
    of the last n_logits[i] rows of every stream, in stream order.  All sessions share one model."""
    B = len(sessions)
    items = (_lib.SdBatchItem * B)()
    for i, (ses, sq) in enumerate(zip(sessions, seqs)):
        assert sq.dtype == torch.int32 and sq.is_cuda
        items[i].session = ses.handle
        items[i].seq = sq.data_ptr()
        items[i].pos0 = ses.cache_len
        items[i].n_new = int(n_new[i])
        items[i].n_logits = int(n_logits[i])
    tot = int(sum(n_logits))
    if logits_out is None:
        logits_out = sessions[0].logits
    assert tot <= logits_out.shape[0]
    check(lib.sd_batch_forward(items, B, logits_out.data_ptr() if tot else None, logits_out.stride(0), _stream()),
          "sd_batch_forward")
    for ses, n in zip(sessions, n_new):
        ses.cache_len += int(n)
    return logits_out[:tot]


def batch_prefill(sessions: List[Session], seqs: List[torch.Tensor], n_new: List[int]) -> None:
    """Batched prefill (sd_batch_prefill): stream i feeds seqs[i][cache_len : cache_len + n_new[i]] with no logits.  The
    streams are packed greedily into passes of at most MAX_PREFILL_ROWS rows / 32 attention groups / 16 streams, so B
    prompts cost ceil(rows / 256) passes over the weights instead of B; a stream whose run does not fit one pass is fed
    by its own chunked Session.forward.  All sessions share one model."""
    st = _stream()
    chunk: List[int] = []
    rows = groups = 0

    def flush():
        nonlocal chunk, rows, groups
        if not chunk:
            return
        items = (_lib.SdBatchItem * len(chunk))()
        for j, i in enumerate(chunk):
            items[j].session = sessions[i].handle
            items[j].seq = seqs[i].data_ptr()
            items[j].pos0 = sessions[i].cache_len
            items[j].n_new = int(n_new[i])
            items[j].n_logits = 0
        check(lib.sd_batch_prefill(items, len(chunk), st), "sd_batch_prefill")
        for i in chunk:
            sessions[i].cache_len += int(n_new[i])
        chunk, rows, groups = [], 0, 0

    cap = min([MAX_PREFILL_ROWS] + [s.max_rows for s in sessions])
    for i, (ses, n) in enumerate(zip(sessions, n_new)):
        n = int(n)
        if n <= 0:
            continue
        g = (n + 7) // 8
        if n > cap or g > 32:
            flush()
            ses.forward(seqs[i][ses.cache_len:ses.cache_len + n], 0)
            continue
        if rows + n > cap or groups + g > 32 or len(chunk) >= 16:
            flush()
        chunk.append(i)
        rows += n
        groups += g
    flush()
