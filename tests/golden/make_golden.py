#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference); never on the GPU box.
Recipe for importing the reference under transformers 5.x: SURVEY.md section 8(c).
The fixtures are data only: seeds, inputs, recorded noise and the reference's outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Every ``torch.multinomial`` the reference issues is intercepted: the Exp(1) variates
it is about to consume are drawn first from a copy of the generator state, the real
call then runs, and the result is asserted equal to argmax(p / noise).  That both
records the noise and proves the multinomial == argmax(p/Exp(1)) identity on every
call in every fixture (torch 2.10.0 CPU).
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import torch  # noqa: E402
import transformers  # noqa: E402
import transformers.models.bloom.modeling_bloom  # noqa: E402,F401
import transformers.generation  # noqa: E402,F401


class _Stub:
    pass


for _n in ("BeamSearchScorer", "BeamScorer"):
    setattr(sys.modules["transformers"], _n, _Stub)
for _n in ("BeamSampleDecoderOnlyOutput", "BeamSampleEncoderDecoderOutput"):
    setattr(sys.modules["transformers.generation"], _n, _Stub)

import sampling as ref_sampling  # noqa: E402
import sampling.utils as ref_utils  # noqa: E402
import sampling.kvcache_model as ref_kv  # noqa: E402
import sampling.speculative_sampling  # noqa: E402,F401
import sampling.autoregressive_sampling  # noqa: E402,F401
# sampling/__init__.py rebinds these names to the functions; take the modules from sys.modules
ref_ss = sys.modules["sampling.speculative_sampling"]
ref_ar = sys.modules["sampling.autoregressive_sampling"]
import sampling.models.modeling_llama as ML  # noqa: E402
import sampling.models.modeling_opt as MO  # noqa: E402

ML.LlamaForCausalLM._tied_weights_keys = {}
MO.OPTForCausalLM._tied_weights_keys = {"lm_head.weight": "model.decoder.embed_tokens.weight"}

from llmspeculativesampling_amd.config import load_config  # noqa: E402
from llmspeculativesampling_amd.synth import make_state_dict, perturb_state_dict  # noqa: E402
import oracle  # noqa: E402

torch.set_num_threads(4)

# --------------------------------------------------------------------------- noise capture
_real_multinomial = torch.multinomial
_real_rand = torch.rand
_real_seed = torch.manual_seed
EVENTS = []


def _multinomial(p, num_samples=1, replacement=False, **kw):
    assert num_samples == 1 and p.dim() == 2            # (1, V) rows; (width, V) in multi_speculative_sampling
    st = torch.get_rng_state()
    e = torch.empty_like(p).exponential_(1)
    torch.set_rng_state(st)
    out = _real_multinomial(p, num_samples=num_samples, replacement=replacement, **kw)
    assert torch.equal(out, torch.argmax(p / e, dim=-1, keepdim=True)), "multinomial != argmax(p/Exp)"
    EVENTS.append(("exp", e.clone()))
    return out


def _rand(*a, **kw):
    r = _real_rand(*a, **kw)
    if tuple(r.shape) == (1,):
        EVENTS.append(("uni", r.clone()))
    return r


def _seed(s):
    EVENTS.append(("seed", int(s)))
    return _real_seed(s)


def capture_on():
    EVENTS.clear()
    torch.multinomial = _multinomial
    torch.rand = _rand
    torch.manual_seed = _seed


def capture_off():
    torch.multinomial = _real_multinomial
    torch.rand = _real_rand
    torch.manual_seed = _real_seed
    ev = list(EVENTS)
    EVENTS.clear()
    return ev


def pack_events(ev):
    kinds = np.array([{"exp": 0, "uni": 1, "seed": 2}[k] for k, _ in ev], dtype=np.uint8)
    exps = [v.float().numpy().reshape(-1) for k, v in ev if k == "exp"]
    unis = [float(v) for k, v in ev if k == "uni"]
    seeds = [v for k, v in ev if k == "seed"]
    return dict(kinds=kinds,
                exp=np.stack(exps).astype(np.float32) if exps else np.zeros((0, 0), np.float32),
                uni=np.array(unis, dtype=np.float32), seed=np.array(seeds, dtype=np.int64))


def pack_events_ragged(ev):
    """As pack_events, for streams whose Exp(1) draws differ in size ((width, V) and (1, V))."""
    kinds = np.array([{"exp": 0, "uni": 1, "seed": 2}[k] for k, _ in ev], dtype=np.uint8)
    exps = [v.float().numpy().reshape(-1) for k, v in ev if k == "exp"]
    return dict(kinds=kinds,
                expflat=np.concatenate(exps).astype(np.float32) if exps else np.zeros(0, np.float32),
                exprows=np.array([v.shape[0] for k, v in ev if k == "exp"], dtype=np.int32),
                expsize=np.array([e.size for e in exps], dtype=np.int64),
                uni=np.array([float(v) for k, v in ev if k == "uni"], dtype=np.float32),
                seed=np.array([v for k, v in ev if k == "seed"], dtype=np.int64))


# --------------------------------------------------------------------------- reference models
def ref_model(cfg, sd):
    if cfg.arch == "llama":
        hc = transformers.LlamaConfig(
            vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
            intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
            num_key_value_heads=cfg.num_key_value_heads, max_position_embeddings=cfg.max_position_embeddings,
            rms_norm_eps=cfg.rms_norm_eps)
        hc.rope_theta, hc.rope_scaling, hc.pretraining_tp = cfg.rope_theta, None, 1
        m = ML.LlamaForCausalLM(hc)
    else:
        hc = transformers.OPTConfig(
            vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
            ffn_dim=cfg.ffn_dim, num_attention_heads=cfg.num_attention_heads,
            max_position_embeddings=cfg.max_position_embeddings,
            do_layer_norm_before=cfg.do_layer_norm_before, word_embed_proj_dim=cfg.word_embed_proj_dim,
            dropout=0.0, attention_dropout=0.0, layerdrop=0.0)
        for k, v in (("_remove_final_layer_norm", False), ("enable_bias", True),
                     ("layer_norm_elementwise_affine", True)):
            if not hasattr(hc, k):
                setattr(hc, k, v)
        m = MO.OPTForCausalLM(hc)
    dtype = next(iter(sd.values())).dtype
    m = m.to(dtype)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    bad = [k for k in missing if "rotary_emb" not in k and "lm_head" not in k]
    assert not bad and not unexpected, (bad, unexpected)
    if cfg.arch == "opt":
        m.lm_head.weight = m.model.decoder.embed_tokens.weight
    return m.eval()


# --------------------------------------------------------------------------- fixtures
def logits_row(seed, V, scale=4.0, dtype=torch.float32):
    rng = np.random.default_rng([seed, V])
    return torch.from_numpy((rng.standard_normal(V, dtype=np.float32) * np.float32(scale))[None]).to(dtype)


def g1_norm_logits():
    """reference utils.norm_logits on seeded rows; inputs are regenerated from the seed by the tests."""
    cases, blobs = [], {}
    grid = [(1.0, 0, 0.0), (1.0, 20, 0.9), (0.7, 50, 0.95), (1.0, 0, 0.8), (1.3, 5, 0.0), (1.0, 1, 0.0)]
    cid = 0
    for V in (257, 32000, 50272):
        for (T, k, p) in grid:
            for dtype in (torch.float32, torch.bfloat16, torch.float16):
                if dtype != torch.float32 and V != 257:
                    continue
                seed = 100 + cid
                x = logits_row(seed, V, dtype=dtype)
                probs = ref_utils.norm_logits(x.clone(), T, k, p)
                assert probs.dtype == dtype
                mine = oracle.norm_logits(x.clone(), T, k, p)
                assert torch.equal(mine, probs), ("oracle != reference", V, T, k, p, dtype)
                pf = probs.float().numpy()[0]
                nz = np.nonzero(pf)[0]
                key = f"c{cid}"
                if k == 0 and p == 0.0 and V > 257:
                    blobs[key + "_dense"] = pf
                    cases.append(dict(id=key, kind="seeded", seed=seed, V=V, scale=4.0, T=T, k=k, p=p,
                                      dtype=str(dtype).split(".")[1], dense=True))
                else:
                    blobs[key + "_idx"] = nz.astype(np.int32)
                    blobs[key + "_val"] = pf[nz]
                    cases.append(dict(id=key, kind="seeded", seed=seed, V=V, scale=4.0, T=T, k=k, p=p,
                                      dtype=str(dtype).split(".")[1], dense=False))
                cid += 1
    # engineered rows (stored inline): ties at the k-th value; ties across the top-p cut; all-equal row
    eng = [
        ("ties_topk", [3.0, 1.0, 2.0, 2.0, 0.5, 2.0, 2.0, -1.0], 1.0, 2, 0.0),
        ("ties_topk_p", [3.0, 1.0, 2.0, 2.0, 0.5, 2.0, 2.0, -1.0], 1.0, 3, 0.7),
        ("all_equal_p50", [0.0, 0.0, 0.0, 0.0], 1.0, 0, 0.5),
        ("all_equal_p75", [0.0] * 8, 1.0, 0, 0.75),
        ("tie_across_cut", [2.0, 1.0, 1.0, 1.0, 1.0, -3.0, 1.0, 0.0], 1.0, 0, 0.6),
        ("single_dominant", [30.0, 0.0, 1.0, -2.0, 5.0], 1.0, 3, 0.9),
        ("temperature_half", [1.0, 2.0, 3.0, 4.0, 0.5, 0.25], 0.5, 4, 0.85),
        ("neg_inf_input", [1.0, float("-inf"), 0.5, 2.0, float("-inf"), 0.0], 1.0, 3, 0.9),
    ]
    for name, row, T, k, p in eng:
        x = torch.tensor([row], dtype=torch.float32)
        probs = ref_utils.norm_logits(x.clone(), T, k, p)
        assert torch.equal(oracle.norm_logits(x.clone(), T, k, p), probs), name
        cases.append(dict(id=name, kind="inline", row=[(v if np.isfinite(v) else "-inf") for v in row],
                          T=T, k=k, p=p, dtype="float32", expect=probs[0].tolist()))
    # error path: NaN logits raise RuntimeError('norm logits error') (utils.py:203-207)
    try:
        ref_utils.norm_logits(torch.tensor([[1.0, float("nan"), 0.0]]), 1.0, 0, 0.0)
        raised = False
    except RuntimeError as e:
        raised = str(e)
    cases.append(dict(id="nan_raises", kind="error", row=[1.0, "nan", 0.0], T=1.0, k=0, p=0.0, expect=raised))
    np.savez_compressed(os.path.join(HERE, "g1_norm_logits.npz"), **blobs)
    json.dump(cases, open(os.path.join(HERE, "g1_norm_logits.json"), "w"), indent=0)
    print("G1:", len(cases), "cases")


def g2_sample_maxfn():
    """reference utils.sample / max_fn with recorded noise."""
    out, blobs = [], {}
    cid = 0
    for V, (T, k, p) in [(257, (1.0, 0, 0.0)), (257, (1.0, 20, 0.9)), (2048, (1.0, 20, 0.9)),
                         (32000, (1.0, 20, 0.9)), (32000, (1.0, 0, 0.0)), (50272, (0.7, 50, 0.95))]:
        for rep in range(4):
            seed = 500 + cid
            probs = ref_utils.norm_logits(logits_row(seed, V), T, k, p)
            _real_seed(9000 + cid)
            capture_on()
            tok = ref_utils.sample(probs)
            ev = capture_off()
            assert len(ev) == 1 and ev[0][0] == "exp"
            blobs[f"s{cid}_noise"] = ev[0][1].numpy()[0]
            out.append(dict(id=f"s{cid}", seed=seed, V=V, T=T, k=k, p=p, token=int(tok)))
            cid += 1
    # fix-up row: the noise favours an index whose prob is < 1e-9 -> argmax(probs) instead (utils.py:228-230)
    V = 64
    probs = torch.zeros(1, V)
    probs[0, 5], probs[0, 9], probs[0, 20] = 0.7, 0.3 - 1e-10, 1e-10
    noise = torch.ones(1, V)
    noise[0, 20] = 1e-12

    def fake(p, num_samples=1, replacement=False):
        return torch.argmax(p / noise, dim=-1, keepdim=True)
    torch.multinomial = fake
    tok = ref_utils.sample(probs)
    torch.multinomial = _real_multinomial
    assert int(tok) == 5
    out.append(dict(id="fixup", inline_probs=probs[0].tolist(), inline_noise=noise[0].tolist(), token=int(tok)))
    # all-but-one-zero row
    probs = torch.zeros(1, V)
    probs[0, 33] = 1.0
    _real_seed(1)
    capture_on()
    tok = ref_utils.sample(probs)
    ev = capture_off()
    out.append(dict(id="onehot", inline_probs=probs[0].tolist(), inline_noise=ev[0][1][0].tolist(), token=int(tok)))
    # all-zero row raises 'prob error' without touching the generator
    _real_seed(2)
    st = torch.get_rng_state()
    try:
        ref_utils.sample(torch.zeros(1, V))
        raised = False
    except RuntimeError as e:
        raised = str(e)
    assert torch.equal(st, torch.get_rng_state())
    out.append(dict(id="allzero_raises", V=V, expect=raised, rng_untouched=True))
    # max_fn rows (G3)
    mf = []
    for i, V in enumerate((257, 32000)):
        pr = ref_utils.norm_logits(logits_row(700 + i, V), 1.0, 20, 0.9)
        qr = ref_utils.norm_logits(logits_row(700 + i, V) + 0.7 * logits_row(800 + i, V, 1.0), 1.0, 20, 0.9)
        res = ref_utils.max_fn(pr - qr)
        assert torch.equal(oracle.max_fn(pr - qr), res)
        nz = np.nonzero(res.numpy()[0])[0]
        blobs[f"m{i}_idx"], blobs[f"m{i}_val"] = nz.astype(np.int32), res.numpy()[0][nz]
        mf.append(dict(id=f"m{i}", V=V, seed_p=700 + i, seed_q=800 + i, mix=0.7, T=1.0, k=20, p=0.9))
    same = ref_utils.max_fn(torch.zeros(1, 16))
    mf.append(dict(id="p_equals_q", expect_sum=float(same.sum())))
    np.savez_compressed(os.path.join(HERE, "g2_sample.npz"), **blobs)
    json.dump(dict(sample=out, max_fn=mf), open(os.path.join(HERE, "g2_sample.json"), "w"), indent=0)
    print("G2/G3:", len(out), "sample cases,", len(mf), "max_fn cases")


class TableModel:
    """Stub 'model' whose logits depend only on the absolute position: lets the reference's
    accept block (speculative_sampling.py:1964-2027) run on chosen p/q rows."""

    def __init__(self, table):
        self.table = table                      # (S_max, V) float32 logits
        from types import SimpleNamespace
        self.config = SimpleNamespace(is_encoder_decoder=False)
        self.device = torch.device("cpu")

    def __call__(self, ids, past_key_values=None, use_cache=True):
        from types import SimpleNamespace
        past = past_key_values[0][0].shape[2] if past_key_values else 0
        q = ids.shape[1]
        kv = torch.zeros(1, 1, past + q, 1)
        return SimpleNamespace(logits=self.table[past:past + q][None].clone(), past_key_values=[(kv, kv)])


def g4_accept():
    """Reference speculative_sampling over position-table models: synthetic p/q rows, gamma in {2,4,8},
    acceptance dial sigma."""
    cases, blobs = [], {}
    cid = 0
    V, L, S = 300, 6, 64
    for gamma in (2, 4, 8):
        for sigma in (0.0, 0.3, 1.0, 3.0):
            for seeded in (None, 42):
                rng = np.random.default_rng([4000, cid])
                z = rng.standard_normal((S, V), dtype=np.float32) * 2.0
                eps = rng.standard_normal((S, V), dtype=np.float32) * 2.0
                qm, pm = TableModel(torch.from_numpy(z)), TableModel(torch.from_numpy(z + np.float32(sigma) * eps))
                prompt = torch.from_numpy(rng.integers(3, V, size=(1, L)))
                _real_seed(7000 + cid)
                capture_on()
                out, d = ref_ss.speculative_sampling(prompt, qm, pm, eos_token_id=2, pad_token_id=None, max_len=20,
                                                     gamma=gamma, temperature=1, top_k=10, top_p=0.9,
                                                     random_seed=seeded, details=True)
                ev = capture_off()
                ro = oracle.speculative_sampling(prompt, qm, pm, 2, None, 20, gamma=gamma, temperature=1, top_k=10,
                                                 top_p=0.9, random_seed=seeded, details=True,
                                                 noise=oracle.RecordedNoise(ev))
                assert torch.equal(ro[0], out) and ro[1]["acc_len"] == d["acc_len"], "oracle != reference (G4)"
                for k2, v in pack_events(ev).items():
                    blobs[f"a{cid}_{k2}"] = v
                blobs[f"a{cid}_out"] = out.numpy()[0].astype(np.int32)
                blobs[f"a{cid}_prompt"] = prompt.numpy()[0].astype(np.int32)
                cases.append(dict(id=f"a{cid}", V=V, L=L, S=S, gamma=gamma, sigma=sigma, random_seed=seeded,
                                  table_seed=[4000, cid], top_k=10, top_p=0.9, max_len=20,
                                  acc_len=d["acc_len"], acc_rate=float(d["acc_rate"]),
                                  target_call_times=d["target_call_times"]))
                cid += 1
    np.savez_compressed(os.path.join(HERE, "g4_accept.npz"), **blobs)
    json.dump(cases, open(os.path.join(HERE, "g4_accept.json"), "w"), indent=0)
    print("G4:", len(cases), "cases; acc_len samples:", [c["acc_len"][:6] for c in cases[:6]])


def g8_lowprec():
    """bf16 / fp16 rows through the reference's norm_logits / sample / max_fn and its whole accept / resample loop
    (position-table models whose logits are 16-bit, as OPT's are: modeling_opt.py:974).  `tie_sensitive` marks the cases
    whose result depends on the order torch's unstable descending sort gives equal logits (oracle.sampling_ref.STABLE_TIES):
    there the HIP kernels (ties in ascending token id) legitimately differ from the recorded run."""
    import oracle.sampling_ref as SR
    cases, blobs = dict(norm=[], sample=[], max_fn=[], trace=[]), {}

    def stable(fn):
        SR.STABLE_TIES = True
        try:
            return fn()
        finally:
            SR.STABLE_TIES = False
    cid = 0
    for V in (32000, 50272, 4096):
        for (T, k, p) in [(1.0, 20, 0.9), (0.7, 50, 0.95), (1.3, 5, 0.0), (1.0, 1, 0.0), (0.8, 64, 0.99), (1.0, 0, 0.0)]:
            for dtype in (torch.bfloat16, torch.float16):
                if (k == 0) and V != 4096:
                    continue                                  # the plain-softmax row only once (dense)
                seed = 1200 + cid
                x = logits_row(seed, V, dtype=dtype)
                probs = ref_utils.norm_logits(x.clone(), T, k, p)
                assert probs.dtype == dtype and torch.equal(oracle.norm_logits(x.clone(), T, k, p), probs)
                st = stable(lambda: oracle.norm_logits(x.clone(), T, k, p))
                pf = probs.float().numpy()[0]
                nz = np.nonzero(pf)[0]
                key = f"n{cid}"
                blobs[key + "_idx"], blobs[key + "_val"] = nz.astype(np.int32), pf[nz]
                cases["norm"].append(dict(id=key, seed=seed, V=V, scale=4.0, T=T, k=k, p=p, dtype=str(dtype).split(".")[1],
                                          tie_sensitive=not torch.equal(st, probs)))
                cid += 1
    # sample on 16-bit rows: the noise is a 16-bit tensor too (empty_like(probs).exponential_)
    for i, (V, dtype) in enumerate([(32000, torch.bfloat16), (32000, torch.float16), (50272, torch.bfloat16), (4096, torch.float16)]):
        for rep in range(3):
            seed = 1500 + 10 * i + rep
            probs = ref_utils.norm_logits(logits_row(seed, V, dtype=dtype), 1.0, 20, 0.9)
            _real_seed(9500 + 10 * i + rep)
            capture_on()
            tok = ref_utils.sample(probs)
            ev = capture_off()
            assert len(ev) == 1 and ev[0][1].dtype == dtype
            key = f"s{i}_{rep}"
            blobs[key + "_noise"] = ev[0][1].float().numpy()[0]
            nz = np.nonzero(probs.float().numpy()[0])[0]
            blobs[key + "_pidx"], blobs[key + "_pval"] = nz.astype(np.int32), probs.float().numpy()[0][nz]
            cases["sample"].append(dict(id=key, V=V, dtype=str(dtype).split(".")[1], token=int(tok)))
    # max_fn(p - q) on 16-bit rows
    for i, (V, dtype) in enumerate([(32000, torch.bfloat16), (32000, torch.float16), (4096, torch.bfloat16)]):
        pr = ref_utils.norm_logits(logits_row(1700 + i, V, dtype=dtype), 1.0, 20, 0.9)
        qr = ref_utils.norm_logits((logits_row(1700 + i, V) + 0.7 * logits_row(1800 + i, V, 1.0)).to(dtype), 1.0, 20, 0.9)
        res = ref_utils.max_fn(pr - qr)
        assert res.dtype == dtype and torch.equal(oracle.max_fn(pr - qr), res)
        for nm, t in (("p", pr), ("q", qr), ("r", res)):
            a = t.float().numpy()[0]
            nz = np.nonzero(a)[0]
            blobs[f"m{i}_{nm}idx"], blobs[f"m{i}_{nm}val"] = nz.astype(np.int32), a[nz]
        cases["max_fn"].append(dict(id=f"m{i}", V=V, dtype=str(dtype).split(".")[1]))
    # the whole loop over 16-bit position tables
    tid = 0
    V, L, S = 512, 6, 64
    for dtype in (torch.bfloat16, torch.float16):
        for gamma, sigma, seeded in [(4, 0.3, None), (4, 1.0, None), (4, 0.3, 42), (2, 0.5, None), (8, 0.2, None), (4, 0.0, None)]:
            rng = np.random.default_rng([4800, tid])
            z = rng.standard_normal((S, V), dtype=np.float32) * 2.0
            eps = rng.standard_normal((S, V), dtype=np.float32) * 2.0
            qt, pt = torch.from_numpy(z).to(dtype), torch.from_numpy(z + np.float32(sigma) * eps).to(dtype)
            qm, pm = TableModel(qt), TableModel(pt)
            prompt = torch.from_numpy(rng.integers(3, V, size=(1, L)))
            _real_seed(7800 + tid)
            capture_on()
            out, d = ref_ss.speculative_sampling(prompt, qm, pm, eos_token_id=2, pad_token_id=None, max_len=24, gamma=gamma,
                                                 temperature=1, top_k=10, top_p=0.9, random_seed=seeded, details=True)
            ev = capture_off()
            assert all(e[1].dtype == dtype for e in ev if e[0] == "exp")
            ro = oracle.speculative_sampling(prompt, qm, pm, 2, None, 24, gamma=gamma, temperature=1, top_k=10, top_p=0.9,
                                             random_seed=seeded, details=True, noise=oracle.RecordedNoise(ev))
            assert torch.equal(ro[0], out) and ro[1]["acc_len"] == d["acc_len"], "oracle != reference (G8)"
            try:
                rs = stable(lambda: oracle.speculative_sampling(prompt, qm, pm, 2, None, 24, gamma=gamma, temperature=1,
                                                                top_k=10, top_p=0.9, random_seed=seeded, details=True,
                                                                noise=oracle.RecordedNoise(ev)))
                tie = not (torch.equal(rs[0], out) and rs[1]["acc_len"] == d["acc_len"])
            except Exception:
                tie = True
            ev32 = [(k2, (v.float() if k2 == "exp" else v)) for k2, v in ev]
            for k2, v in pack_events(ev32).items():
                blobs[f"t{tid}_{k2}"] = v
            blobs[f"t{tid}_out"] = out.numpy()[0].astype(np.int32)
            blobs[f"t{tid}_prompt"] = prompt.numpy()[0].astype(np.int32)
            cases["trace"].append(dict(id=f"t{tid}", V=V, L=L, S=S, gamma=gamma, sigma=sigma, random_seed=seeded,
                                       table_seed=[4800, tid], top_k=10, top_p=0.9, max_len=24, dtype=str(dtype).split(".")[1],
                                       acc_len=d["acc_len"], acc_rate=float(d["acc_rate"]),
                                       target_call_times=d["target_call_times"], tie_sensitive=tie))
            tid += 1
    np.savez_compressed(os.path.join(HERE, "g8_lowprec.npz"), **blobs)
    json.dump(cases, open(os.path.join(HERE, "g8_lowprec.json"), "w"), indent=0)
    print("G8:", {k: len(v) for k, v in cases.items()}, "tie-sensitive norm rows:",
          sum(c["tie_sensitive"] for c in cases["norm"]), "traces:", sum(c["tie_sensitive"] for c in cases["trace"]))


def _tree_inputs(rng, V, widths, parents):
    """A hand-made draft tree for get_seq_att_mask: level l has len(widths[l]) beams, beam j hangs below beam
    parents[l][j] of the previous level (level 0 hangs below the prompt)."""
    all_input_idx = [torch.zeros(len(w), dtype=torch.long) for w in widths]
    all_beam_idx = [torch.tensor(pp, dtype=torch.long) for pp in parents]
    all_next_token = [torch.from_numpy(rng.integers(3, V, size=len(w))) for w in widths]
    return all_input_idx, all_beam_idx, all_next_token


@torch.no_grad()
def g9_tree():
    """Tree attention: the reference's get_seq_att_mask, KVCacheModel.forward_tree_attention /
    rollback_tree_attention on its own model classes (extra attention mask + per-node position ids), and the
    acceptance-count recursion get_num_acc_prob / get_expect_cnt_by_thres."""
    from oracle import tree_ref
    cases, blobs = dict(tree=[], dp=[]), {}
    shapes = [([0, 1, 2], [0, 0, 0]), ([0, 1, 2], [0, 0, 2]), ([0, 1, 2], [1, 2, 2])]
    for ci, (cfg_name, seed) in enumerate([("tiny-llama-gqa", 61), ("tiny-llama-target", 62), ("tiny-opt-pre", 63), ("tiny-opt-post", 64)]):
        cfg = load_config(cfg_name)
        sd = make_state_dict(cfg, seed)
        model = ref_model(cfg, sd)
        rng = np.random.default_rng([9000, ci])
        V, P = cfg.vocab_size, 7
        prompt = torch.from_numpy(rng.integers(3, V, size=(1, P)))
        ai, ab, at = _tree_inputs(rng, V, [w for w, _ in shapes], [pp for _, pp in shapes])
        out_seq, mask, pos, pids = ref_utils.get_seq_att_mask(1, ai, ab, at, P, 0, device="cpu")
        o2 = tree_ref.get_seq_att_mask(1, ai, ab, at, P, 0)
        assert all(torch.equal(a, b) for a, b in zip((out_seq, mask, pos, pids), o2)), "oracle get_seq_att_mask != reference"
        kv = ref_kv.KVCacheModel(model, 1, 20, 0.9)
        p1 = kv.forward_tree_attention(out_seq, prompt, mask, pids, pos.clone())
        okv = oracle.RefKVCacheModel(oracle.RefCausalLM(cfg, sd), 1, 20, 0.9)
        op1 = okv.forward_tree_attention(out_seq, prompt, mask, pids, pos.clone())
        assert float((op1 - p1).abs().max()) < 1e-5, "oracle forward_tree_attention != reference"
        # accept the chain  level0 beam 0 -> level1 beam 1 -> level2 beam 0  (slots 0, 4, 6): the mask row of the leaf
        leaf = 6
        keep = mask[0, leaf][None].clone()
        kv.rollback_tree_attention(torch.tensor([0]), keep)
        okv.rollback_tree_attention(torch.tensor([0]), keep)
        k_ref = kv._past_key_values[-1][0]
        assert float((okv._past_key_values[-1][0] - k_ref).abs().max()) < 1e-5
        path = [int(out_seq[0, s_]) for s_ in torch.nonzero(mask[0, leaf, P:]).flatten().tolist()]
        # second round on the compacted cache: prompt + accepted path + one more token, a two-level tree
        prefix2 = torch.cat([prompt, torch.tensor([path + [int(rng.integers(3, V))]])], dim=1)
        P2 = prefix2.shape[1]
        ai2, ab2, at2 = _tree_inputs(rng, V, [[0, 1], [0, 1]], [[0, 0], [1, 1]])
        out2, mask2, pos2, pids2 = ref_utils.get_seq_att_mask(1, ai2, ab2, at2, P2, 0, device="cpu")
        p2 = kv.forward_tree_attention(out2, prefix2, mask2, pids2, pos2.clone())
        op2 = okv.forward_tree_attention(out2, prefix2, mask2, pids2, pos2.clone())
        assert float((op2 - p2).abs().max()) < 1e-5
        key = f"t{ci}"
        for nm, t in (("prompt", prompt), ("tok", torch.stack(at)), ("beam", torch.stack(ab)), ("seq", out_seq), ("mask", mask),
                      ("pos", pos), ("pids", pids), ("p1", p1), ("keep", keep), ("k_last", k_ref), ("hist", kv._prob_history[:, :P + 3]),
                      ("prefix2", prefix2), ("tok2", torch.stack(at2)), ("beam2", torch.stack(ab2)), ("p2", p2)):
            blobs[f"{key}_{nm}"] = t.detach().numpy()
        cases["tree"].append(dict(id=key, cfg=cfg_name, seed=seed, P=P, leaf=leaf, top_k=20, top_p=0.9))
    for di, (V, m, sg) in enumerate([(50, 2, 0.5), (50, 3, 1.0), (200, 4, 0.3), (200, 4, 2.0), (1000, 5, 0.7)]):
        rng = np.random.default_rng([9100, di])
        z = torch.from_numpy(rng.standard_normal(V, dtype=np.float32) * 2)
        q = torch.softmax(z, 0)
        p = torch.softmax(z + sg * torch.from_numpy(rng.standard_normal(V, dtype=np.float32)), 0)
        prob, expect = ref_utils.get_num_acc_prob(p, q, m)
        oprob, oexp = tree_ref.get_num_acc_prob(p, q, m)
        assert torch.allclose(oprob, prob, atol=1e-6) and abs(float(oexp) - float(expect)) < 1e-5
        cnts = [ref_utils.get_expect_cnt_by_thres(prob, th) for th in (0.3, 0.5, 0.7, 0.9)]
        assert cnts == [tree_ref.get_expect_cnt_by_thres(prob, th) for th in (0.3, 0.5, 0.7, 0.9)]
        blobs[f"d{di}_p"], blobs[f"d{di}_q"], blobs[f"d{di}_prob"] = p.numpy(), q.numpy(), prob.numpy()
        cases["dp"].append(dict(id=f"d{di}", V=V, m=m, expect=float(expect), thres=[0.3, 0.5, 0.7, 0.9], counts=cnts))
    np.savez_compressed(os.path.join(HERE, "g9_tree.npz"), **blobs)
    json.dump(cases, open(os.path.join(HERE, "g9_tree.json"), "w"), indent=0)
    print("G9:", {k: len(v) for k, v in cases.items()})


def g5_traces():
    """End-to-end token traces of the reference's own model classes on tiny configs."""
    cases, blobs = [], {}
    specs = [
        # id, draft cfg, draft seed, target cfg, target spec, kwargs
        ("llama_corr", "tiny-llama-target", 11, "tiny-llama-target", ("perturb", 12, 0.12), dict(top_k=20, top_p=0.9)),
        ("llama_corr_plain", "tiny-llama-target", 11, "tiny-llama-target", ("perturb", 12, 0.05), dict(top_k=0, top_p=0)),
        ("llama_same", "tiny-llama-target", 11, "tiny-llama-target", ("same",), dict(top_k=20, top_p=0.9)),
        ("llama_unrelated", "tiny-llama-draft", 21, "tiny-llama-target", ("seed", 22), dict(top_k=20, top_p=0.9)),
        ("llama_seeded", "tiny-llama-target", 11, "tiny-llama-target", ("perturb", 12, 0.12), dict(top_k=20, top_p=0.9, random_seed=42)),
        ("llama_gqa", "tiny-llama-gqa", 31, "tiny-llama-gqa", ("perturb", 32, 0.1), dict(top_k=10, top_p=0.95, temperature=0.8)),
        ("llama_gamma8", "tiny-llama-target", 11, "tiny-llama-target", ("perturb", 12, 0.08), dict(top_k=20, top_p=0.9, gamma=8)),
        ("llama_gamma2", "tiny-llama-draft", 21, "tiny-llama-target", ("seed", 22), dict(top_k=5, top_p=0.0, gamma=2)),
        ("opt_pre_corr", "tiny-opt-pre", 41, "tiny-opt-pre", ("perturb", 42, 0.1), dict(top_k=20, top_p=0.9)),
        ("opt_post_pair", "tiny-opt-pre", 41, "tiny-opt-post", ("seed", 43), dict(top_k=20, top_p=0.9)),
        ("opt_post_corr", "tiny-opt-post", 44, "tiny-opt-post", ("perturb", 45, 0.1), dict(top_k=0, top_p=0.9)),
    ]
    for cid, (name, dcfg_n, dseed, tcfg_n, tspec, kw) in enumerate(specs):
        dcfg, tcfg = load_config(dcfg_n), load_config(tcfg_n)
        dsd = make_state_dict(dcfg, dseed)
        if tspec[0] == "same":
            tsd = dsd
        elif tspec[0] == "perturb":
            tsd = perturb_state_dict(dsd, tspec[1], tspec[2])
        else:
            tsd = make_state_dict(tcfg, tspec[1])
        dm, tm = ref_model(dcfg, dsd), ref_model(tcfg, tsd)
        V = dcfg.vocab_size
        rng = np.random.default_rng([5000, cid])
        L = 10 + cid
        prompt = torch.from_numpy(rng.integers(3, V, size=(1, L)))
        eos = 2
        _real_seed(123 + cid)
        capture_on()
        out, d = ref_ss.speculative_sampling(prompt, dm, tm, eos_token_id=eos, pad_token_id=None, max_len=24,
                                             details=True, **kw)
        ev = capture_off()
        # the oracle (own forwards + own loops) replays the recorded stream and must reproduce the reference
        od, ot = oracle.RefCausalLM(dcfg, dsd), oracle.RefCausalLM(tcfg, tsd)
        ro = oracle.speculative_sampling(prompt, od, ot, eos, None, 24, details=True,
                                         noise=oracle.RecordedNoise(ev), **kw)
        assert torch.equal(ro[0], out), ("oracle != reference (G5)", name, ro[0], out)
        assert ro[1]["acc_len"] == d["acc_len"]
        for k2, v in pack_events(ev).items():
            blobs[f"{name}_{k2}"] = v
        blobs[f"{name}_out"] = out.numpy()[0].astype(np.int32)
        blobs[f"{name}_prompt"] = prompt.numpy()[0].astype(np.int32)
        cases.append(dict(id=name, draft_cfg=dcfg_n, draft_seed=dseed, target_cfg=tcfg_n, target_spec=list(tspec),
                          kwargs=kw, eos=eos, max_len=24, L=L, outer_seed=123 + cid,
                          acc_len=d["acc_len"], acc_rate=float(d["acc_rate"]),
                          target_call_times=d["target_call_times"], approx_call_times=d["approx_call_times"],
                          rows_fed_draft=ro[1]["_rows_fed_draft"], rows_fed_target=ro[1]["_rows_fed_target"],
                          out_len=int(out.shape[1])))
        print("  G5", name, "out_len", out.shape[1], "acc_len", d["acc_len"])

    # EOS cases: prompt already holds an EOS, and the models are steered to emit EOS quickly
    dcfg = load_config("tiny-llama-target")
    dsd = make_state_dict(dcfg, 11)
    dsd = {k: v.clone() for k, v in dsd.items()}
    dsd["lm_head.weight"][2] += dsd["lm_head.weight"].abs().mean() * 0.0   # keep identical; eos chosen from the trace
    tsd = perturb_state_dict(dsd, 12, 0.12)
    dm, tm = ref_model(dcfg, dsd), ref_model(dcfg, tsd)
    base = [c for c in cases if c["id"] == "llama_corr"][0]
    base_out = blobs["llama_corr_out"]
    L0 = base["L"]
    eos_tok = int(base_out[L0 + 5])            # a token the trace is known to generate: now it is EOS
    prompt = torch.from_numpy(blobs["llama_corr_prompt"].astype(np.int64))[None].clone()
    for name, pr in (("eos_generated", prompt), ("eos_in_prompt", torch.cat([prompt[:, :3], torch.tensor([[eos_tok]]), prompt[:, 3:]], 1))):
        _real_seed(123)
        capture_on()
        out, d = ref_ss.speculative_sampling(pr, dm, tm, eos_token_id=eos_tok, pad_token_id=None, max_len=24,
                                             top_k=20, top_p=0.9, details=True)
        ev = capture_off()
        od, ot = oracle.RefCausalLM(dcfg, dsd), oracle.RefCausalLM(dcfg, tsd)
        ro = oracle.speculative_sampling(pr, od, ot, eos_tok, None, 24, top_k=20, top_p=0.9, details=True,
                                         noise=oracle.RecordedNoise(ev))
        assert torch.equal(ro[0], out), name
        for k2, v in pack_events(ev).items():
            blobs[f"{name}_{k2}"] = v
        blobs[f"{name}_out"] = out.numpy()[0].astype(np.int32)
        blobs[f"{name}_prompt"] = pr.numpy()[0].astype(np.int32)
        cases.append(dict(id=name, draft_cfg="tiny-llama-target", draft_seed=11, target_cfg="tiny-llama-target",
                          target_spec=["perturb", 12, 0.12], kwargs=dict(top_k=20, top_p=0.9), eos=eos_tok,
                          max_len=24, L=int(pr.shape[1]), outer_seed=123, acc_len=d["acc_len"],
                          acc_rate=float(d["acc_rate"]), target_call_times=d["target_call_times"],
                          approx_call_times=d["approx_call_times"],
                          rows_fed_draft=ro[1]["_rows_fed_draft"], rows_fed_target=ro[1]["_rows_fed_target"],
                          out_len=int(out.shape[1])))
        print("  G5", name, "out_len", out.shape[1], "eos", eos_tok, "acc_len", d["acc_len"])

    # autoregressive_sampling traces (A9)
    ar = []
    for cid, (cfg_n, seed, kw) in enumerate([("tiny-llama-target", 12, dict(top_k=20, top_p=0.9)),
                                             ("tiny-opt-post", 43, dict(top_k=0, top_p=0.0, temperature=0.9))]):
        cfg = load_config(cfg_n)
        sd = make_state_dict(cfg, seed)
        m = ref_model(cfg, sd)
        rng = np.random.default_rng([6000, cid])
        prompt = torch.from_numpy(rng.integers(3, cfg.vocab_size, size=(1, 9)))
        _real_seed(77 + cid)
        capture_on()
        out = ref_ar.autoregressive_sampling(prompt, m, 16, eos_token_id=2, **kw)
        ev = capture_off()
        ro = oracle.autoregressive_sampling(prompt, oracle.RefCausalLM(cfg, sd), 16, 2,
                                            noise=oracle.RecordedNoise(ev), **kw)
        assert torch.equal(ro, out), "oracle AR != reference"
        for k2, v in pack_events(ev).items():
            blobs[f"ar{cid}_{k2}"] = v
        blobs[f"ar{cid}_out"] = out.numpy()[0].astype(np.int32)
        blobs[f"ar{cid}_prompt"] = prompt.numpy()[0].astype(np.int32)
        ar.append(dict(id=f"ar{cid}", cfg=cfg_n, seed=seed, kwargs=kw, N=16, eos=2, out_len=int(out.shape[1])))
    np.savez_compressed(os.path.join(HERE, "g5_traces.npz"), **blobs)
    json.dump(dict(spec=cases, ar=ar), open(os.path.join(HERE, "g5_traces.json"), "w"), indent=0)
    print("G5:", len(cases), "speculative traces,", len(ar), "AR traces")


def g6_logits():
    """Reference model classes: prefill + incremental logits (q in {1,2,5}) at tiny configs, fp32 and bf16."""
    cases, blobs = [], {}
    for cid, (cfg_n, seed) in enumerate([("tiny-llama-target", 12), ("tiny-llama-draft", 21), ("tiny-llama-gqa", 31),
                                         ("tiny-opt-pre", 41), ("tiny-opt-post", 43)]):
        cfg = load_config(cfg_n)
        for dtype in (torch.float32, torch.bfloat16):
            sd = make_state_dict(cfg, seed, dtype=dtype)
            m = ref_model(cfg, sd)
            om = oracle.RefCausalLM(cfg, sd)
            rng = np.random.default_rng([6500, cid])
            ids = torch.from_numpy(rng.integers(3, cfg.vocab_size, size=(1, 20)))
            splits = [12, 1, 2, 5]
            past, opast, pos = None, None, 0
            tag = f"{cfg_n}_{str(dtype).split('.')[1]}"
            for si, q in enumerate(splits):
                chunk = ids[:, pos:pos + q]
                with torch.no_grad():
                    r = m(chunk, past_key_values=past, use_cache=True) if past is not None else m(chunk)
                o = om(chunk, past_key_values=opast)
                err = (r.logits.float() - o.logits.float()).abs().max().item()
                tol = 2e-5 if dtype == torch.float32 else 0.25
                assert err <= tol, ("oracle forward != reference forward", tag, si, err)
                assert r.logits.dtype == o.logits.dtype
                blobs[f"{tag}_s{si}"] = r.logits.float().numpy()[0]
                past, opast, pos = r.past_key_values, o.past_key_values, pos + q
            kshape = tuple(past[0][0].shape)
            blobs[f"{tag}_ids"] = ids.numpy()[0].astype(np.int32)
            cases.append(dict(id=tag, cfg=cfg_n, seed=seed, dtype=str(dtype).split(".")[1], splits=splits,
                              kv_shape=list(kshape), logits_dtype=str(r.logits.dtype).split(".")[1]))
            print("  G6", tag, "kv", kshape, "logits", r.logits.dtype)
    np.savez_compressed(os.path.join(HERE, "g6_logits.npz"), **blobs)
    json.dump(cases, open(os.path.join(HERE, "g6_logits.json"), "w"), indent=0)
    print("G6:", len(cases), "cases")


def g7_multi():
    """multi_speculative_sampling(strategy='iid') token traces of the reference's own model classes
    (SURVEY.md section 8(f) rank 2): width-w drafts, batched target forward, rollback(end, choice)."""
    ref_multi = ref_ss.multi_speculative_sampling
    cases, blobs = [], {}
    specs = [
        # id, draft cfg, draft seed, target cfg, target spec, width, kwargs
        ("m_llama_corr", "tiny-llama-target", 11, "tiny-llama-target", ("perturb", 12, 0.12), 3, dict(top_k=20, top_p=0.9)),
        ("m_llama_same", "tiny-llama-target", 11, "tiny-llama-target", ("same",), 2, dict(top_k=20, top_p=0.9)),
        ("m_llama_unrelated", "tiny-llama-draft", 21, "tiny-llama-target", ("seed", 22), 4, dict(top_k=20, top_p=0.9, gamma=3)),
        ("m_llama_seeded", "tiny-llama-target", 11, "tiny-llama-target", ("perturb", 12, 0.2), 3, dict(top_k=20, top_p=0.9, random_seed=42)),
        ("m_llama_gqa_w1", "tiny-llama-gqa", 31, "tiny-llama-gqa", ("perturb", 32, 0.1), 1, dict(top_k=10, top_p=0.95, temperature=0.8)),
        ("m_opt_post_pair", "tiny-opt-pre", 41, "tiny-opt-post", ("seed", 43), 3, dict(top_k=20, top_p=0.9, gamma=2)),
        ("m_opt_pre_corr", "tiny-opt-pre", 41, "tiny-opt-pre", ("perturb", 42, 0.1), 2, dict(top_k=0, top_p=0.9)),
    ]
    for cid, (name, dcfg_n, dseed, tcfg_n, tspec, width, kw) in enumerate(specs):
        dcfg, tcfg = load_config(dcfg_n), load_config(tcfg_n)
        dsd = make_state_dict(dcfg, dseed)
        tsd = dsd if tspec[0] == "same" else (perturb_state_dict(dsd, tspec[1], tspec[2]) if tspec[0] == "perturb"
                                              else make_state_dict(tcfg, tspec[1]))
        dm, tm = ref_model(dcfg, dsd), ref_model(tcfg, tsd)
        rng = np.random.default_rng([7000, cid])
        L = 8 + cid
        prompt = torch.from_numpy(rng.integers(3, dcfg.vocab_size, size=(1, L)))
        eos = 2
        _real_seed(321 + cid)
        capture_on()
        out, d = ref_multi(prompt, dm, tm, eos_token_id=eos, pad_token_id=None, max_len=14, width=width,
                           strategy="iid", details=True, **kw)
        ev = capture_off()
        od, ot = oracle.RefCausalLM(dcfg, dsd), oracle.RefCausalLM(tcfg, tsd)
        ro = oracle.multi_speculative_sampling(prompt, od, ot, eos, None, 14, width=width, strategy="iid",
                                               details=True, noise=oracle.RecordedNoise(ev), **kw)
        assert torch.equal(ro[0], out), ("oracle != reference (G7)", name, ro[0], out)
        assert ro[1]["acc_len"] == d["acc_len"] and abs(ro[1]["acc_rate"] - d["acc_rate"]) < 1e-12
        for k2, v in pack_events_ragged(ev).items():
            blobs[f"{name}_{k2}"] = v
        blobs[f"{name}_out"] = out.numpy()[0].astype(np.int32)
        blobs[f"{name}_prompt"] = prompt.numpy()[0].astype(np.int32)
        cases.append(dict(id=name, draft_cfg=dcfg_n, draft_seed=dseed, target_cfg=tcfg_n, target_spec=list(tspec),
                          width=width, kwargs=kw, eos=eos, max_len=14, L=L, outer_seed=321 + cid,
                          acc_len=d["acc_len"], acc_rate=float(d["acc_rate"]),
                          target_call_times=d["target_call_times"], approx_call_times=d["approx_call_times"],
                          rows_fed_draft=ro[1]["_rows_fed_draft"], rows_fed_target=ro[1]["_rows_fed_target"],
                          out_len=int(out.shape[1])))
        print("  G7", name, "width", width, "out_len", out.shape[1], "acc_len", d["acc_len"])
    # EOS stop: a token the first trace is known to emit becomes EOS
    base_out, L0 = blobs["m_llama_corr_out"], cases[0]["L"]
    eos_tok = int(base_out[L0 + 4])
    dcfg = load_config("tiny-llama-target")
    dsd = make_state_dict(dcfg, 11)
    tsd = perturb_state_dict(dsd, 12, 0.12)
    dm, tm = ref_model(dcfg, dsd), ref_model(dcfg, tsd)
    prompt = torch.from_numpy(blobs["m_llama_corr_prompt"].astype(np.int64))[None].clone()
    _real_seed(321)
    capture_on()
    out, d = ref_multi(prompt, dm, tm, eos_token_id=eos_tok, pad_token_id=None, max_len=14, width=3, strategy="iid",
                       top_k=20, top_p=0.9, details=True)
    ev = capture_off()
    ro = oracle.multi_speculative_sampling(prompt, oracle.RefCausalLM(dcfg, dsd), oracle.RefCausalLM(dcfg, tsd), eos_tok,
                                           None, 14, width=3, strategy="iid", top_k=20, top_p=0.9, details=True,
                                           noise=oracle.RecordedNoise(ev))
    assert torch.equal(ro[0], out), "oracle != reference (G7 eos)"
    for k2, v in pack_events_ragged(ev).items():
        blobs[f"m_eos_{k2}"] = v
    blobs["m_eos_out"] = out.numpy()[0].astype(np.int32)
    blobs["m_eos_prompt"] = prompt.numpy()[0].astype(np.int32)
    cases.append(dict(id="m_eos", draft_cfg="tiny-llama-target", draft_seed=11, target_cfg="tiny-llama-target",
                      target_spec=["perturb", 12, 0.12], width=3, kwargs=dict(top_k=20, top_p=0.9), eos=eos_tok,
                      max_len=14, L=int(prompt.shape[1]), outer_seed=321, acc_len=d["acc_len"],
                      acc_rate=float(d["acc_rate"]), target_call_times=d["target_call_times"],
                      approx_call_times=d["approx_call_times"], rows_fed_draft=ro[1]["_rows_fed_draft"],
                      rows_fed_target=ro[1]["_rows_fed_target"], out_len=int(out.shape[1])))
    print("  G7 m_eos out_len", out.shape[1], "eos", eos_tok)
    np.savez_compressed(os.path.join(HERE, "g7_multi.npz"), **blobs)
    json.dump(cases, open(os.path.join(HERE, "g7_multi.json"), "w"), indent=0)
    print("G7:", len(cases), "multi traces")


def g10_filter_lowprec():
    """top_k_top_p_filter called directly on bf16 / fp16 rows (reference utils.py:152-179): the reference sorts, softmaxes
    and cumsums in the tensor's dtype, so the kept set near the top-p cut is decided on 16-bit sums (the
    stand-alone filter had no dtype mode; G8 only covered it through norm_logits).  Records the kept indices;
    `tie_sensitive` as in G8."""
    import oracle.sampling_ref as SR
    cases, blobs = [], {}
    cid = 0
    for V in (32000, 50272, 4096):
        for (k, p) in [(20, 0.9), (50, 0.95), (0, 0.8), (64, 0.99), (5, 0.0), (0, 0.5)]:
            for dtype in (torch.bfloat16, torch.float16):
                seed = 2600 + cid
                x = logits_row(seed, V, dtype=dtype)
                out = ref_utils.top_k_top_p_filter(x.clone(), top_k=k, top_p=p)
                assert out.dtype == dtype and torch.equal(oracle.top_k_top_p_filter(x.clone(), k, p), out)
                SR.STABLE_TIES = True
                try:
                    st = oracle.top_k_top_p_filter(x.clone(), k, p)
                finally:
                    SR.STABLE_TIES = False
                kept = np.nonzero(torch.isfinite(out[0]).numpy())[0].astype(np.int32)
                # Does the kept set hinge on the ORDER in which torch's vectorised CPU softmax sums its fp32 denominator
                # over the row?  Redo the filter with an exactly rounded softmax (float64, then the row dtype): if that
                # moves the cut, no other implementation can be expected to reproduce torch's last bit there.
                alt = None
                if p > 0:
                    zz = x.clone()
                    if k > 0:
                        zz[zz < torch.topk(zz, min(k, V))[0][..., -1, None]] = float("-inf")
                    srt, order = torch.sort(zz, descending=True)
                    cs = torch.cumsum(torch.softmax(srt.double(), dim=-1).to(dtype), dim=-1)
                    rem = cs > p
                    rem[..., 1:] = rem[..., :-1].clone()
                    rem[..., 0] = False
                    zz[0, order[rem]] = float("-inf")
                    alt = np.nonzero(torch.isfinite(zz[0]).numpy())[0].astype(np.int32)
                    if alt.size == kept.size and (alt == kept).all():
                        alt = None
                # the fp32 run of the same (16-bit-valued) row: where it keeps a different set the dtype mode matters
                k32 = np.nonzero(torch.isfinite(ref_utils.top_k_top_p_filter(x.float().clone(), top_k=k, top_p=p)[0]).numpy())[0]
                key = f"f{cid}"
                blobs[key + "_kept"] = kept
                if alt is not None:
                    blobs[key + "_kept_exact_softmax"] = alt
                cases.append(dict(id=key, seed=seed, V=V, scale=4.0, k=k, p=p, dtype=str(dtype).split(".")[1],
                                  n_kept=int(kept.size), sum_order_sensitive=alt is not None, differs_from_fp32=bool(k32.size != kept.size or (k32 != kept).any()),
                                  tie_sensitive=not torch.equal(st, out)))
                cid += 1
    np.savez_compressed(os.path.join(HERE, "g10_filter_lowprec.npz"), **blobs)
    json.dump(cases, open(os.path.join(HERE, "g10_filter_lowprec.json"), "w"), indent=0)
    print("G10:", len(cases), "rows; kept set differs from the fp32 run in", sum(c["differs_from_fp32"] for c in cases),
          "; tie-sensitive", sum(c["tie_sensitive"] for c in cases), "; softmax-sum-order sensitive",
          [c["id"] for c in cases if c["sum_order_sensitive"]])


def misc():
    """Facts the design leans on, recorded from the live reference environment."""
    facts = {"torch": torch.__version__, "transformers": transformers.__version__}
    r = torch.tensor([0.3000000119])
    facts["cmp_is_fp32"] = bool(not (r > 0.30000001)) and bool(r > 0.2999999)   # the python double is rounded to fp32
    x = torch.tensor([[1., 2., 2., 1., 2.]])
    facts["sort_desc_stable_indices"] = torch.sort(x, descending=True)[1][0].tolist()
    x = torch.tensor([[3.0, 1.0, 2.0, 2.0, 0.5, 2.0, 2.0, -1.0]])
    facts["topk2_keeps"] = int(torch.isfinite(ref_utils.top_k_top_p_filter(x.clone(), top_k=2)).sum())
    json.dump(facts, open(os.path.join(HERE, "facts.json"), "w"), indent=0)
    print("facts:", facts)


if __name__ == "__main__":
    todo = dict(misc=misc, g1=g1_norm_logits, g2=g2_sample_maxfn, g4=g4_accept, g5=g5_traces, g6=g6_logits, g7=g7_multi, g8=g8_lowprec, g9=g9_tree, g10=g10_filter_lowprec)
    for name in (sys.argv[1:] or list(todo)):             # e.g. `make_golden.py g7` regenerates one fixture set
        todo[name]()
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE))
    print("fixtures total bytes:", tot)
