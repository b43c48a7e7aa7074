"""Round-2 parity tests (`pytest -m gpu`): the paths bench.py actually times - the native device-RNG iteration
(sd_spec_iteration), the stream-batched loop and the bf16 fused kernels at the headline shapes - held to the CPU oracle,
plus the drop-in boundary (HF modules, KVCacheModel multi / choice, bad token ids, top_k_top_p_filter).

The device RNG is replayed into the oracle through sd_philox_exp / sd_philox_uniform (tests/philox_replay.py), so the
comparison is token for token, not statistical.
"""
import os
import time

import numpy as np
import pytest
import torch

import oracle
from golden_io import logits_row
from philox_replay import PhiloxOracleNoise
from llmspeculativesampling_amd.config import ModelConfig, load_config
from llmspeculativesampling_amd.synth import make_state_dict, perturb_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import types
    import llmspeculativesampling_amd.sampling as S
    from llmspeculativesampling_amd import _lib, engine, noise
    return types.SimpleNamespace(S=S, lib=_lib.lib, L=_lib, engine=engine, noise=noise)


def _st():
    return torch.cuda.current_stream().cuda_stream


def _pair(kind, seed=11):
    """(cfg_d, sd_d, cfg_t, sd_t) fp32 tiny pairs: correlated / identical / unrelated Llama, OPT pre-LN -> post-LN."""
    if kind == "opt":
        dc, tc = load_config("tiny-opt-pre"), load_config("tiny-opt-post")
        return dc, make_state_dict(dc, seed), tc, make_state_dict(tc, seed + 1)
    cfg = load_config("tiny-llama-target")
    dsd = make_state_dict(cfg, seed)
    if kind == "same":
        return cfg, dsd, cfg, dsd
    if kind == "unrelated":
        return cfg, dsd, cfg, make_state_dict(cfg, seed + 5)
    return cfg, dsd, cfg, perturb_state_dict(dsd, seed + 1, 0.12)


# --------------------------------------------------------------------------- Philox stream itself
def test_philox_variates_are_strictly_positive_and_in_range(hip):
    """u = (23 bits + 1/2) * 2^-23 lies strictly inside (0,1), so Exp(1) = -log(u) is finite and > 0 for
    every counter; the uniforms of the accept scan lie in [0,1).  Swept over 64 draws x 32000 elements."""
    V = 32000
    e = torch.empty(V, dtype=torch.float32, device="cuda")
    lo, hi = float("inf"), 0.0
    for d in range(64):
        assert hip.lib.sd_philox_exp(987654321 + d, d * 7919, V, e.data_ptr(), _st()) == 0
        lo, hi = min(lo, float(e.min())), max(hi, float(e.max()))
    assert lo > 0.0 and np.isfinite(hi) and hi < 17.0            # -log(2^-24) = 16.6
    u = torch.empty(4096, dtype=torch.float32, device="cuda")
    assert hip.lib.sd_philox_uniform(5, 0, 4096, u.data_ptr(), _st()) == 0
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0 and 0.45 < float(u.mean()) < 0.55
    m = float(torch.stack([e]).mean())
    assert 0.95 < m < 1.05                                       # Exp(1) mean


# --------------------------------------------------------------------------- native iteration vs the oracle
NATIVE_CASES = [
    ("corr_g4", "corr", dict(gamma=4, top_k=20, top_p=0.9), 2, 24),
    ("corr_g2", "corr", dict(gamma=2, top_k=20, top_p=0.9), 2, 24),
    ("corr_g8", "corr", dict(gamma=8, top_k=20, top_p=0.9), 2, 30),
    ("corr_plain_softmax", "corr", dict(gamma=4, top_k=0, top_p=0.0), 2, 20),
    ("same_all_accept", "same", dict(gamma=4, top_k=10, top_p=0.0), 2, 24),
    ("unrelated", "unrelated", dict(gamma=4, top_k=20, top_p=0.9), 2, 16),
    ("seeded_quirk", "corr", dict(gamma=4, top_k=20, top_p=0.9, random_seed=42), 2, 24),
    ("opt_pre_to_post", "opt", dict(gamma=4, top_k=20, top_p=0.9), 2, 20),
    ("eos_stop", "corr", dict(gamma=4, top_k=5, top_p=0.0), None, 40),     # eos chosen from the oracle's own output
]


@pytest.mark.parametrize("name,kind,kw,eos,max_len", NATIVE_CASES, ids=[c[0] for c in NATIVE_CASES])
def test_native_device_loop_equals_oracle_on_the_device_rng_stream(hip, name, kind, kw, eos, max_len):
    """speculative_sampling(rng=DeviceNoise(seed)) -> _native_device_loop -> sd_spec_iteration (what bench.py times)
    against oracle.speculative_sampling fed the SAME Philox variates: ids, acc_len, call counts identical (fp32)."""
    dc, dsd, tc, tsd = _pair(kind)
    prompt = torch.from_numpy(np.random.default_rng(5).integers(3, dc.vocab_size, size=(1, 13)))
    seed = 4242
    if eos is None:                                              # pick a token the run really generates
        nz = PhiloxOracleNoise(hip.lib, seed, kw["gamma"], _st)
        full = oracle.speculative_sampling(prompt, oracle.RefCausalLM(dc, dsd), oracle.RefCausalLM(tc, tsd), -1, None,
                                           max_len, noise=nz, **kw)
        eos = int(full[0, 13 + 9])
    nz = PhiloxOracleNoise(hip.lib, seed, kw["gamma"], _st)
    want, wd = oracle.speculative_sampling(prompt, oracle.RefCausalLM(dc, dsd), oracle.RefCausalLM(tc, tsd), eos, None,
                                           max_len, details=True, noise=nz, **kw)
    dm = hip.engine.SpecDecModel.from_state_dict(dc, dsd, dtype=torch.float32)
    tm = dm if kind == "same" else hip.engine.SpecDecModel.from_state_dict(tc, tsd, dtype=torch.float32)
    got, gd = hip.S.speculative_sampling(prompt.cuda(), dm, tm, eos, None, max_len, details=True,
                                         rng=hip.noise.DeviceNoise(seed), **kw)
    np.testing.assert_array_equal(got.cpu().numpy(), want.numpy())
    assert gd["acc_len"] == wd["acc_len"]
    assert gd["target_call_times"] == wd["target_call_times"] and gd["approx_call_times"] == wd["approx_call_times"]
    assert abs(float(gd["acc_rate"]) - float(wd["acc_rate"])) < 1e-4
    if name == "eos_stop":
        assert got.shape[1] < 13 + max_len and int(got[0, -1]) == eos
    if name == "same_all_accept":
        assert all(a == kw["gamma"] for a in gd["acc_len"])
    # native mode reports the device time of the two phases (HIP events) where the reference reports process_time
    assert gd["approx_time"] > 0 and gd["target_time"] > 0 and gd["target_model_time"] == gd["target_time"]


@pytest.mark.parametrize("gamma,n_streams", [(4, 4), (2, 4), (8, 8)], ids=["g4x4", "g2x4", "g8x8_two_passes"])
def test_stream_batched_loop_equals_oracle_per_stream(hip, gamma, n_streams):
    """speculative_sampling_batch (throughput mode: B streams through shared weight passes, the native lock-step loop
    sd_spec_batch_generate) against B separate oracle runs, each fed its own stream's Philox variates; one stream stops
    at EOS while the others continue.  gamma = 2 / 4 / 8 is config 4's sweep; 8 streams x 9 rows = 72 verify rows exceed
    one pass's row budget, so that case also covers the loop's split into passes of whole streams."""
    dc, dsd, tc, tsd = _pair("corr", seed=21)
    V = dc.vocab_size
    prompts = [torch.from_numpy(np.random.default_rng(40 + i).integers(3, V, size=(1, 9 + 2 * i))) for i in range(n_streams)]
    seeds = [900 + i for i in range(n_streams)]
    kw = dict(gamma=gamma, top_k=20, top_p=0.9)
    od, ot = oracle.RefCausalLM(dc, dsd), oracle.RefCausalLM(tc, tsd)
    probe = oracle.speculative_sampling(prompts[1], od, ot, -1, None, 24, noise=PhiloxOracleNoise(hip.lib, seeds[1], gamma, _st), **kw)
    eos = int(probe[0, prompts[1].shape[1] + 6])
    wants = []
    for p, sd_ in zip(prompts, seeds):
        wants.append(oracle.speculative_sampling(p, od, ot, eos, None, 24, details=True,
                                                 noise=PhiloxOracleNoise(hip.lib, sd_, gamma, _st), **kw))
    dm = hip.engine.SpecDecModel.from_state_dict(dc, dsd, dtype=torch.float32)
    tm = hip.engine.SpecDecModel.from_state_dict(tc, tsd, dtype=torch.float32)
    outs, ds = hip.S.speculative_sampling_batch([p.cuda() for p in prompts], dm, tm, eos, None, 24, details=True,
                                                seeds=seeds, **kw)
    stopped = 0
    for (want, wd), got, gd in zip(wants, outs, ds):
        np.testing.assert_array_equal(got.cpu().numpy(), want.numpy())
        assert gd["acc_len"] == wd["acc_len"] and gd["target_call_times"] == wd["target_call_times"]
        stopped += int(want[0, -1]) == eos
    assert 1 <= stopped < n_streams
    assert 0 < sum(sum(wd["acc_len"]) for _, wd in wants) < gamma * sum(len(wd["acc_len"]) for _, wd in wants)


def test_autoregressive_device_rng_equals_oracle(hip):
    cfg = load_config("tiny-llama-target")
    sd = make_state_dict(cfg, 31)
    prompt = torch.from_numpy(np.random.default_rng(6).integers(3, cfg.vocab_size, size=(1, 11)))
    want = oracle.autoregressive_sampling(prompt, oracle.RefCausalLM(cfg, sd), 20, -1, 1.0, 20, 0.9,
                                          noise=PhiloxOracleNoise(hip.lib, 77, 1, _st))
    m = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=torch.float32)
    got = hip.S.autoregressive_sampling(prompt.cuda(), m, 20, -1, 1.0, 20, 0.9, rng=hip.noise.DeviceNoise(77))
    np.testing.assert_array_equal(got.cpu().numpy(), want.numpy())


# --------------------------------------------------------------------------- bf16 native path
BF16_CFG = dict(arch="llama", vocab_size=8192, hidden_size=256, intermediate_size=704, num_hidden_layers=2,
                num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=512, rms_norm_eps=1e-6)


@pytest.mark.parametrize("case", ["perturbed", "all_accept", "long_prompt", "split_lm_head"])
def test_native_bf16_iteration_bit_equals_python_loop(hip, case, capsys):
    """The benchmarked configuration (bf16 weights, sd_spec_iteration, device Philox, vocab wide enough for the
    16-workgroup norm fast path, lm_head output slab handed straight to the norm) against the Python-orchestrated HIP
    loop (verbose=True) under the same Philox seed: tokens, acc_len and acc_rate must be bit-equal.  Covers a pair with
    partial accepts, an all-accept pair (rollback(n+2), 2-row draft step, bonus sample), a prompt longer than one
    256-row prefill chunk, and the logits_kernel fallback (lm_head forced to split its k-range via SD_GEMM_UNITS)."""
    cfg = ModelConfig(**BF16_CFG)
    dsd = make_state_dict(cfg, 5, dtype=torch.bfloat16)
    if case == "all_accept":
        tsd = dsd
    else:
        tsd = {k: v.to(torch.bfloat16) for k, v in perturb_state_dict({a: b.float() for a, b in dsd.items()}, 6, 0.05).items()}
    L = 300 if case == "long_prompt" else 24
    prompt = torch.from_numpy(np.random.default_rng(3).integers(3, cfg.vocab_size, size=(1, L))).cuda()
    dm = hip.engine.SpecDecModel.from_state_dict(cfg, dsd, dtype=torch.bfloat16)
    tm = dm if case == "all_accept" else hip.engine.SpecDecModel.from_state_dict(cfg, tsd, dtype=torch.bfloat16)
    if case == "split_lm_head":
        os.environ["SD_GEMM_UNITS"] = "1536"                     # 512 n-tiles -> 3 k-slabs: the head leaves partial slabs
    try:
        kw = dict(gamma=4, top_k=20, top_p=0.9)
        a, da = hip.S.speculative_sampling(prompt, dm, tm, -1, None, 40, details=True, rng=hip.noise.DeviceNoise(123), **kw)
        b, db = hip.S.speculative_sampling(prompt, dm, tm, -1, None, 40, details=True, rng=hip.noise.DeviceNoise(123),
                                           verbose=True, **kw)
    finally:
        os.environ.pop("SD_GEMM_UNITS", None)
    capsys.readouterr()
    assert torch.equal(a, b)
    assert da["acc_len"] == db["acc_len"] and da["target_call_times"] == db["target_call_times"]
    assert float(da["acc_rate"]) == float(db["acc_rate"])
    if case == "all_accept":
        assert all(x == 4 for x in da["acc_len"])
    if case == "perturbed":
        assert 0 < sum(da["acc_len"]) < 4 * len(da["acc_len"])   # partial accepts really occur


@pytest.mark.parametrize("case", ["perturbed_g4", "all_accept_g4", "perturbed_g8_two_passes", "seeded_g2", "split_lm_head"])
def test_stream_batched_bf16_fused_tail_bit_equals_the_dense_tail(hip, case):
    """The lock-step loop's sampling tail (round 4): the draft head leaves tile maxima and clears the streams' probability
    rows (EPI_HEAD with one pointer per stream), the norm + sample runs on them without a logits copy or a candidate pass,
    the target rows come back with candidate lists and the accept scan + residual / bonus sample is ONE launch on them.
    Against the round-1 tail (SD_BATCH_FUSED_TAIL=0: logits copy, norm_cand + norm_probs, dense accept scan + resample)
    under the same Philox seeds every stream's tokens, accepted lengths and acceptance ratios must be bit-equal - bf16
    weights, a vocabulary wide enough for the tile path, streams of different prompt lengths, one of which stops at EOS;
    16 streams x 9 rows need two verify passes (the lists then stay off and the loop takes the dense accept)."""
    cfg = ModelConfig(**BF16_CFG)
    dsd = make_state_dict(cfg, 5, dtype=torch.bfloat16)
    tsd = dsd if case.startswith("all_accept") else \
        {k: v.to(torch.bfloat16) for k, v in perturb_state_dict({a: b.float() for a, b in dsd.items()}, 6, 0.05).items()}
    dm = hip.engine.SpecDecModel.from_state_dict(cfg, dsd, dtype=torch.bfloat16)
    tm = dm if case.startswith("all_accept") else hip.engine.SpecDecModel.from_state_dict(cfg, tsd, dtype=torch.bfloat16)
    gamma = 8 if "g8" in case else (2 if "g2" in case else 4)
    B = 16 if "two_passes" in case else 6
    rng = np.random.default_rng(17)
    prompts = [torch.from_numpy(rng.integers(3, cfg.vocab_size, size=(1, 5 + 7 * (i % 5)))).cuda() for i in range(B)]
    seeds = [4100 + i for i in range(B)]
    kw = dict(gamma=gamma, top_k=20, top_p=0.9)
    if case == "seeded_g2":
        kw["random_seed"] = 42
    if case == "split_lm_head":
        os.environ["SD_GEMM_UNITS"] = "1536"
    try:
        probe = hip.S.speculative_sampling_batch(prompts[1:2], dm, tm, -1, None, 24, seeds=seeds[1:2], **kw)
        eos = int(probe[0][0, prompts[1].shape[1] + 5])
        runs = {}
        for mode in ("0", "1"):
            os.environ["SD_BATCH_FUSED_TAIL"] = mode
            runs[mode] = hip.S.speculative_sampling_batch(prompts, dm, tm, eos, None, 24, details=True, seeds=seeds, **kw)
    finally:
        os.environ.pop("SD_BATCH_FUSED_TAIL", None)
        os.environ.pop("SD_GEMM_UNITS", None)
    (o0, d0), (o1, d1) = runs["0"], runs["1"]
    stopped = 0
    for a, b, da, db, p in zip(o0, o1, d0, d1, prompts):
        assert torch.equal(a, b), (a, b)
        assert da["acc_len"] == db["acc_len"] and da["target_call_times"] == db["target_call_times"]
        assert float(da["acc_rate"]) == float(db["acc_rate"])
        stopped += a.shape[1] < p.shape[1] + 24
    assert stopped >= 1
    if case.startswith("all_accept"):
        assert all(x == gamma for d in d1 for x in d["acc_len"])
    if case.startswith("perturbed"):
        tot = sum(sum(d["acc_len"]) for d in d1)
        assert 0 < tot < gamma * sum(len(d["acc_len"]) for d in d1)


# --------------------------------------------------------------------------- headline shapes vs the oracle
def _host_sd(m):
    sd = {n: m._synth_get(n).cpu() for n in m._synth_names}
    return sd


def test_headline_shapes_bf16_error_against_fp32_truth(hip):
    """A 2-layer model with Llama-2-13b's layer shape (hidden 5120, 40 heads x 128, inter 13824, vocab 32000), bf16:
    the HIP forward's error against an fp32 forward of the same (bf16-valued) weights, next to the error of the
    reference's own bf16 arithmetic (oracle in bf16, which rounds where the reference rounds).  Bar: HIP error <= 1.5x
    the reference-bf16 error (+ a small absolute floor), on the prefill rows and on a gamma+1-row verify."""
    cfg = ModelConfig(arch="llama", vocab_size=32000, hidden_size=5120, intermediate_size=13824, num_hidden_layers=2,
                      num_attention_heads=40, num_key_value_heads=40, max_position_embeddings=256, rms_norm_eps=1e-5)
    m = hip.engine.SpecDecModel.synthetic(cfg, seed=9, dtype=torch.bfloat16, max_pos=64)
    sd16 = _host_sd(m)
    sd32 = {k: v.float() for k, v in sd16.items()}
    ids = torch.from_numpy(np.random.default_rng(4).integers(3, cfg.vocab_size, size=(1, 29)))
    o16, o32 = oracle.RefCausalLM(cfg, sd16), oracle.RefCausalLM(cfg, sd32)
    ses = m.new_session(64)
    past16 = past32 = None
    pos = 0
    for q in (24, 5):
        chunk = ids[:, pos:pos + q]
        r16 = o16(chunk, past_key_values=past16)
        r32 = o32(chunk, past_key_values=past32)
        past16, past32 = r16.past_key_values, r32.past_key_values
        nl = min(q, 5)
        got = ses.forward(chunk[0].to(torch.int32).cuda(), nl).cpu()
        truth = r32.logits[0, -nl:]
        e_hip = float((got - truth).abs().max())
        e_ref = float((r16.logits.float()[0, -nl:] - truth).abs().max())
        rms_hip = float((got - truth).pow(2).mean().sqrt())
        rms_ref = float((r16.logits.float()[0, -nl:] - truth).pow(2).mean().sqrt())
        print(f"rows {q}: max err hip {e_hip:.4f} ref-bf16 {e_ref:.4f}; rms hip {rms_hip:.5f} ref-bf16 {rms_ref:.5f}; "
              f"|logit| max {float(truth.abs().max()):.2f}")
        assert rms_hip <= 1.5 * rms_ref + 1e-3, (q, rms_hip, rms_ref)
        assert e_hip <= 1.5 * e_ref + 0.02, (q, e_hip, e_ref)
        pos += q


def test_headline_pair_first_verify_vs_oracle_full_size(hip):
    """BASELINE configs[1] at its real shapes: llama-68m -> Llama-2-13b, bf16, the same synthetic weights on the GPU and
    in the oracle (torch-CPU bf16).  A 24-token prompt, 4 drafted tokens: the HIP target's logits for the gamma+1 verify
    rows and the p_hist rows norm_probs makes of them (T=1, k=20, p=0.9) against oracle.RefCausalLM + oracle.norm_logits;
    the draft's prefill + one step likewise.  bf16 through 40 layers: logits within 4 % of the logit scale, total
    variation of every probability row <= 0.15 (measured 0.03 - 0.11), and the top-1 token agrees wherever the oracle's margin is clear."""
    dcfg, tcfg = load_config("llama-68m"), load_config("llama-2-13b")
    dm = hip.engine.SpecDecModel.synthetic(dcfg, seed=1, dtype=torch.bfloat16, max_pos=64)
    tm = hip.engine.SpecDecModel.synthetic(tcfg, seed=2, dtype=torch.bfloat16, max_pos=64)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ids = torch.from_numpy(np.random.default_rng(8).integers(3, tcfg.vocab_size, size=(1, 28)))
    for m, cfg, name in ((tm, tcfg, "target"), (dm, dcfg, "draft")):
        om = oracle.RefCausalLM(cfg, _host_sd(m))
        ses = m.new_session(64)
        r = om(ids[:, :23])
        ses.forward(ids[0, :23].to(torch.int32).cuda(), 0)
        r2 = om(ids[:, 23:28], past_key_values=r.past_key_values)
        got = ses.forward(ids[0, 23:28].to(torch.int32).cuda(), 5)
        want = r2.logits.float()[0]
        scale = float(want.abs().max())
        err = float((got.cpu() - want).abs().max())
        print(f"{name}: verify-row logits max err {err:.4f} of scale {scale:.2f}")
        assert err <= 0.04 * scale, (name, err, scale)
        p_hip = hip.S.norm_logits(got, 1.0, 20, 0.9).cpu()
        for i in range(5):
            p_ref = oracle.norm_logits(want[i:i + 1], 1.0, 20, 0.9)[0]
            tv = 0.5 * float((p_hip[i] - p_ref).abs().sum())
            assert tv <= 0.15, (name, i, tv)
            top2 = torch.topk(want[i], 2).values
            if float(top2[0] - top2[1]) > 4 * err:
                assert int(p_hip[i].argmax()) == int(p_ref.argmax())
        del om


def test_acceptance_dial_pair_accept_length_vs_oracle(hip):
    """bench.py's acceptance dial (synth.py): on a reduced pair with the dial construction (draft hidden 256 inside a
    target of hidden 512, 3 layers) the mean accept length falls monotonically with sigma, sigma = 0 is (nearly)
    all-accept, and the oracle replaying the device's Philox stream on the same weights gives the same accept
    lengths statistically (bf16 rounding may flip individual tokens)."""
    from llmspeculativesampling_amd.synth import dial_draft_transform, dial_target_transform
    dcfg = ModelConfig(arch="llama", vocab_size=4096, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                       num_attention_heads=4, num_key_value_heads=4, max_position_embeddings=256, rms_norm_eps=1e-5)
    tcfg = ModelConfig(arch="llama", vocab_size=4096, hidden_size=512, intermediate_size=1024, num_hidden_layers=3,
                       num_attention_heads=4, num_key_value_heads=4, max_position_embeddings=256, rms_norm_eps=1e-5)
    base = hip.engine.SpecDecModel.synthetic(dcfg, seed=11, dtype=torch.bfloat16, transform=dial_draft_transform(0.0, 11))
    tgt = hip.engine.SpecDecModel.synthetic(tcfg, seed=12, dtype=torch.bfloat16,
                                            transform=dial_target_transform(base._synth_get, 256, 512))
    prompt = torch.from_numpy(np.random.default_rng(2).integers(3, 4096, size=(1, 20)))
    means = []
    for sg in (0.0, 0.1, 0.3, 1.0):
        drf = base if sg == 0.0 else hip.engine.SpecDecModel.synthetic(dcfg, seed=11, dtype=torch.bfloat16,
                                                                       transform=dial_draft_transform(sg, 11))
        out, d = hip.S.speculative_sampling(prompt.cuda(), drf, tgt, -1, None, 60, gamma=4, top_k=20, top_p=0.9, details=True,
                                            rng=hip.noise.DeviceNoise(55))
        means.append(float(np.mean(d["acc_len"])))
        if sg == 0.1:
            want, wd = oracle.speculative_sampling(prompt, oracle.RefCausalLM(dcfg, _host_sd(drf)),
                                                   oracle.RefCausalLM(tcfg, _host_sd(tgt)), -1, None, 60, gamma=4, top_k=20,
                                                   top_p=0.9, details=True, noise=PhiloxOracleNoise(hip.lib, 55, 4, _st))
            assert abs(float(np.mean(wd["acc_len"])) - means[-1]) <= 0.6, (wd["acc_len"], d["acc_len"])
    print("mean accept length per sigma:", means)
    assert means[0] >= 3.0 and means[-1] <= 1.0             # (the 68m -> 13b pair of bench.py reaches 3.76 at sigma = 0)
    assert all(means[i] >= means[i + 1] - 0.3 for i in range(len(means) - 1))


# --------------------------------------------------------------------------- drop-in boundary
def _hf_pair():
    import transformers
    lc = transformers.LlamaConfig(vocab_size=512, hidden_size=64, num_hidden_layers=2, intermediate_size=128,
                                  num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=128,
                                  rms_norm_eps=1e-5, tie_word_embeddings=False)
    oc = transformers.OPTConfig(vocab_size=512, hidden_size=64, num_hidden_layers=2, ffn_dim=128, num_attention_heads=4,
                                max_position_embeddings=128, do_layer_norm_before=True, word_embed_proj_dim=64)
    torch.manual_seed(3)
    return transformers.LlamaForCausalLM(lc).eval(), transformers.OPTForCausalLM(oc).eval()


@pytest.mark.parametrize("arch", ["llama", "opt"])
def test_hf_module_as_model_matches_state_dict_path_and_the_module_itself(hip, arch):
    """INTEGRATION.md's route: a transformers LlamaForCausalLM / OPTForCausalLM (random-init, as evaluation.py:183-253
    would hand over a loaded one) passed straight to speculative_sampling.  as_specdec_model / from_hf must give the ids
    of the from_state_dict path, the conversion must be cached per module, and the engine's logits must match the
    module's own forward (stock transformers, fp32, 1e-3)."""
    from llmspeculativesampling_amd.config import config_from_hf
    llama, opt = _hf_pair()
    mod = llama if arch == "llama" else opt
    cfg = config_from_hf(mod.config)
    sd = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    if arch == "opt":
        sd.setdefault("lm_head.weight", sd["model.decoder.embed_tokens.weight"])
    prompt = torch.from_numpy(np.random.default_rng(1).integers(3, 512, size=(1, 12)))
    torch.manual_seed(5)
    a = hip.S.speculative_sampling(prompt.cuda(), mod, mod, 2, None, 12, gamma=3, top_k=10, top_p=0.9)
    m2 = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=torch.float32)
    torch.manual_seed(5)
    b = hip.S.speculative_sampling(prompt.cuda(), m2, m2, 2, None, 12, gamma=3, top_k=10, top_p=0.9)
    assert torch.equal(a, b)
    assert hip.engine.as_specdec_model(mod) is hip.engine.as_specdec_model(mod)
    with torch.no_grad():
        want = mod(prompt).logits[0].float()
    ses = hip.engine.as_specdec_model(mod).new_session(32)
    got = ses.forward(prompt[0].to(torch.int32).cuda(), 12).cpu()
    assert float((got - want).abs().max()) <= 1e-3
    # the oracle on the module's weights agrees on the tokens under the same host seed (full drop-in chain)
    torch.manual_seed(5)
    om = oracle.RefCausalLM(cfg, sd)
    c = oracle.speculative_sampling(prompt, om, om, 2, None, 12, gamma=3, top_k=10, top_p=0.9)
    np.testing.assert_array_equal(a.cpu().numpy(), c.numpy())


def test_kvcache_model_multi_iid_and_rollback_choice_match_oracle(hip):
    """KVCacheModel.generate(x, gamma, multi=w, strategy="iid") and rollback(end, choice) on the drop-in class
    (reference kvcache_model.py:273-276, 180-200, 239-244, 390-396, 433-436): tokens of every replica, the (w, S, V)
    probability history and the (w, H, S, D) cache shape against the oracle wrapper fed the same noise."""
    cfg = load_config("tiny-llama-target")
    sd = make_state_dict(cfg, 12)
    V = cfg.vocab_size
    prompt = torch.from_numpy(np.random.default_rng(7).integers(3, V, size=(1, 9)))
    rec = oracle.RecordingNoise()
    torch.manual_seed(1)
    ok = oracle.RefKVCacheModel(oracle.RefCausalLM(cfg, sd), 1.0, 20, 0.9, rec)
    x1 = ok.generate(prompt, 2)                                   # batch 1: prefill + 2 tokens
    x3 = ok.generate(x1, 3, multi=3, strategy="iid")              # 3 replicas, 3 tokens each
    hist3 = ok._prob_history.clone()
    ok.rollback(x3.shape[1] - 2, choice=1)
    x4 = ok.generate(x3[1:2, :x3.shape[1] - 1], 2)                # continue replica 1 alone

    m = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=torch.float32)
    kv = hip.S.KVCacheModel(m, 1.0, 20, 0.9, noise=hip.noise.ReplayNoise(rec.events, "cuda"))
    y1 = kv.generate(prompt.cuda(), 2)
    np.testing.assert_array_equal(y1.cpu().numpy(), x1.numpy())
    y3 = kv.generate(y1, 3, multi=3, strategy="iid")
    np.testing.assert_array_equal(y3.cpu().numpy(), x3.numpy())
    ph = kv._prob_history
    assert tuple(ph.shape) == tuple(hist3.shape)
    np.testing.assert_allclose(ph.cpu().numpy(), hist3.numpy(), atol=1e-5)
    k0, v0 = kv._past_key_values[0]
    assert tuple(k0.shape) == (3, cfg.num_key_value_heads, x3.shape[1] - 1, cfg.head_dim)
    kv.rollback(x3.shape[1] - 2, choice=1)
    assert tuple(kv._prob_history.shape) == (1, x3.shape[1] - 2, V)
    assert tuple(kv._past_key_values[0][0].shape) == (1, cfg.num_key_value_heads, x3.shape[1] - 2, cfg.head_dim)
    y4 = kv.generate(y3[1:2, :y3.shape[1] - 1], 2)
    np.testing.assert_array_equal(y4.cpu().numpy(), x4.numpy())
    with pytest.raises(NotImplementedError):
        kv.generate(y4, 1, multi=2, strategy="beam")


def test_out_of_range_token_ids_raise_like_the_reference(hip, capsys):
    """nn.Embedding raises IndexError for an id outside the table (a tokenizer's added pad id is the usual case); the
    reference's speculative_sampling wraps it into RuntimeError('s') (speculative_sampling.py:2044-2046), its
    autoregressive_sampling lets it through.  The HIP gather itself must never read outside the table."""
    cfg = load_config("tiny-llama-target")
    m = hip.engine.SpecDecModel.from_state_dict(cfg, make_state_dict(cfg, 12), dtype=torch.float32)
    V = cfg.vocab_size
    bad = torch.tensor([[5, 6, V, 7]], device="cuda")
    neg = torch.tensor([[5, -1, 7]], device="cuda")
    for x in (bad, neg):
        with pytest.raises(RuntimeError, match="^s$"):
            hip.S.speculative_sampling(x, m, m, 2, None, 4)
        with pytest.raises(IndexError):
            hip.S.autoregressive_sampling(x, m, 4, 2)
        with pytest.raises(IndexError):
            hip.S.KVCacheModel(m)._forward_with_kvcache(x)
    capsys.readouterr()
    assert torch.equal(hip.S.speculative_sampling(bad, m, m, 2, None, 0), bad)       # max_len 0 never touches the model
    # straight through the C ABI: clamped, finite logits, no fault
    ses = m.new_session(16)
    out = ses.forward(torch.tensor([5, V + 1000, -7], dtype=torch.int32, device="cuda"), 3)
    assert bool(torch.isfinite(out).all())


def test_top_k_top_p_filter_in_place_and_support(hip):
    """reference utils.py:152-179: the argument is mutated and returned; k = 0, p = 0 leaves it untouched; the kept set is
    the filter's own decision (a kept logit whose probability underflows stays finite)."""
    x = logits_row(77, 1000, 3.0).cuda()
    keep = x.clone()
    y = hip.S.top_k_top_p_filter(x, 0, 0.0)
    assert y is x and torch.equal(x, keep)
    want = oracle.top_k_top_p_filter(keep.cpu(), 20, 0.9)
    y = hip.S.top_k_top_p_filter(x, 20, 0.9)
    assert y is x
    assert torch.equal(torch.isinf(x.cpu()), torch.isinf(want))
    assert torch.equal(x.cpu()[~torch.isinf(want)], want[~torch.isinf(want)])
    # underflow: second-largest logit is 200 below the top one -> probability 0 in fp32, still inside top-k = 3
    z = torch.full((1, 64), -500.0, device="cuda")
    z[0, 3], z[0, 9], z[0, 11] = 50.0, -150.0, -160.0
    w = hip.S.top_k_top_p_filter(z.clone(), 3, 0.0)
    assert sorted(torch.nonzero(torch.isfinite(w[0])).flatten().tolist()) == [3, 9, 11]
    ow = oracle.top_k_top_p_filter(z.cpu(), 3, 0.0)
    assert torch.equal(torch.isfinite(w.cpu()), torch.isfinite(ow))


# --------------------------------------------------------------------------- small-model decode path (small_kernels.h)
SMALL_CFGS = {
    "llama68m_like": dict(arch="llama", vocab_size=16384, hidden_size=768, intermediate_size=3072, num_hidden_layers=2,
                          num_attention_heads=12, num_key_value_heads=12, max_position_embeddings=256, rms_norm_eps=1e-6),
    "llama_gqa_h256": dict(arch="llama", vocab_size=16384, hidden_size=256, intermediate_size=704, num_hidden_layers=3,
                           num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=256, rms_norm_eps=1e-5),
    "opt125m_like": dict(arch="opt", vocab_size=16384, hidden_size=768, ffn_dim=3072, num_hidden_layers=2,
                         num_attention_heads=12, max_position_embeddings=256, do_layer_norm_before=True,
                         word_embed_proj_dim=768),
    # below 1024 n-tiles the per-op chain's head sums two k-slabs where the small path keeps the whole k-range
    "llama68m_v8192": dict(arch="llama", vocab_size=8192, hidden_size=768, intermediate_size=3072, num_hidden_layers=2,
                           num_attention_heads=12, num_key_value_heads=12, max_position_embeddings=256, rms_norm_eps=1e-6),
    "llama_gqa_h256_v4096": dict(arch="llama", vocab_size=4096, hidden_size=256, intermediate_size=704, num_hidden_layers=3,
                                 num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=256, rms_norm_eps=1e-5),
}


@pytest.mark.parametrize("name", list(SMALL_CFGS))
def test_small_model_path_is_bit_identical_to_the_launch_per_op_chain(hip, name):
    """The prologue-fused decode chain for small models (5 launches per layer; norm + residual recomputed inside the
    consuming GEMM) must give bit-identical logits and KV rows to the per-op chain (SD_SMALL_PATH=0) for 1..4 new rows,
    and both must agree with the oracle forward in bf16.  From 1024 n-tiles on (V = 16384) the per-op chain's head also
    keeps the whole k-range in one workgroup; the two cases below that state the documented one-ulp tolerance of the logits."""
    cfg = ModelConfig(**SMALL_CFGS[name])
    sd = make_state_dict(cfg, 90, dtype=torch.bfloat16)
    m = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=torch.bfloat16)
    ids = torch.from_numpy(np.random.default_rng(19).integers(3, cfg.vocab_size, size=(1, 41))).to(torch.int32).cuda()[0]
    outs = {}
    modes = {"per_op": {"SD_SMALL_PATH": "0", "SD_FUSE_EMBED_QKV": "0"},      # one launch per op (the round-1 chain)
             "default": {"SD_SMALL_PATH": "0"},                                # (round 3's default) embedding + norm fused into QKV(0)
             "1": {"SD_SMALL_PATH": "2"},                                     # every norm fused into its consumer
             "hybrid": {}}                                                    # (the default) the layers' norms as QKV / gate-up prologues;
                                                                               # O, down and the head on the per-op kernels
    for flag, env in modes.items():
        os.environ.update(env)
        try:
            ses = m.new_session(64)
            ses.forward(ids[:30], 0)                              # prompt (many rows: the per-op chain either way)
            got, pos = [], 30
            for q in (1, 2, 1, 4, 3):
                got.append(ses.forward(ids[pos:pos + q], q).clone())
                pos += q
            outs[flag] = (torch.cat(got), ses.kv[:, :, :, :pos].clone())
        finally:
            for k in env:
                os.environ.pop(k, None)
    outs["0"] = outs["per_op"]
    # the DEFAULT route (per-op chain + embedding / first norm fused into the layer-0 QKV GEMM) is bit-identical at any V
    assert torch.equal(outs["default"][0], outs["0"][0]) and torch.equal(outs["default"][1], outs["0"][1])
    # SD_SMALL_PATH=2 (every seam a prologue): every K / V row is bit-identical; so are the logits once the per-op head keeps the
    # whole k-range too (>= 1024 n-tiles).  Below that the two heads add the same products in a different fp32 order, and a
    # logit may land on the other side of a bf16 rounding boundary: never by more than one bf16 ulp, and no NaN (the
    # hidden-256 case has waves with an empty k-range: DESIGN.md section 7, "uninitialised MFMA operand")
    assert not bool(torch.isnan(outs["1"][0]).any()) and not bool(torch.isnan(outs["1"][1].float()).any())
    assert torch.equal(outs["1"][1], outs["0"][1])
    # SD_SMALL_PATH=1 (round 4): the head is the per-op chain's own, so the logits are bit-identical at any vocabulary size
    assert torch.equal(outs["hybrid"][1], outs["0"][1]) and torch.equal(outs["hybrid"][0], outs["0"][0])
    if cfg.vocab_size >= 16384:
        assert torch.equal(outs["1"][0], outs["0"][0])
    else:
        a, b = outs["1"][0], outs["0"][0]
        assert bool(((a - b).abs() <= 2.0 ** -7 * torch.maximum(a.abs(), b.abs())).all())
        assert float((a != b).float().mean()) < 0.01
    om = oracle.RefCausalLM(cfg, sd)
    want = om(ids[None].long().cpu()).logits.float()[0, 30:41]
    got = outs["1"][0].cpu()
    assert float((got - want[: got.shape[0]]).abs().max()) <= 0.04 * float(want.abs().max())


# --------------------------------------------------------------------------- bounded in-launch waits of the fused launches
def _fused_status(hip, ses):
    import ctypes as C
    torch.cuda.synchronize()
    w = C.c_uint(0)
    assert hip.lib.sd_session_fused_status(ses.handle, C.byref(w)) == 0, hip.lib.sd_last_error().decode()
    return w.value


def test_fused_launch_wait_timeout_surfaces_as_norm_logits_error(hip):
    """VERDICT r3 item 8 / ADVICE r3: the in-launch waits of the fused attention + O launch and of the k-split down
    projection are bounded; when a bound is hit the launch must NOT return plausible numbers.  A test hook
    (sd_session_test_skew_wait) makes the fused launches of ONE forward expect an arrival that never comes: every wait of
    that forward runs into its 20 ms limit, the logits must be NaN in every row, the sampler must raise the reference's
    RuntimeError('norm logits error') (utils.py:186-188), the session's status word must name both kinds of wait - and
    the forward after that must be back in step (counters re-zeroed, same logits as an undisturbed session, no further
    status bit): one failed launch does not break the session."""
    cfg = ModelConfig(arch="llama", vocab_size=4096, hidden_size=1024, intermediate_size=2048, num_hidden_layers=2,
                      num_attention_heads=8, num_key_value_heads=8, max_position_embeddings=128, rms_norm_eps=1e-5)
    m = hip.engine.SpecDecModel.synthetic(cfg, seed=5, dtype=torch.bfloat16)
    ids = torch.from_numpy(np.random.default_rng(3).integers(3, cfg.vocab_size, size=(1, 40))).to(torch.int32).cuda()[0]
    ref = m.new_session(64)
    ref.forward(ids[:20], 0)
    want5 = ref.forward(ids[20:25], 5).clone()
    want3 = ref.forward(ids[25:28], 3).clone()
    ses = m.new_session(64)
    ses.forward(ids[:20], 0)
    ses.profile(True)                                             # the 5-row forward takes the fused launches ...
    got = ses.forward(ids[20:25], 5).clone()
    prof = ses.profile_read()
    ses.profile(False)
    assert torch.equal(got, want5) and prof["norm_residual"][1] == 1 and _fused_status(hip, ses) == 0
    ses.rollback(20)
    assert hip.lib.sd_session_test_skew_wait(ses.handle, 1) == 0
    t0 = time.time()
    bad = ses.forward(ids[20:25], 5).clone()
    torch.cuda.synchronize()
    assert 0.02 <= time.time() - t0 < 5.0                          # 2 layers x 2 bounded waits of 20 ms, not a hang
    assert bool(torch.isnan(bad).any(dim=1).all()), "a timed-out fused launch returned finite logit rows"
    with pytest.raises(RuntimeError, match="norm logits error"):
        hip.S.norm_logits(bad[:1].clone(), 1.0, 20, 0.9)
    assert _fused_status(hip, ses) == 3                            # bit 0: attention + O, bit 1: down projection
    assert _fused_status(hip, ses) == 0                            # (read clears)
    ses.rollback(20)
    again = ses.forward(ids[20:25], 5).clone()                     # re-synchronised: the very next forward is clean
    assert torch.equal(again, want5)
    assert torch.equal(ses.forward(ids[25:28], 3), want3)
    assert _fused_status(hip, ses) == 0


# --------------------------------------------------------------------------- config 3: OPT in bf16 end to end
def test_opt_bf16_pair_end_to_end_vs_oracle(hip):
    """BASELINE config 3's family (OPT draft -> OPT target, bf16): OPT keeps its logits in the weight dtype, so the
    probability histories, the accept ratios, max_fn(p - q) and every draw run in bf16 (SD_NORM_DT_BF16).  Same outer
    seed into the oracle's bf16 CPU run and the HIP run (live host generator, which also consumes bf16 noise rows): the
    two bf16 forwards round in a different order, so an occasional token may flip; the bar is a long identical prefix
    and matching acceptance statistics.  Also: native device-RNG loop == Python-orchestrated loop bit for bit."""
    dcfg = ModelConfig(arch="opt", vocab_size=8192, hidden_size=128, ffn_dim=512, num_hidden_layers=2, num_attention_heads=2,
                       max_position_embeddings=512, do_layer_norm_before=True, word_embed_proj_dim=128)
    tcfg = ModelConfig(arch="opt", vocab_size=8192, hidden_size=256, ffn_dim=1024, num_hidden_layers=3, num_attention_heads=4,
                       max_position_embeddings=512, do_layer_norm_before=True, word_embed_proj_dim=256)
    dsd = make_state_dict(dcfg, 41, dtype=torch.bfloat16)
    tsd = make_state_dict(tcfg, 42, dtype=torch.bfloat16)
    # correlate the pair: the target's embedding / head start from the draft's (padded), so accepts really occur
    with torch.no_grad():
        tsd["model.decoder.embed_tokens.weight"][:, :128] = dsd["model.decoder.embed_tokens.weight"]
        tsd["lm_head.weight"] = tsd["model.decoder.embed_tokens.weight"]
    prompt = torch.from_numpy(np.random.default_rng(3).integers(3, 8192, size=(1, 20)))
    torch.manual_seed(21)
    want, wd = oracle.speculative_sampling(prompt, oracle.RefCausalLM(dcfg, dsd), oracle.RefCausalLM(tcfg, tsd), 2, None, 32,
                                           gamma=4, top_k=20, top_p=0.9, details=True)
    dm = hip.engine.SpecDecModel.from_state_dict(dcfg, dsd, dtype=torch.bfloat16)
    tm = hip.engine.SpecDecModel.from_state_dict(tcfg, tsd, dtype=torch.bfloat16)
    assert dm.norm_mode == hip.L.SD_NORM_DT_BF16 and dm.probs_dtype == torch.bfloat16
    torch.manual_seed(21)
    got, gd = hip.S.speculative_sampling(prompt.cuda(), dm, tm, 2, None, 32, gamma=4, top_k=20, top_p=0.9, details=True)
    w, g = want[0].tolist(), got[0].cpu().tolist()
    common = next((i for i, (a, b) in enumerate(zip(w, g)) if a != b), min(len(w), len(g)))
    print("OPT bf16: identical prefix", common - 20, "generated tokens; acc_len oracle", wd["acc_len"], "hip", gd["acc_len"])
    assert common >= 20 + 6, (common, w, g)
    assert abs(float(np.mean(gd["acc_len"])) - float(np.mean(wd["acc_len"]))) <= 1.0
    kv = hip.S.KVCacheModel(tm, 1.0, 20, 0.9)
    q = kv._forward_with_kvcache(prompt.cuda())
    assert q.dtype == torch.bfloat16 and kv._prob_history.dtype == torch.bfloat16      # the reference's history dtype
    a, da = hip.S.speculative_sampling(prompt.cuda(), dm, tm, -1, None, 32, gamma=4, top_k=20, top_p=0.9, details=True,
                                       rng=hip.noise.DeviceNoise(9))
    b, db = hip.S.speculative_sampling(prompt.cuda(), dm, tm, -1, None, 32, gamma=4, top_k=20, top_p=0.9, details=True,
                                       rng=hip.noise.DeviceNoise(9), verbose=True)
    assert torch.equal(a, b) and da["acc_len"] == db["acc_len"]


# --------------------------------------------------------------------------- fp16 models (what the reference harness loads)
FP16_CFGS = {
    "llama_d64_gqa": dict(arch="llama", vocab_size=1024, hidden_size=256, intermediate_size=704, num_hidden_layers=2,
                          num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=512, rms_norm_eps=1e-6),
    "llama_d128": dict(arch="llama", vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                       num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=512, rms_norm_eps=1e-5),
    "opt_d64_pre": dict(arch="opt", vocab_size=1024, hidden_size=128, ffn_dim=512, num_hidden_layers=2,
                        num_attention_heads=2, max_position_embeddings=512, do_layer_norm_before=True, word_embed_proj_dim=128),
    "opt_d32_post": dict(arch="opt", vocab_size=1024, hidden_size=128, ffn_dim=256, num_hidden_layers=2,
                         num_attention_heads=4, max_position_embeddings=512, do_layer_norm_before=False, word_embed_proj_dim=64),
}


@pytest.mark.parametrize("name", list(FP16_CFGS))
def test_fp16_forward_vs_oracle(hip, name):
    """SD_F16: fp16 weights / activations / KV (the dtype evaluation.py:185 loads) through the same kernels as bf16
    (v_mfma_f32_16x16x32_f16, fp32 accumulation, one rounding per op where the reference materialises a tensor) against
    the oracle forward in fp16: prefill of 150 rows in chunks, then steps of 1, 2, 5 and 9 rows; KV rows likewise.
    fp16 has 3 more mantissa bits than bf16: the bound is 8x tighter than the bf16 one."""
    cfg = ModelConfig(**FP16_CFGS[name])
    sd = make_state_dict(cfg, 77, dtype=torch.float16, gain=0.5)
    om = oracle.RefCausalLM(cfg, sd)
    m = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=torch.float16)
    ses = m.new_session(256)
    assert ses.kv.dtype == torch.float16
    ids = torch.from_numpy(np.random.default_rng(9).integers(3, cfg.vocab_size, size=(1, 167)))
    past, pos = None, 0
    for q in (150, 1, 2, 5, 9):
        chunk = ids[:, pos:pos + q]
        o = om(chunk, past_key_values=past)
        past = o.past_key_values
        nl = min(q, 9)
        got = ses.forward(chunk[0].to(torch.int32).cuda(), nl).cpu()
        want = o.logits.float()[0, -nl:]
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 0.005 * scale + 2e-3, (name, q, float((got - want).abs().max()), scale)
        pos += q
    k, v = ses.past_key_values()[1]
    ok, ov = past[1]
    assert float((k.float().cpu() - ok.float()).abs().max()) <= 0.01 * max(1.0, float(ok.float().abs().max()))
    assert float((v.float().cpu() - ov.float()).abs().max()) <= 0.01 * max(1.0, float(ov.float().abs().max()))


def test_fp16_speculative_pair_and_hf_module(hip):
    """fp16 end to end: a Llama pair (fp32 probability rows from fp16-rounded logits, SD_NORM_ROUND_F16) and an OPT pair
    (fp16 probability rows, SD_NORM_DT_F16) against the oracle in fp16 under the same host seed; and a transformers
    module in fp16 keeps its dtype through from_hf (it used to be converted to bf16)."""
    import transformers
    for arch in ("llama", "opt"):
        cfg = ModelConfig(**FP16_CFGS["llama_d64_gqa" if arch == "llama" else "opt_d64_pre"])
        dsd = make_state_dict(cfg, 5, dtype=torch.float16, gain=0.5)
        tsd = {k: v.to(torch.float16) for k, v in perturb_state_dict({a: b.float() for a, b in dsd.items()}, 6, 0.05).items()}
        prompt = torch.from_numpy(np.random.default_rng(3).integers(3, cfg.vocab_size, size=(1, 20)))
        torch.manual_seed(31)
        want, wd = oracle.speculative_sampling(prompt, oracle.RefCausalLM(cfg, dsd), oracle.RefCausalLM(cfg, tsd), 2, None, 32,
                                               gamma=4, top_k=20, top_p=0.9, details=True)
        dm = hip.engine.SpecDecModel.from_state_dict(cfg, dsd, dtype=torch.float16)
        tm = hip.engine.SpecDecModel.from_state_dict(cfg, tsd, dtype=torch.float16)
        assert dm.norm_mode == (hip.L.SD_NORM_DT_F16 if arch == "opt" else 0)
        torch.manual_seed(31)
        got, gd = hip.S.speculative_sampling(prompt.cuda(), dm, tm, 2, None, 32, gamma=4, top_k=20, top_p=0.9, details=True)
        w, g = want[0].tolist(), got[0].cpu().tolist()
        common = next((i for i, (a, b) in enumerate(zip(w, g)) if a != b), min(len(w), len(g)))
        print(arch, "fp16: identical prefix", common - 20, "of", len(w) - 20, "generated tokens; acc_len", wd["acc_len"], gd["acc_len"])
        assert common >= 20 + 8, (arch, common, w, g)
        a, da = hip.S.speculative_sampling(prompt.cuda(), dm, tm, -1, None, 24, gamma=4, top_k=20, top_p=0.9, details=True,
                                           rng=hip.noise.DeviceNoise(9))
        b, db = hip.S.speculative_sampling(prompt.cuda(), dm, tm, -1, None, 24, gamma=4, top_k=20, top_p=0.9, details=True,
                                           rng=hip.noise.DeviceNoise(9), verbose=True)
        assert torch.equal(a, b) and da["acc_len"] == db["acc_len"]
    lc = transformers.LlamaConfig(vocab_size=512, hidden_size=64, num_hidden_layers=2, intermediate_size=128, num_attention_heads=4,
                                  num_key_value_heads=2, max_position_embeddings=128, rms_norm_eps=1e-5, tie_word_embeddings=False)
    torch.manual_seed(3)
    mod = transformers.LlamaForCausalLM(lc).eval().half()
    sm = hip.engine.as_specdec_model(mod)
    assert sm.dtype == torch.float16 and sm.new_session(16).kv.dtype == torch.float16


# --------------------------------------------------------------------------- config 5: tensor-parallel target, fp8 KV
TP_CFG = dict(arch="llama", vocab_size=1024, hidden_size=256, intermediate_size=704, num_hidden_layers=3,
              num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=256, rms_norm_eps=1e-5)


def _run_ranks(fns):
    """Run one callable per tensor-parallel rank, each on its own host thread and HIP stream (the loopback group's
    all-reduce rendezvous needs all ranks in flight at once)."""
    import threading
    out, errs = [None] * len(fns), []

    def work(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                out[r] = fns[r]()
                torch.cuda.current_stream().synchronize()
        except Exception as e:                                   # noqa: BLE001
            errs.append((r, repr(e)))
    ts = [threading.Thread(target=work, args=(r,)) for r in range(len(fns))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errs and all(not t.is_alive() for t in ts), errs
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_tensor_parallel_shards_vs_unsharded_forward(hip, dtype):
    """Two Megatron shards of a GQA Llama (2 of 4 query heads, 1 of 2 KV heads, half of the MLP each) on one GPU through
    the loopback group (same kernels and the same tp_reduce as the RCCL path; the all-reduce is an in-process
    rendezvous): logits of both ranks are bit-identical to each other and equal the unsharded engine / the oracle within
    the dtype's bound; each rank's KV arena holds its own KV head."""
    from llmspeculativesampling_amd import tp
    cfg = ModelConfig(**TP_CFG)
    sd = make_state_dict(cfg, 70, dtype=dtype)
    full = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=dtype)
    groups = tp.TPGroup.loopback(2)
    shards = [tp.shard_model(cfg, sd, r, 2, group=groups[r], dtype=dtype) for r in range(2)]
    ids = torch.from_numpy(np.random.default_rng(4).integers(3, cfg.vocab_size, size=(1, 40))).to(torch.int32).cuda()[0]
    ref_ses = full.new_session(64)
    sess = [m.new_session(64) for m in shards]
    om = oracle.RefCausalLM(cfg, sd)
    past, pos = None, 0
    for q in (30, 1, 5, 4):
        want_full = ref_ses.forward(ids[pos:pos + q], min(q, 5)).clone()
        got = _run_ranks([lambda r=r: sess[r].forward(ids[pos:pos + q], min(q, 5)).clone() for r in range(2)])
        o = om(ids[None, pos:pos + q].long().cpu(), past_key_values=past)
        past = o.past_key_values
        want = o.logits.float()[0, -min(q, 5):]
        assert torch.equal(got[0], got[1])                       # every rank sees the same all-reduced sums
        scale = float(want.abs().max())
        tol = 1e-3 if dtype == torch.float32 else 0.04 * scale
        assert float((got[0].cpu() - want).abs().max()) <= tol, (q, float((got[0].cpu() - want).abs().max()))
        assert float((got[0] - want_full).abs().max()) <= tol
        pos += q
    for r in range(2):                                            # rank r's arena = KV head r of the full model
        assert tuple(sess[r].kv.shape) == (3, 2, 1, 64, 64)
        ktol = 1e-4 if dtype == torch.float32 else 0.05
        ref = ref_ses.kv[:, :, r:r + 1, :pos].float()
        assert float((sess[r].kv[:, :, :, :pos].float() - ref).abs().max()) <= ktol * max(1.0, float(ref.abs().max()))


def test_tensor_parallel_target_in_speculative_sampling(hip):
    """The whole decode loop with a 2-way tensor-parallel target (fp32, loopback group): every rank runs the same
    speculative_sampling call on its shard with the same device-RNG seed and ends with the tokens of the unsharded run."""
    from llmspeculativesampling_amd import tp
    cfg = ModelConfig(**TP_CFG)
    dcfg = load_config("tiny-llama-draft") if False else cfg
    dsd = make_state_dict(cfg, 70)
    tsd = perturb_state_dict(dsd, 71, 0.1)
    draft = hip.engine.SpecDecModel.from_state_dict(dcfg, dsd, dtype=torch.float32)
    full = hip.engine.SpecDecModel.from_state_dict(cfg, tsd, dtype=torch.float32)
    prompt = torch.from_numpy(np.random.default_rng(8).integers(3, cfg.vocab_size, size=(1, 12))).cuda()
    kw = dict(gamma=4, top_k=20, top_p=0.9)
    want, wd = hip.S.speculative_sampling(prompt, draft, full, -1, None, 24, details=True, rng=hip.noise.DeviceNoise(31), **kw)
    groups = tp.TPGroup.loopback(2)
    shards = [tp.shard_model(cfg, tsd, r, 2, group=groups[r], dtype=torch.float32) for r in range(2)]
    drafts = [hip.engine.SpecDecModel.from_state_dict(dcfg, dsd, dtype=torch.float32) for _ in range(2)]
    res = _run_ranks([lambda r=r: hip.S.speculative_sampling(prompt, drafts[r], shards[r], -1, None, 24, details=True,
                                                              rng=hip.noise.DeviceNoise(31), **kw) for r in range(2)])
    for out, d in res:
        assert torch.equal(out, want) and d["acc_len"] == wd["acc_len"]
    assert 0 < sum(wd["acc_len"]) < 4 * len(wd["acc_len"])


@pytest.mark.parametrize("name", ["llama_d128", "llama_d64_gqa"])
def test_fp8_kv_arena_vs_16bit_arena(hip, name):
    """fp8 (OCP e4m3) KV arena: K / V rows quantised where they are appended, widened in the attention kernel.  Same bf16
    model with a bf16 arena and with an fp8 arena over a 150-token prompt + verify-sized steps: the arena is half the
    bytes, its rows equal the bf16 ones within e4m3's 2^-4 relative step, and the logits stay within 6 % of the logit
    scale (measured ~2 %) of the bf16-KV run and of the oracle."""
    from test_gpu_parity import MID_CFGS
    cfg = ModelConfig(**MID_CFGS[name])
    sd = make_state_dict(cfg, 77, dtype=torch.bfloat16)
    m = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=torch.bfloat16)
    a, b = m.new_session(256), m.new_session(256, kv_dtype="fp8")
    assert b.kv.dtype == torch.uint8 and b.kv.numel() * 2 == a.kv.numel() * a.kv.element_size()
    ids = torch.from_numpy(np.random.default_rng(9).integers(3, cfg.vocab_size, size=(1, 167))).to(torch.int32).cuda()[0]
    om = oracle.RefCausalLM(cfg, sd)
    past, pos = None, 0
    worst = 0.0
    for q in (150, 1, 5, 9):
        la = a.forward(ids[pos:pos + q], min(q, 9)).clone()
        lb = b.forward(ids[pos:pos + q], min(q, 9)).clone()
        o = om(ids[None, pos:pos + q].long().cpu(), past_key_values=past)
        past = o.past_key_values
        want = o.logits.float()[0, -min(q, 9):]
        scale = float(want.abs().max())
        worst = max(worst, float((lb - la).abs().max()) / scale)
        assert float((lb - la).abs().max()) <= 0.06 * scale, (q, float((lb - la).abs().max()), scale)
        assert float((lb.cpu() - want).abs().max()) <= 0.08 * scale
        pos += q
    print(f"{name}: fp8-KV logits within {worst:.3%} of the logit scale of the bf16-KV run")
    # layer 0's K / V depend only on the embeddings: there the two arenas hold the same values up to e4m3's rounding
    # (3 mantissa bits: at most 2^-4 relative; deeper layers also see the quantised attention of the layers below)
    for x16, x8 in zip(a.past_key_values()[0], b.past_key_values()[0]):
        err = (x8.float() - x16.float()).abs()
        assert bool((err <= 0.0625 * x16.float().abs() + 2e-3).all())


def test_rccl_group_of_one_runs_the_all_reduce_path(hip):
    """The RCCL side of tp_reduce on the one-GPU box: a communicator of world size 1 created from a unique id through the
    C ABI (RCCL resolved with dlopen), kept on the session with SD_TP_FORCE=1, so every O / down projection goes through
    fold + ncclAllReduce (an identity here): logits must equal the plain forward's."""
    from llmspeculativesampling_amd import tp
    cfg = ModelConfig(**TP_CFG)
    sd = make_state_dict(cfg, 70, dtype=torch.bfloat16)
    m = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=torch.bfloat16)
    ids = torch.from_numpy(np.random.default_rng(4).integers(3, cfg.vocab_size, size=(1, 24))).to(torch.int32).cuda()[0]
    want = m.new_session(32).forward(ids, 5).clone()
    grp = tp.TPGroup.rccl(0, 1, lambda b: b)
    os.environ["SD_TP_FORCE"] = "1"
    try:
        ses = m.new_session(32)
        grp.bind(ses)
        got = ses.forward(ids, 5).clone()
    finally:
        os.environ.pop("SD_TP_FORCE", None)
    torch.cuda.synchronize()
    assert float((got - want).abs().max()) <= 0.04 * float(want.abs().max())


def test_rccl_token_gather_of_one_rank(hip):
    """sd_comm_* (SURVEY.md 8(b)): the throughput-mode gather as libspecdec's own ncclAllGather.  On the one-GPU box the
    communicator has world size 1 (the id comes from sd_comm_unique_id, as rank 0 would broadcast it): the gathered
    [world][rows][width] block must be the rank's own packed rows, and dist.TokenComm / pack_outputs agree with it.  The
    N > 1 algebra of gather_streams is covered over gloo in tests/test_host_cpu.py."""
    import ctypes as C
    from llmspeculativesampling_amd import dist as D
    ident = (C.c_char * 128)()
    assert hip.lib.sd_comm_unique_id(ident) == 0, hip.lib.sd_last_error()
    h = C.c_void_p()
    assert hip.lib.sd_comm_init(0, 1, ident, C.byref(h)) == 0, hip.lib.sd_last_error()
    r, w = C.c_int(-1), C.c_int(-1)
    assert hip.lib.sd_comm_rank(h, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (0, 1)
    outs = [torch.arange(5 + 3 * i, dtype=torch.int64).unsqueeze(0) + 100 * i for i in range(3)]
    mine = D.pack_outputs(outs, 40, "cuda")
    got = torch.full((1, 3, 40), -7, dtype=torch.int32, device="cuda")
    assert hip.lib.sd_comm_all_gather_tokens(h, mine.data_ptr(), got.data_ptr(), 3, 40, _st()) == 0, hip.lib.sd_last_error()
    torch.cuda.synchronize()
    assert torch.equal(got[0], mine)
    assert hip.lib.sd_comm_destroy(h) == 0


# --------------------------------------------------------------------------- tree attention
from golden_io import load as _load_golden                          # noqa: E402
G9_META, G9 = _load_golden("g9_tree")


@pytest.mark.parametrize("case", G9_META["tree"], ids=[c["id"] for c in G9_META["tree"]])
def test_tree_attention_forward_and_rollback_golden(hip, case):
    """KVCacheModel.forward_tree_attention / rollback_tree_attention through the HIP engine against the values the
    reference produced with its own model classes (G9): one tree forward over 9 nodes (tree-mask attention: every node
    sees the prefix and its ancestors; RoPE / learned position = depth; K/V appended at consecutive arena slots),
    probabilities of all gathered nodes, then the KV gather-compaction of the accepted path and a second round on the
    compacted cache.  fp32: probabilities within 1e-5, cache rows within 1e-4."""
    key = case["id"]
    cfg = load_config(case["cfg"])
    sd = make_state_dict(cfg, case["seed"])
    m = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=torch.float32)
    P = case["P"]
    prompt = torch.from_numpy(G9[key + "_prompt"]).cuda()
    tok, beam = torch.from_numpy(G9[key + "_tok"]), torch.from_numpy(G9[key + "_beam"])
    ai = [torch.zeros(tok.shape[1], dtype=torch.long) for _ in range(tok.shape[0])]
    seq, mask, pos, pids = hip.S.get_seq_att_mask(1, ai, list(beam), list(tok), P, 0, device="cuda")
    kv = hip.S.KVCacheModel(m, 1, case["top_k"], case["top_p"])
    p1 = kv.forward_tree_attention(seq, prompt, mask, pids, pos.clone())
    want = torch.from_numpy(G9[key + "_p1"])
    assert torch.equal(p1.cpu() > 0, want > 0)
    np.testing.assert_allclose(p1.cpu().numpy(), want.numpy(), atol=1e-5)
    assert tuple(kv._prob_history.shape) == (1, P + seq.shape[1], cfg.vocab_size)
    kv.rollback_tree_attention(torch.tensor([0]), torch.from_numpy(G9[key + "_keep"]))
    k_last = kv._past_key_values[-1][0].cpu().numpy()
    np.testing.assert_allclose(k_last, G9[key + "_k_last"], atol=1e-4)
    np.testing.assert_allclose(kv._prob_history.cpu().numpy(), G9[key + "_hist"], atol=1e-5)
    prefix2 = torch.from_numpy(G9[key + "_prefix2"]).cuda()
    tok2, beam2 = torch.from_numpy(G9[key + "_tok2"]), torch.from_numpy(G9[key + "_beam2"])
    ai2 = [torch.zeros(tok2.shape[1], dtype=torch.long) for _ in range(tok2.shape[0])]
    seq2, mask2, pos2, pids2 = hip.S.get_seq_att_mask(1, ai2, list(beam2), list(tok2), prefix2.shape[1], 0, device="cuda")
    p2 = kv.forward_tree_attention(seq2, prefix2, mask2, pids2, pos2.clone())
    np.testing.assert_allclose(p2.cpu().numpy(), G9[key + "_p2"], atol=1e-5)


@pytest.mark.parametrize("dtype,kv_dtype", [(torch.bfloat16, None), (torch.bfloat16, "fp8"), (torch.float16, None)],
                         ids=["bf16", "bf16_fp8kv", "fp16"])
def test_tree_attention_wide_tree_16bit_vs_oracle(hip, dtype, kv_dtype):
    """A 40-node tree (5 levels x 8 beams, the shape of num_beams = 8, gamma = 4 plus the root level) on the MFMA
    attention path, GQA, D = 64, after a 150-token prefix, in bf16 / fp16 and with the fp8 KV arena: node probabilities
    against the oracle's tree forward in the same dtype (total variation <= 0.05 per node; 0.12 with fp8 KV)."""
    from test_gpu_parity import MID_CFGS
    cfg = ModelConfig(**MID_CFGS["llama_d64_gqa"])
    sd = make_state_dict(cfg, 77, dtype=dtype, gain=0.7)
    m = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=dtype)
    m.kv_dtype = kv_dtype
    rng = np.random.default_rng(12)
    P, W, LV = 150, 8, 5
    prompt = torch.from_numpy(rng.integers(3, cfg.vocab_size, size=(1, P)))
    ai = [torch.zeros(W, dtype=torch.long) for _ in range(LV)]
    ab = [torch.from_numpy(rng.integers(0, W, size=W)) for _ in range(LV)]
    at = [torch.from_numpy(rng.integers(3, cfg.vocab_size, size=W)) for _ in range(LV)]
    seq, mask, pos, pids = hip.S.get_seq_att_mask(1, ai, ab, at, P, 0)
    okv = oracle.RefKVCacheModel(oracle.RefCausalLM(cfg, sd), 1, 20, 0.9)
    want = okv.forward_tree_attention(seq, prompt, mask, pids, pos.clone()).float()
    kv = hip.S.KVCacheModel(m, 1, 20, 0.9)
    got = kv.forward_tree_attention(seq.cuda(), prompt.cuda(), mask.cuda(), pids.cuda(), pos.clone().cuda()).float().cpu()
    tv = 0.5 * (got - want).abs().sum(-1)
    print(f"tree of {seq.shape[1]} nodes: max total variation {float(tv.max()):.4f}, mean {float(tv.mean()):.4f}")
    assert float(tv.max()) <= (0.12 if kv_dtype else 0.05)


def test_tree_attention_split_keys_chunk_without_visible_key(hip):
    """with the keys of a group cut over several workgroups, a chunk can hold only tree rows a node may not
    see (every score -inf).  Its partial must be (zeros, m = -inf, l = 0) - what attn_combine_kernel skips - not
    exp(-inf + inf) = NaN.  150 cached keys + 40 tree nodes in 32-key chunks: the last chunk is all tree rows."""
    from test_gpu_parity import MID_CFGS
    cfg = ModelConfig(**MID_CFGS["llama_d64_gqa"])
    sd = make_state_dict(cfg, 77, dtype=torch.bfloat16, gain=0.7)
    m = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=torch.bfloat16)
    rng = np.random.default_rng(12)
    P, W, LV = 150, 8, 5
    prompt = torch.from_numpy(rng.integers(3, cfg.vocab_size, size=(1, P)))
    ai = [torch.zeros(W, dtype=torch.long) for _ in range(LV)]
    ab = [torch.from_numpy(rng.integers(0, W, size=W)) for _ in range(LV)]
    at = [torch.from_numpy(rng.integers(3, cfg.vocab_size, size=W)) for _ in range(LV)]
    seq, mask, pos, pids = hip.S.get_seq_att_mask(1, ai, ab, at, P, 0)
    outs = []
    for split in (False, True):
        if split:
            os.environ["SD_ATTN_SPLIT_KEYS"], os.environ["SD_ATTN_KEYS_PER_SPLIT"] = "64", "32"
        try:
            kv = hip.S.KVCacheModel(m, 1, 20, 0.9)               # (the tunables are sampled when the session is created)
            outs.append(kv.forward_tree_attention(seq.cuda(), prompt.cuda(), mask.cuda(), pids.cuda(), pos.clone().cuda()).float().cpu())
        finally:
            os.environ.pop("SD_ATTN_SPLIT_KEYS", None)
            os.environ.pop("SD_ATTN_KEYS_PER_SPLIT", None)
    assert bool(torch.isfinite(outs[1]).all())
    # split attention keeps un-rounded exp() weights until the combine, the one-workgroup form rounds the probabilities to
    # bf16 first (as the reference does): both are held to the oracle's bf16 tree forward under the wide-tree test's bar
    okv = oracle.RefKVCacheModel(oracle.RefCausalLM(cfg, sd), 1, 20, 0.9)
    want = okv.forward_tree_attention(seq, prompt, mask, pids, pos.clone()).float()
    for o in outs:
        tv = 0.5 * (o - want).abs().sum(-1)
        assert float(tv.max()) <= 0.05, float(tv.max())


def test_model_from_local_checkpoint_directory(hip, tmp_path):
    """SpecDecModel.from_pretrained_dir (what bench.py uses under SPECDEC_MODEL_DIR): logits equal the from_hf route's
    bit for bit and match the module's own forward."""
    llama, _ = _hf_pair()
    llama.save_pretrained(str(tmp_path / "m"))
    a = hip.engine.SpecDecModel.from_pretrained_dir(str(tmp_path / "m"), dtype=torch.float32)
    b = hip.engine.SpecDecModel.from_hf(llama, dtype=torch.float32)
    ids = torch.from_numpy(np.random.default_rng(1).integers(3, 512, size=(12,))).to(torch.int32).cuda()
    la = a.new_session(32).forward(ids, 12).clone()
    lb = b.new_session(32).forward(ids, 12).clone()
    assert torch.equal(la, lb)
    with torch.no_grad():
        want = llama(ids[None].long().cpu()).logits[0].float()
    assert float((la.cpu() - want).abs().max()) <= 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_batched_prefill_equals_per_stream_prefill(hip, dtype):
    """engine.batch_prefill / sd_batch_prefill: the prompts of several streams in ONE pass over the
    weights (row table of contiguous runs, up to 256 rows / 32 attention groups) against stream-by-stream
    Session.forward: every stream's K / V rows and the logits of the step that follows.  fp32 is bit-identical (the GEMM
    is row-independent); bf16 takes another k-slab plan at another row count, so it is held to the bf16 bar of the other
    forward tests (0.25 on logits of scale ~10).  Five streams of 40 / 127 / 70 / 9 / 300 rows: two packed passes + one
    stream that does not fit a pass and goes through its own chunked forward."""
    cfg = ModelConfig(**BF16_CFG)
    sd = make_state_dict(cfg, 5, dtype=dtype)
    m = hip.engine.SpecDecModel.from_state_dict(cfg, sd, dtype=dtype)
    lens = [40, 127, 70, 9, 300]
    rng = np.random.default_rng(17)
    seqs = [torch.from_numpy(rng.integers(3, cfg.vocab_size, size=(n + 1,))).to(torch.int32).cuda() for n in lens]
    a = [m.new_session(320) for _ in lens]
    b = [m.new_session(320) for _ in lens]
    for ses, sq, n in zip(a, seqs, lens):
        ses.forward(sq[:n], 0)
    hip.engine.batch_prefill(b, seqs, lens)
    tol = 0.0 if dtype == torch.float32 else 0.25
    for sa, sb, sq, n in zip(a, b, seqs, lens):
        assert sb.cache_len == n == sa.cache_len
        ka, kb = sa.kv[:, :, :, :n].float(), sb.kv[:, :, :, :n].float()
        assert float((ka - kb).abs().max()) <= tol * 0.1, (n, float((ka - kb).abs().max()))
        la = sa.forward(sq[n:n + 1], 1).clone()
        lb = sb.forward(sq[n:n + 1], 1).clone()
        assert float((la - lb).abs().max()) <= tol, (n, float((la - lb).abs().max()))
    # capacity errors: more than 256 rows / a logits request
    items = (hip.L.SdBatchItem * 2)()
    for j in range(2):
        items[j].session, items[j].seq, items[j].pos0, items[j].n_new, items[j].n_logits = a[j].handle, seqs[j].data_ptr(), 0, 200, 0
    assert hip.lib.sd_batch_prefill(items, 2, _st()) == hip.L.SD_ERR_CAPACITY
    items[0].n_new, items[1].n_new, items[1].n_logits = 8, 8, 1
    assert hip.lib.sd_batch_prefill(items, 2, _st()) == hip.L.SD_ERR_INVALID


def test_fused_attention_oproj_launch_vs_two_launches(hip):
    """fused_kernels.h: attention + O projection as one launch (O's weights stream while attention runs; hand-off through
    a device counter and write-through stores) against the two-launch path (SD_FUSE_ATTN_O=0) at Llama-2-13b's layer
    shape (2 layers): K / V rows are bit-identical (they do not depend on the O projection); logits agree within the bf16
    bar of the other forward tests - the fused O projection folds ONE k-slab where the streaming kernel folds four - and,
    the point of the exercise, twenty repetitions of the same steps give bit-identical results every time: a stale read
    of the attention rows across XCDs would show up as run-to-run differences.  Steps of 1 - 9 rows take the fused
    launch (the 9-row step with two attention row groups), the 200-row prefill does not."""
    cfg = ModelConfig(arch="llama", vocab_size=32000, hidden_size=5120, intermediate_size=13824, num_hidden_layers=2,
                      num_attention_heads=40, num_key_value_heads=40, max_position_embeddings=512, rms_norm_eps=1e-5)
    m = hip.engine.SpecDecModel.synthetic(cfg, seed=9, dtype=torch.bfloat16, max_pos=400)
    ids = torch.from_numpy(np.random.default_rng(4).integers(3, cfg.vocab_size, size=(400,))).to(torch.int32).cuda()
    steps = (5, 1, 8, 5, 9, 5, 2)

    def run():
        ses = m.new_session(400)
        ses.forward(ids[:200], 0)
        got, pos = [], 200
        for q in steps:
            got.append(ses.forward(ids[pos:pos + q], q).clone())
            pos += q
        return torch.cat(got), ses.kv[:, :, :, :pos].clone()
    os.environ["SD_FUSE_ATTN_O"] = "0"
    try:
        ref_logits, ref_kv = run()
    finally:
        os.environ.pop("SD_FUSE_ATTN_O", None)
    first = run()
    assert torch.equal(first[1][0], ref_kv[0])                    # layer 0's K / V rows: independent of any O projection
    assert not bool(torch.isnan(first[0]).any())
    assert float((first[0] - ref_logits).abs().max()) <= 0.04 * float(ref_logits.abs().max())
    for _ in range(19):
        again = run()
        assert torch.equal(again[0], first[0]) and torch.equal(again[1], first[1])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,shape,dtype", [("1", "13b", torch.bfloat16), ("2", "13b", torch.bfloat16), ("2", "70b", torch.bfloat16),
                                              ("2", "13b", torch.float16)],
                         ids=["attention_seam", "both_seams", "both_seams_70b_gqa", "both_seams_fp16"])
def test_norm_on_load_layer_path_vs_residual_norm_launches(hip, mode, shape, dtype):
    """normload_kernels.h: for <= 16 rows of a 16-bit Llama model the residual add runs in the epilogue of the GEMM that
    produces the rows and RMSNorm in the operand load of the GEMM that consumes them (per-tile sums of squares handed
    over, summed in a fixed order), against the path with residual_norm_kernel launches (SD_NORM_ON_LOAD=0) at
    Llama-2-13b's layer shape (2 layers).  The two paths round the same values at the same points - x' = rnd(x + rnd(o)),
    rnd(w * rnd(x' * r)) - and differ only in the order in which the 5120 squares of a row are added up (r in its last
    bits), so: layer 0's K / V rows are bit-identical (they precede any residual add), the residual-stream-dependent
    logits agree within the bf16 bar of the other forward tests, and ten repetitions are bit-identical (no atomics, no
    order that depends on scheduling).  The 70b shape (hidden 8192, grouped-query attention) takes the MLP -> next-layer
    seam only (its O projection is too wide for the fused attention launch) and adds its partials up in two passes."""
    if shape == "13b":
        cfg = ModelConfig(arch="llama", vocab_size=32000, hidden_size=5120, intermediate_size=13824, num_hidden_layers=2,
                          num_attention_heads=40, num_key_value_heads=40, max_position_embeddings=512, rms_norm_eps=1e-5)
    else:
        cfg = ModelConfig(arch="llama", vocab_size=32000, hidden_size=8192, intermediate_size=28672, num_hidden_layers=3,
                          num_attention_heads=64, num_key_value_heads=8, max_position_embeddings=512, rms_norm_eps=1e-5)
    m = hip.engine.SpecDecModel.synthetic(cfg, seed=9, dtype=dtype, max_pos=400)
    ids = torch.from_numpy(np.random.default_rng(4).integers(3, cfg.vocab_size, size=(400,))).to(torch.int32).cuda()
    steps = (5, 1, 8, 5, 16, 3, 2, 4, 6, 7)

    def run():
        ses = m.new_session(400)
        ses.forward(ids[:200], 0)
        got, pos = [], 200
        for q in steps:
            got.append(ses.forward(ids[pos:pos + q], q).clone())
            pos += q
        return torch.cat(got), ses.kv[:, :, :, :pos].clone()
    os.environ["SD_NORM_ON_LOAD"] = "0"
    try:
        ref_logits, ref_kv = run()
    finally:
        os.environ.pop("SD_NORM_ON_LOAD", None)
    os.environ["SD_NORM_ON_LOAD"] = mode
    try:
        # the path under test really is taken: a 5-row step launches residual_norm_kernel once per layer seam that keeps it
        ses = m.new_session(400)
        ses.forward(ids[:200], 0)
        ses.profile(True)
        ses.forward(ids[200:205], 5)
        launches = ses.profile_read()["norm_residual"][1]
        ses.profile(False)
        L = cfg.num_hidden_layers
        kept = {("1", "13b"): L, ("2", "13b"): 1, ("2", "70b"): L + 1}[(mode, shape)]    # (70b: the attention seam keeps its launch)
        assert launches == kept, (launches, kept)
        first = run()
        assert torch.equal(first[1][0], ref_kv[0])
        assert not bool(torch.isnan(first[0]).any())
        assert float((first[0] - ref_logits).abs().max()) <= 0.04 * float(ref_logits.abs().max())
        for _ in range(9):
            again = run()
            assert torch.equal(again[0], first[0]) and torch.equal(again[1], first[1])
    finally:
        os.environ.pop("SD_NORM_ON_LOAD", None)


@pytest.mark.gpu
def test_norm_on_load_seams_with_rows_of_two_streams(hip):
    """The norm-on-load seams look at rows, not streams: a shared pass over 3 + 5 = 8 rows of two sequences (positions,
    KV arenas and attention differ per row through the row table) takes them at the 13b layer shape and must give each
    stream the logits of its own forward within the bf16 bar, and the same K / V rows in layer 0."""
    cfg = ModelConfig(arch="llama", vocab_size=32000, hidden_size=5120, intermediate_size=13824, num_hidden_layers=2,
                      num_attention_heads=40, num_key_value_heads=40, max_position_embeddings=512, rms_norm_eps=1e-5)
    m = hip.engine.SpecDecModel.synthetic(cfg, seed=12, dtype=torch.bfloat16, max_pos=256)
    rng = np.random.default_rng(5)
    lens, new = [60, 33], [3, 5]
    seqs = [torch.from_numpy(rng.integers(3, cfg.vocab_size, size=(L + n,)).astype(np.int32)).cuda() for L, n in zip(lens, new)]
    solo = [m.new_session(128) for _ in lens]
    both = [m.new_session(128) for _ in lens]
    want = []
    for a, b, sq, L, n in zip(solo, both, seqs, lens, new):
        a.forward(sq[:L], 0)
        b.forward(sq[:L], 0)
        want.append(a.forward(sq[L:L + n], n).clone())
    for b in both:
        b.profile(True)
    got = hip.engine.batch_forward(both, seqs, new, new).clone()
    assert both[0].profile_read()["norm_residual"][1] == 1 + cfg.num_hidden_layers     # attention seam (two streams: no fused launch) + the final norm
    want = torch.cat(want, 0)
    assert not bool(torch.isnan(got).any())
    assert float((got - want).abs().max()) <= 0.03 * float(want.abs().max())
    for a, b, L, n in zip(solo, both, lens, new):
        assert torch.equal(a.kv[0, :, :, :L + n], b.kv[0, :, :, :L + n])


# --------------------------------------------------------------------------- (f)4: the beam variant end to end (parity unpinned)
@pytest.mark.parametrize("arch,nb,thres,seed", [("llama", 3, 0.7, 0), ("llama", 2, 0.5, 1), ("llama", 5, 0.9, 2), ("opt", 4, 0.7, 3)])
def test_beam_speculative_sampling_v2_vs_oracle(hip, arch, nb, thres, seed):
    """beam_speculative_sampling_v2 (reference speculative_sampling.py:18-581, extra_sample_cnt == 1) on the engine - the
    draft's num_beams KV arenas sharing one weight pass per beam step, the reorder / per-step snapshots on the few rows
    that differ, the tree verify and the path compaction - against oracle.beam_ref's restatement fed the SAME variates
    (recorded from the oracle's run, replayed into the device path): tokens, accepted lengths, accepted-beam counts,
    expected counts and call counts must be identical, with accepted, rejected and all-accepted levels in the run, and
    with an EOS stop.  The draft side of this variant is PARITY UNPINNED (sampling/beam.py, oracle/beam_ref.py: the
    reference's own beam path needs transformers 4.35's BeamSearchScorer, absent here); its target side is pinned by G9."""
    from oracle import beam_ref
    from llmspeculativesampling_amd.noise import ReplayNoise
    if arch == "llama":
        dc, dsd, tc, tsd = _pair("corr", seed=11 + seed)
    else:
        dc, dsd, tc, tsd = _pair("opt", seed=11 + seed)
    V = dc.vocab_size
    prompt = torch.from_numpy(np.random.default_rng(60 + seed).integers(3, V, size=(1, 9)))
    kw = dict(gamma=4, width=nb, num_beams=nb, extra_sample_cnt=1, expect_thres=thres, top_k=20, top_p=0.9, details=True)
    od, ot = oracle.RefCausalLM(dc, dsd), oracle.RefCausalLM(tc, tsd)
    torch.manual_seed(seed)
    probe, _ = beam_ref.beam_speculative_sampling_v2(prompt, od, ot, -1, None, 20, **kw)
    dm = hip.engine.SpecDecModel.from_state_dict(dc, dsd, dtype=torch.float32)
    tm = hip.engine.SpecDecModel.from_state_dict(tc, tsd, dtype=torch.float32)
    for eos in (-1, int(probe[0, 9 + 12])):
        rec = oracle.RecordingNoise()
        torch.manual_seed(seed)
        want, wd = beam_ref.beam_speculative_sampling_v2(prompt, od, ot, eos, None, 20, noise=rec, **kw)
        got, gd = hip.S.beam_speculative_sampling_v2(prompt.cuda(), dm, tm, eos, None, 20, rng=ReplayNoise(rec.events, "cuda"), **kw)
        np.testing.assert_array_equal(got.cpu().numpy(), want.numpy())
        for key in ("acc_len", "num_beams_list", "expect_cnt_list", "target_call_times", "approx_call_times"):
            assert gd[key] == wd[key], key
        assert abs(float(gd["acc_rate"]) - float(wd["acc_rate"])) < 1e-6
        if eos >= 0:
            assert int(want[0, -1]) == eos and want.shape[1] < 9 + 20
        elif arch == "llama":                                    # (the OPT pair is unrelated models: hardly any accept)
            assert 0 < sum(wd["acc_len"]) < 4 * len(wd["acc_len"])
    # device Philox draws: same loop, no host noise; the run must be well-formed and reproducible
    a, ad = hip.S.beam_speculative_sampling_v2(prompt.cuda(), dm, tm, -1, None, 20, rng=hip.noise.DeviceNoise(5), **kw)
    b, _ = hip.S.beam_speculative_sampling_v2(prompt.cuda(), dm, tm, -1, None, 20, rng=hip.noise.DeviceNoise(5), **kw)
    assert torch.equal(a, b) and a.shape[1] == 9 + sum(x + 1 for x in ad["acc_len"]) >= 9 + 20
