"""CPU-side checks that need no GPU: the C ABI loads and exports every symbol of include/specdec.h, the host
logic (noise providers, configs, synthetic weights, stream sharding + gather over gloo with world_size 2)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes as C
    hdr = open(os.path.join(ROOT, "include", "specdec.h")).read()
    declared = set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", hdr))
    types = {"sd_status", "sd_dtype", "sd_arch", "sd_accept_result", "sd_model_config", "sd_model_weights",
             "sd_model", "sd_session"}
    declared -= types
    assert len(declared) >= 18
    from llmspeculativesampling_amd import _lib
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert _lib.lib.sd_version() == 4           # a host-only call: no GPU needed


def test_struct_layouts_match_header():
    import ctypes as C
    from llmspeculativesampling_amd import _lib
    assert C.sizeof(_lib.SdAcceptResult) == 16 + 64 + 64 + 64
    assert C.sizeof(_lib.SdMultiResult) == 208 + 16 + 2 * 1024 and C.sizeof(_lib.SdMultiItem) == 24
    assert C.sizeof(_lib.SdModelConfig) == 15 * 4
    assert C.sizeof(_lib.SdBatchStream) == 9 * 8 + 5 * 4 + 4 + 2 * 8 + 2 * 4 + 3 * 8      # (4 bytes of padding before `seed`)
    assert C.sizeof(_lib.SdModelWeights) == 21 * 8


def test_argument_errors_do_not_need_a_gpu():
    import ctypes as C
    from llmspeculativesampling_amd import _lib
    with pytest.raises(ValueError):
        _lib.check(_lib.lib.sd_norm_probs(None, 1, 8, 8, 1.0, 0, 0.0, 0, None, 8, None, None, None), "sd_norm_probs")
    buf = (C.c_float * 8)()
    with pytest.raises(ValueError, match="temperature"):
        _lib.check(_lib.lib.sd_norm_probs(C.addressof(buf), 1, 8, 8, 0.0, 0, 0.0, 0, C.addressof(buf), 8, None, None, None),
                   "sd_norm_probs")
    with pytest.raises(ValueError, match="N % 16"):
        _lib.check(_lib.lib.sd_pack_weight_bf16(C.addressof(buf), C.addressof(buf), 8, 32, None), "sd_pack_weight_bf16")


def test_configs_and_param_counts():
    from llmspeculativesampling_amd.config import load_config
    c13 = load_config("llama-2-13b")
    # W_stream = 12.852 G params for Llama-2-13b (norm vectors add 0.4 M)
    assert abs(c13.n_params(streamed_only=True) - 12.852e9) < 2e6
    assert c13.head_dim == 128
    c68 = load_config("llama-68m")
    assert abs(c68.n_params(streamed_only=True) - 43.45e6) < 1e5
    o350 = load_config("opt-350m")
    assert not o350.do_layer_norm_before and o350.word_embed_proj_dim == 512
    assert abs(o350.n_params() - 331.2e6) < 1e6           # reference class has 331.2 M
    assert abs(load_config("opt-125m").n_params() - 125.2e6) < 1e6


def test_synthetic_weights_are_platform_stable():
    from llmspeculativesampling_amd.config import load_config
    from llmspeculativesampling_amd.synth import make_state_dict
    sd = make_state_dict(load_config("tiny-llama-draft"), 21)
    # fingerprint recorded when the golden fixtures were generated (numpy PCG64 is bit-stable across platforms)
    v = sd["model.layers.0.self_attn.q_proj.weight"]
    assert v.shape == (32, 32)
    assert abs(float(v.double().sum()) - (-6.197300100546272)) < 1e-6, float(v.double().sum())


def test_replay_noise_contract():
    from llmspeculativesampling_amd.noise import ReplayNoise
    ev = [("exp", torch.ones(4)), ("seed", 7), ("uni", torch.tensor([0.25])), ("seed", 7), ("uni", torch.tensor([0.25])),
          ("exp", torch.ones(4))]
    nz = ReplayNoise(ev, "cpu")
    assert nz.exponential(4).shape == (4,)
    r, tok = nz.uniforms(4, 7)
    assert r.tolist() == [0.25, 0.25, 2.0, 2.0]
    nz.realign(tok, 2)
    with pytest.raises(RuntimeError):
        nz.realign(tok, 3)
    nz.skip_exponential(4)
    assert nz.exhausted()
    with pytest.raises(RuntimeError, match="exhausted"):
        nz.exponential(4)


def test_host_torch_noise_realign_matches_lazy_draws():
    """Drawing gamma uniforms up front and re-aligning == the reference's lazy draws that stop at a reject."""
    from llmspeculativesampling_amd.noise import HostTorchNoise
    for consumed in (1, 2, 4):
        torch.manual_seed(5)
        lazy = [float(torch.rand(1)) for _ in range(consumed)]
        after_lazy = float(torch.rand(1))
        torch.manual_seed(5)
        nz = HostTorchNoise("cpu")
        r, tok = nz.uniforms(4, None)
        nz.realign(tok, consumed)
        assert r.tolist()[:consumed] == lazy
        assert float(torch.rand(1)) == after_lazy
    torch.manual_seed(9)
    nz = HostTorchNoise("cpu")
    r, tok = nz.uniforms(4, 42)                # reseed quirk: all equal, nothing to re-align
    assert tok is None and len(set(r.tolist())) == 1
    torch.manual_seed(42)
    assert float(torch.rand(1)) == r.tolist()[0]


def test_shard_streams_round_robin():
    from llmspeculativesampling_amd.dist import shard_streams
    assert shard_streams(64, 3, 8) == list(range(3, 64, 8))
    allr = sorted(s for r in range(8) for s in shard_streams(61, r, 8))
    assert allr == list(range(61))


def _gloo_worker(rank, world, port, n_streams, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from llmspeculativesampling_amd.dist import gather_streams, shard_streams
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    outs = []
    for s in shard_streams(n_streams, rank, world):
        n = 5 + s                                         # ragged lengths
        outs.append(torch.arange(n, dtype=torch.int64).unsqueeze(0) + 100 * s)
    res = gather_streams(outs, n_streams, width=32, device="cpu")
    ok = all(res[s].tolist() == (torch.arange(5 + s) + 100 * s).tolist() for s in range(n_streams))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)              # the timing reduction bench.py does
    q.put((rank, ok, float(t)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_streams", [4, 5])
def test_gather_streams_world_size_2_gloo(n_streams):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + n_streams
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, n_streams, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in got)
    assert all(t == 2.0 for _, _, t in got)


def _gloo_fallback_worker(rank, world, port, mode, q):
    """gather_streams with libspecdec's own collective REQUESTED (as on an RCCL group) but failing in `mode`:
    'probe' - RCCL not resolvable on rank 1 only; 'id' - rank 0 cannot create the unique id; 'init' - sd_comm_init fails
    on rank 1 only (rank 0's succeeds: a stub handle); 'ok' - every step succeeds (stub collective)."""
    import ctypes as C
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from llmspeculativesampling_amd import dist as D
    from llmspeculativesampling_amd import _lib
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    calls = []

    class FakeLib:
        def sd_comm_probe(self):
            calls.append("probe")
            return 1 if (mode == "probe" and rank == 1) else 0

        def sd_comm_unique_id(self, buf):
            calls.append("id")
            if mode == "id":
                return 1
            C.memmove(buf, bytes(range(128)), 128)
            return 0

        def sd_comm_init(self, r, w, ident, out):
            calls.append("init")
            assert bytes(ident) == bytes(range(128)) and (r, w) == (rank, world)
            if mode == "init" and rank == 1:
                return 1
            out._obj.value = 0x1234                               # (byref(handle): a non-null stub communicator)
            return 0

        def sd_comm_destroy(self, h):
            calls.append("destroy")
            return 0

        def sd_last_error(self):
            return b"stub failure"
    fake = FakeLib()
    _lib.lib = fake                                           # dist.py imports `lib` from _lib at call time
    D._use_own_collective = lambda group, mine: True          # as if the group were RCCL and the buffers on a GPU
    gathered_by = []
    real_all_gather = dist.all_gather
    dist.all_gather = lambda *a, **k: (gathered_by.append("torch"), real_all_gather(*a, **k))[1]
    D.TokenComm.all_gather_tokens = lambda self, mine: (gathered_by.append("own"), torch.stack(
        [t for t in (lambda g: (real_all_gather(g, mine), g)[1])([torch.empty_like(mine) for _ in range(world)])]))[1]
    outs = [torch.arange(5 + s, dtype=torch.int64).unsqueeze(0) + 100 * s for s in D.shard_streams(5, rank, world)]
    res = D.gather_streams(outs, 5, width=32, device="cpu")
    res2 = D.gather_streams(outs, 5, width=32, device="cpu")    # the decision is cached: no second negotiation
    ok = all(r[s].tolist() == (torch.arange(5 + s) + 100 * s).tolist() for r in (res, res2) for s in range(5))
    q.put((rank, ok, gathered_by, calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["probe", "id", "init", "ok"])
def test_gather_streams_fallback_never_hangs_world_size_2_gloo(mode):
    """whichever step of creating libspecdec's RCCL communicator fails, on whichever rank, BOTH ranks
    must take the same sequence of collectives, end up on torch.distributed's all_gather together and return the right
    streams - the old code skipped a broadcast on the failing rank and hung the others.  Two gloo ranks with a stubbed
    sd_comm_*; 'ok' checks that the own collective is used when every step succeeds."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30400 + (os.getpid() % 400) + {"probe": 0, "id": 1, "init": 2, "ok": 3}[mode]
    procs = [ctx.Process(target=_gloo_fallback_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in got), got
    want = "own" if mode == "ok" else "torch"
    assert all(by == [want, want] for _, _, by, _ in got), got
    calls = {r: c for r, _, _, c in got}
    if mode == "probe":
        assert calls[0] == ["probe"] and calls[1] == ["probe"]                    # nobody goes on to the id / init
    elif mode == "id":
        assert calls[0] == ["probe", "id"] and calls[1] == ["probe"]              # None was broadcast: no init anywhere
    elif mode == "init":
        assert calls[0] == ["probe", "id", "init", "destroy"] and calls[1] == ["probe", "init"]
    else:
        assert calls[0] == ["probe", "id", "init"] and calls[1] == ["probe", "init"]


# --------------------------------------------------------------------------- evaluation-driver host pieces (8(f) rank 3)
def test_harness_chatalpaca_reader_and_power_sum(tmp_path):
    """read_chatalpaca: one prompt per assistant turn with the cumulative, newline-joined history
    (reference evaluation.py:347-364); total_power: samples strictly inside (t1, t2), first one dropped, cut-off
    lines skipped (evaluation.py:135-152)."""
    import json
    from llmspeculativesampling_amd import harness
    p = tmp_path / "chat.json"
    convs = [{"conversations": [{"from": "human", "value": "hi"}, {"from": "gpt", "value": "hello"},
                                {"from": "human", "value": "more"}, {"from": "gpt", "value": "sure"}]},
             {"conversations": [{"from": "human", "value": "a"}, {"from": "human", "value": "b"},
                                {"from": "gpt", "value": "c"}]}]
    p.write_text("\n".join(json.dumps(c) for c in convs) + "\n")
    prompts, answers = harness.read_chatalpaca(str(p))
    assert prompts == ["hi\n", "hi\nhello\nmore\n", "a\nb\n"]
    assert answers == ["hello", "sure", "c"]
    lines = ["9.0 100.0", "10.5 200.0", "11.5 300.0", "12.5 400.0", "13.5", "garbage x", "20.0 500.0"]
    assert harness.total_power(lines, 10.0, 13.0) == 700.0          # 200 is the dropped first sample
    assert harness.total_power([], 0.0, 1.0) == 0.0
    tok = harness.ByteTokenizer(512)
    ids = tok.encode("héllo\n", return_tensors="pt")
    assert ids.shape[0] == 1 and int(ids[0, 0]) == 1 and tok.decode(ids[0].tolist()) == "héllo\n"
    ds = harness.synthetic_prompts(5, 1000)
    assert all(d.shape[0] == 1 and 32 <= d.shape[1] <= 512 and int(d.min()) >= 3 and int(d.max()) < 1000 for d in ds)
    assert [d.shape[1] for d in ds] == [d.shape[1] for d in harness.synthetic_prompts(5, 1000)]
    agg = dict(approx_time=2e9, target_time=3e9, other_time=1e9, acc_len_sum=30.0, acc_rate=[0.5, 0.7],
               target_call_times=10, approx_call_times=10, target_model_time=2.5e9, target_pre_cache_time=0.25e9,
               target_post_prob_time=0.5e9)
    out = harness.speculative_log_lines("google speculative decoding (with KVCache)", int(6e9), 40, agg, [-1.0, -3.0], 90.0)
    assert "total time 6.0 s, total tokens 40, average time 0.15 s/token" in out[0]
    assert out[2].startswith("average accepted len 3.0, target call times 10, acc rate 0.6")
    assert out[-2] == "power/token: 2.25"
    assert out[-1] == "target_model_time: 2.5, pre cache time: 0.25, post prob time: 0.5"      # evaluation.py:581


def test_sampling_package_exports_every_reference_name():
    """The reference harness imports nine names from ``sampling`` (sampling/__init__.py:1-7, evaluation.py:13-14): all of
    them resolve here; the out-of-scope variants raise NotImplementedError when called."""
    import llmspeculativesampling_amd.sampling as S
    ref_all = ["speculative_sampling", "speculative_sampling_v2", "autoregressive_sampling", "multi_speculative_sampling",
               "beam_speculative_sampling", "BiLD_sampling", "mjsd_speculative_sampling", "random_width_beam_sampling",
               "beam_speculative_sampling_v2"]
    for n in ref_all:
        assert callable(getattr(S, n)) and n in S.__all__
    for n in ("speculative_sampling_v2", "beam_speculative_sampling", "BiLD_sampling", "mjsd_speculative_sampling",
              "random_width_beam_sampling"):
        with pytest.raises(NotImplementedError):
            getattr(S, n)(None, None, None)
    with pytest.raises(NotImplementedError, match="extra_sample_cnt"):           # built for one input sequence per verify
        S.beam_speculative_sampling_v2(None, None, None, 2, None, 8, num_beams=4, extra_sample_cnt=2)


# --------------------------------------------------------------------------- bench.py launcher
def _run_bench(extra_args, extra_env, timeout=180):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPECDEC_BENCH_STUB="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra_args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus 2` (the driver's command shape, no launcher environment) must itself start 2 ranks that
    rendezvous on 127.0.0.1, run the timed protocol and print ONE JSON line with n_gpus = 2 (stub decode step, gloo)."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--prompt-len", "8", "--max-len", "4"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["new_tokens"] == 2 * 3 * 4 and d["scaling"] == "weak"


def test_bench_single_rank_and_world_mismatch():
    import json
    r = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "0", "--prompt-len", "8", "--max-len", "4"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1
    # a launcher environment that disagrees with --gpus is an error, never a mislabelled line
    r = _run_bench(["--gpus", "2", "--steps", "1"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_launcher_propagates_rank_failure():
    r = _run_bench(["--gpus", "2", "--steps", "1", "--prompt-len", "8", "--max-len", "4"], {"SPECDEC_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0


# --------------------------------------------------------------------------- tensor-parallel sharding (config 5)
def _tp_gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import oracle
    from oracle.tp_ref import llama_forward_tp
    from oracle.models_ref import llama_forward
    from llmspeculativesampling_amd.config import ModelConfig
    from llmspeculativesampling_amd.synth import make_state_dict
    from llmspeculativesampling_amd.tp import shard_config, shard_tensor
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = ModelConfig(arch="llama", vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=64, rms_norm_eps=1e-5)
    sd = make_state_dict(cfg, 7)                       # every rank builds the same full model, keeps its slice
    local = shard_config(cfg, world)
    lsd = {n: shard_tensor(cfg, n, t, rank, world) for n, t in sd.items()}

    def ar(t):
        t = t.clone()
        dist.all_reduce(t)
        return t
    ids = torch.arange(3, 15)[None]
    lg, past = llama_forward_tp(local, lsd, ids[:, :9], None, ar)
    lg2, _ = llama_forward_tp(local, lsd, ids[:, 9:], past, ar)
    want, wpast = llama_forward(cfg, sd, ids[:, :9], None)
    want2, _ = llama_forward(cfg, sd, ids[:, 9:], wpast)
    err = max(float((lg - want).abs().max()), float((lg2 - want2).abs().max()))
    shapes_ok = (local.num_attention_heads, local.num_key_value_heads, local.intermediate_size, local.head_dim) == (2, 1, 128, 32) \
        and tuple(past[0][0].shape) == (1, 1, 9, 32)
    q.put((rank, err, shapes_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_tensor_parallel_sharding_world_size_2_gloo():
    """tp.shard_config / tp.shard_tensor (Megatron slices of a GQA Llama: whole heads by rows for q/k/v and gate/up, input
    columns for o/down) + an all-reduce of the two row-parallel outputs per layer == the unsharded forward; two gloo ranks,
    prefill + incremental step with per-shard KV (one KV head each)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 400)
    procs = [ctx.Process(target=_tp_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, _, ok in got)
    assert all(err < 1e-4 for _, err, _ in got), got


def test_shard_config_rejects_uneven_splits_and_non_llama():
    from llmspeculativesampling_amd.config import load_config
    from llmspeculativesampling_amd.tp import shard_config
    c70 = load_config("llama-2-70b")
    l8 = shard_config(c70, 8)
    assert (l8.num_attention_heads, l8.num_key_value_heads, l8.intermediate_size, l8.head_dim, l8.hidden_size) == (8, 1, 3584, 128, 8192)
    assert abs(l8.n_params(True) * 2 / 1e9 - 17.64) < 0.1          # GB of bf16 weights one rank streams per verify
    with pytest.raises(ValueError):
        shard_config(c70, 16)                                       # 8 KV heads do not split over 16 ranks
    with pytest.raises(NotImplementedError):
        shard_config(load_config("opt-13b"), 2)


# --------------------------------------------------------------------------- tree-attention host helpers (8(f) rank 4)
def test_tree_helpers_match_reference_fixtures():
    """The drop-in's get_seq_att_mask / get_num_acc_prob / get_expect_cnt_by_thres (host-side index and scalar
    arithmetic, reference utils.py:95-148, 247-350) against the values recorded from the reference (G9)."""
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_io import load
    import llmspeculativesampling_amd.sampling as S
    meta, blobs = load("g9_tree")
    for case in meta["tree"]:
        key = case["id"]
        for suffix, plen in (("", case["P"]), ("2", int(blobs[key + "_prefix2"].shape[1]))):
            tok, beam = torch.from_numpy(blobs[f"{key}_tok{suffix}"]), torch.from_numpy(blobs[f"{key}_beam{suffix}"])
            ai = [torch.zeros(tok.shape[1], dtype=torch.long) for _ in range(tok.shape[0])]
            seq, mask, pos, pids = S.get_seq_att_mask(1, ai, list(beam), list(tok), plen, 0)
            if suffix == "":
                for got, nm in ((seq, "seq"), (mask, "mask"), (pos, "pos"), (pids, "pids")):
                    np.testing.assert_array_equal(got.numpy(), blobs[f"{key}_{nm}"])
            assert mask.shape == (1, seq.shape[1], plen + seq.shape[1]) and bool(mask[0, :, :plen].all())
    for case in meta["dp"]:
        p, q = torch.from_numpy(blobs[case["id"] + "_p"]), torch.from_numpy(blobs[case["id"] + "_q"])
        prob, expect = S.get_num_acc_prob(p, q, case["m"])
        np.testing.assert_allclose(prob.numpy(), blobs[case["id"] + "_prob"], atol=1e-6)
        assert abs(float(expect) - case["expect"]) < 1e-5
        assert [S.get_expect_cnt_by_thres(prob, th) for th in case["thres"]] == case["counts"]
    # two inputs (extra_sample_cnt = 2): ragged rows are padded, the padding rows see nothing of the tree
    ai = [torch.tensor([0, 1, 0]), torch.tensor([0, 1, 1])]
    seq, mask, pos, pids = S.get_seq_att_mask(2, ai, [torch.tensor([0, 1, 0]), torch.tensor([0, 1, 2])],
                                              [torch.tensor([5, 6, 7]), torch.tensor([8, 9, 10])], 4, 99)
    assert seq.tolist() == [[5, 7, 8], [6, 9, 10]] and pos[:2].tolist() == [[0, -1], [1, -1]]
    assert pids.tolist() == [[4, 4, 5], [4, 5, 5]]


def test_library_binds_to_the_hip_runtime_torch_loaded():
    """libspecdec.so has to share ONE HIP runtime with torch (which ships its own libamdhip64): loaded before torch it
    pulls in /opt/rocm's copy, and the first launch on a torch device pointer then fails with "no ROCm-capable device"
    (seen with __graft_entry__.build() followed by smoke() in one process).  _lib therefore imports torch first."""
    import subprocess
    import sys
    code = ("import sys; import llmspeculativesampling_amd._lib as L; assert 'torch' in sys.modules; "
            "maps = open('/proc/self/maps').read(); "
            "hips = sorted({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}); "
            "assert len(hips) == 1, hips; print(hips[0])")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "torch" in out.stdout                                   # the one runtime in the process is torch's copy


def test_new_entry_points_reject_null_arguments_without_a_gpu():
    """Argument validation of the round-2 entries happens before any HIP call: NULL handles / buffers come back as
    SD_ERR_INVALID with a message (no GPU needed)."""
    import ctypes as C
    from llmspeculativesampling_amd import _lib
    lib = _lib.lib
    z = C.c_int(0)
    assert lib.sd_spec_generate(None, None, None, 8, 2, 0, None, None, 0, None, None, None, None, 1, None, None, None, None,
                                None, C.byref(z), C.byref(z), None) == _lib.SD_ERR_INVALID
    assert b"sd_spec_generate" in lib.sd_last_error()
    assert lib.sd_spec_batch_generate(None, 0, 4, 1.0, 20, 0.9, 32000, 32000, 2, 0, None, 0, 0, None, 0, None, 0, None, 64,
                                      None, None, None, 0, C.byref(z), C.byref(z), None) == _lib.SD_ERR_INVALID
    assert b"sd_spec_batch_generate" in lib.sd_last_error()
    assert lib.sd_accept_resample(None, None, 32000, 32000, None, 4, 4, None, 1, 0, 0, None, None, 0, 0, None, None) \
        == _lib.SD_ERR_INVALID
    assert lib.sd_norm_probs_lists(None, 1, 32000, 32000, 1.0, 20, 0.9, 0, None, 32000, None, None, None, None) \
        == _lib.SD_ERR_INVALID
    assert lib.sd_session_fused_status(None, None) == _lib.SD_ERR_INVALID
    assert lib.sd_session_test_skew_wait(None, 1) == _lib.SD_ERR_INVALID and lib.sd_session_ao_stamps(None, None, 1) == _lib.SD_ERR_INVALID
    # round 3: the RCCL gather of the sharded streams (sd_comm_*) and the filter's dtype mode
    h = C.c_void_p()
    assert lib.sd_comm_init(0, 1, None, C.byref(h)) == _lib.SD_ERR_INVALID and b"sd_comm_init" in lib.sd_last_error()
    assert lib.sd_comm_init(2, 2, C.addressof((C.c_char * 128)()), C.byref(h)) == _lib.SD_ERR_INVALID      # rank outside the world
    assert lib.sd_comm_all_gather_tokens(None, None, None, 1, 8, None) == _lib.SD_ERR_INVALID
    assert lib.sd_comm_rank(None, C.byref(z), C.byref(z)) == _lib.SD_ERR_INVALID
    assert lib.sd_comm_destroy(None) == _lib.SD_OK
    assert lib.sd_batch_prefill(None, 0, None) == _lib.SD_ERR_INVALID and b"sd_batch_prefill" in lib.sd_last_error()
    buf = (C.c_float * 16)()
    assert lib.sd_topk_topp_filter(C.addressof(buf), 1, 16, 16, 4, 0.9, 7, C.addressof(buf), 16, None) == _lib.SD_ERR_INVALID
    assert b"dtype_mode" in lib.sd_last_error()
    assert lib.sd_cand_list_bytes(3) == 3 * (4 + 128 * 4 + 128 * 4) and lib.sd_cand_list_bytes(0) == 0


def test_local_checkpoint_getter_and_bench_lookup(tmp_path, monkeypatch):
    """SURVEY.md 8(d) / VERDICT r2: SPECDEC_MODEL_DIR.  A checkpoint directory written by save_pretrained is read tensor by
    tensor through safetensors (nothing is executed from the files, nothing is fetched); bench.find_checkpoint resolves the
    BASELINE names inside $SPECDEC_MODEL_DIR and returns None when the variable or the directory is absent."""
    import transformers
    import bench
    from llmspeculativesampling_amd.engine import checkpoint_getter
    torch.manual_seed(0)
    lc = transformers.LlamaConfig(vocab_size=256, hidden_size=32, num_hidden_layers=2, intermediate_size=64,
                                  num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=64,
                                  tie_word_embeddings=False)
    oc = transformers.OPTConfig(vocab_size=256, hidden_size=32, num_hidden_layers=1, ffn_dim=64, num_attention_heads=4,
                                max_position_embeddings=64, do_layer_norm_before=True, word_embed_proj_dim=32)
    for sub, mod in (("llama-68m", transformers.LlamaForCausalLM(lc)), ("facebook/opt-125m", transformers.OPTForCausalLM(oc))):
        d = tmp_path / sub
        d.mkdir(parents=True)
        mod.save_pretrained(str(d))
        cfg, get, names = checkpoint_getter(str(d))
        sd = mod.state_dict()
        assert cfg.hidden_size == 32 and cfg.vocab_size == 256
        for n in names:
            want = sd[n] if n in sd else sd["model.decoder.embed_tokens.weight"]       # tied OPT head
            assert torch.equal(get(n), want), n
    monkeypatch.delenv("SPECDEC_MODEL_DIR", raising=False)
    assert bench.find_checkpoint("llama-68m") is None
    monkeypatch.setenv("SPECDEC_MODEL_DIR", str(tmp_path))
    assert bench.find_checkpoint("llama-68m") == str(tmp_path / "llama-68m")
    assert bench.find_checkpoint("opt-125m") == str(tmp_path / "facebook" / "opt-125m")
    assert bench.find_checkpoint("llama-2-13b") is None


def test_host_code_under_address_sanitizer():
    """SURVEY.md section 5 / VERDICT r2: a host-side -fsanitize=address configuration.  libspecdec's host C++ (argument
    validation, struct hand-off, RCCL / loop plumbing reachable without a GPU) is rebuilt with AddressSanitizer
    (_build.build_asan: host code only, ~12 s) and the ABI / argument tests of this file run against it in a child
    interpreter with the sanitizer runtime preloaded; any report fails the child."""
    import subprocess
    if os.environ.get("SD_ASAN_CHILD"):
        pytest.skip("already inside the sanitizer child")
    from llmspeculativesampling_amd import _build
    lib = _build.build_asan()
    rt = _build.asan_runtime()
    assert os.path.exists(lib) and os.path.exists(rt), (lib, rt)
    env = dict(os.environ, LD_PRELOAD=rt, SD_LIBSPECDEC=lib, SD_ASAN_CHILD="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23")
    sel = ("test_library_exports_every_declared_symbol or test_struct_layouts_match_header or "
           "test_argument_errors_do_not_need_a_gpu or test_new_entry_points_reject_null_arguments_without_a_gpu")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", sel, "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out, out[-4000:]
    assert r.returncode == 0 and "4 passed" in out, out[-4000:]
