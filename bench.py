#!/usr/bin/env python3
"""Headline benchmark: accepted tokens/sec + mean accept-len of speculative_sampling,
llama-68m -> Llama-2-13b, gamma = 4, bf16, prompt 128, max_len 128 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W          (N > 1: this process spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one speculative_sampling() call over one prompt stream (SURVEY.md 8(d) C2 / C4 inputs:
stream s has prompt seed 1000+s and RNG seed 2000+s).  Streams are independent, so ranks shard them
with no collective on the data path; one all_gather of token ids at the end of the timed region
(SURVEY.md 8(e)).  Weights are random-init from the committed config JSONs (no checkpoints offline), so the
accept length of the headline pair is ~0 by construction and `value` is essentially 1 token per draft+verify
iteration; `mean_accept_len` is reported next to it, and `acceptance_sweep` repeats the same workload on a
synthetic pair of the same two architectures whose distributions are correlated by a knob (synth.py, "Acceptance
dial"), so that partial accepts, the all-accept branch and the bonus sample are timed too.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--draft", default="llama-68m")
    ap.add_argument("--target", default="llama-2-13b")
    ap.add_argument("--gamma", type=int, default=4)
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--prompt-lens", default="fixed", choices=["fixed", "synthetic-c3"],
                    help="fixed: every stream's prompt has --prompt-len tokens (BASELINE configs[1], [3]); synthetic-c3: "
                         "SURVEY.md 8(d) C3's fallback inputs for configs[2] when chatalpaca is not on disk - 100 synthetic "
                         "prompts with lengths U{32..512}, seed 5 (harness.synthetic_prompts, reference evaluation.py:347-364 "
                         "reads the real set); stream s takes prompt s mod 100")
    ap.add_argument("--max-len", type=int, default=128)
    ap.add_argument("--top-k", type=int, default=20)
    ap.add_argument("--top-p", type=float, default=0.9)
    ap.add_argument("--rng", default="device", choices=["device", "host"],
                    help="device = on-device Philox (throughput mode); host = torch CPU generator in the reference's order")
    ap.add_argument("--cpu-baseline", type=int, default=1, help="time the CPU oracle on a bounded sample (rank 0, N=1)")
    ap.add_argument("--cpu-max-len", type=int, default=32)
    ap.add_argument("--cpu-budget-s", type=float, default=45.0, help="wall budget for the CPU baseline runs")
    ap.add_argument("--cpu-prompt-len", type=int, default=128)
    ap.add_argument("--batch-streams", "--streams-per-gpu", dest="batch_streams", type=int, default=1,
                    help="throughput mode (SURVEY 8(f)): this many prompt streams decode in lockstep per step and share every "
                         "weight pass (speculative_sampling_batch); 1 = the single-stream path of BASELINE configs[1].  "
                         "BASELINE configs[3] (64 streams over 8 GPUs) is `--gpus 8 --streams-per-gpu 8 --steps 1`")
    ap.add_argument("--profile-classes", type=int, default=1, help="per-op-class HIP-event timing of one verify step")
    ap.add_argument("--accept-sweep", type=int, default=1,
                    help="after the headline, time the same workload on the acceptance-dial pair at each --sweep-sigmas (N=1)")
    ap.add_argument("--sweep-sigmas", default="0,0.04,0.08,0.16,0.32,1.0")
    ap.add_argument("--sweep-steps", type=int, default=4)
    ap.add_argument("--tp", type=int, default=1,
                    help="tensor-parallel degree of the TARGET (BASELINE config 5: --target llama-2-70b --tp 8 --kv-dtype fp8): "
                         "all --gpus ranks then decode ONE stream per step together (tp must equal gpus)")
    ap.add_argument("--kv-dtype", default="model", choices=["model", "fp8"], help="dtype of the target's KV arena")
    return ap.parse_args(argv)


class _stdout_to_stderr:
    """Native libraries (gloo's "[Gloo] Rank 0 is connected to ..." lines) print on file descriptor 1 while a process group is set
    up; the contract is ONE JSON line on stdout, so fd 1 points at stderr for that span."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks (fresh interpreters, one per GPU, RCCL
    rendezvous on 127.0.0.1) BEFORE anything in this process touches the GPU, and wait for them.  Rank 0 prints the
    JSON line; the exit code is non-zero if any rank failed."""
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0:
                rc = rc or code
                print(f"[bench] rank {r} exited with code {code}", file=sys.stderr, flush=True)
                for q in alive:                      # a dead rank would leave the others in a collective forever
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def stub_worker(args, rank, world):
    """SPECDEC_BENCH_STUB=1 (CPU tests of the launcher): the real protocol - barrier, K timed steps, the gather of
    the generated ids, max-over-ranks time, one JSON line from rank 0 - over gloo with a fake decode step."""
    import torch.distributed as dist
    from llmspeculativesampling_amd.dist import gather_streams
    if os.environ.get("SPECDEC_BENCH_STUB_FAIL_RANK") == str(rank):
        return 3                                     # launcher test: a rank that dies before the rendezvous
    if world > 1:
        with _stdout_to_stderr():
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
    t0 = time.time()
    outs, new_tokens = [], 0
    for i in range(args.steps):
        s = rank + i * world
        outs.append(torch.arange(args.prompt_len + args.max_len, dtype=torch.int64)[None] + s)
        new_tokens += args.max_len
    width = args.prompt_len + args.max_len + args.gamma + 1
    if world > 1:
        streams = gather_streams(outs, args.steps * world, width, device="cpu")
        assert [int(x[0]) for x in streams] == list(range(args.steps * world))
        dist.barrier()
    stats = torch.tensor([time.time() - t0, float(new_tokens)], dtype=torch.float64)
    if world > 1:
        tmax, tot = stats[:1].clone(), stats[1:].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        stats = torch.cat([tmax, tot])
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": float(stats[1]) / max(float(stats[0]), 1e-9), "unit": "tokens/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "new_tokens": float(stats[1]),
                          "scaling": "weak"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def algorithmic_verify_bytes(cfg, gamma, S, wbytes=2, kvbytes=2):
    """B_verify = W_stream*b_w + KV read at context S + KV write of gamma+1 rows + logits."""
    w = cfg.n_params(streamed_only=True) * wbytes
    kv_row = 2 * cfg.num_hidden_layers * cfg.num_key_value_heads * cfg.head_dim * kvbytes
    return w + kv_row * S + kv_row * (gamma + 1) + (gamma + 1) * cfg.vocab_size * 4


def find_checkpoint(name):
    """$SPECDEC_MODEL_DIR/<name> (also the hub-style spellings of the BASELINE pairs) if it is a local HF checkpoint
    directory; None otherwise.  Nothing is ever fetched."""
    root = os.environ.get("SPECDEC_MODEL_DIR")
    if not root:
        return None
    alts = [name, name.replace("llama-2-", "Llama-2-"), name.replace("llama-2-", "Llama-2-") + "-hf",
            os.path.join("JackFram", name), os.path.join("meta-llama", name.replace("llama-2-", "Llama-2-") + "-hf"),
            os.path.join("facebook", name)]
    for a in alts:
        d = os.path.join(root, a)
        if os.path.isfile(os.path.join(d, "config.json")):
            return d
    return None


def prompt_for(stream, V, L):
    g = torch.Generator().manual_seed(1000 + stream)
    return torch.randint(3, V, (1, L), generator=g)


def host_cpu_share():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box exposes
    256 hardware threads but grants a 16-core share per GPU; oversubscribing torch's pool stalls it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    env = os.environ.get("SPECDEC_CPU_THREADS")
    if env:
        n = int(env)
    return max(1, min(n, 16 if not env else n))


def cpu_baseline(args, dcfg, tcfg, dm, tm):
    """The oracle (torch-CPU restatement, pinned to the reference by tests/golden) on the host cores,
    same weights, same dtype, a bounded sample of the same workload: same prompt length, fewer new tokens.
    Runs grow (max_len 1, then as many as the time budget allows) so a slow host still yields a number."""
    import oracle
    ncores = host_cpu_share()
    torch.set_num_threads(ncores)
    t0 = time.time()
    dsd = {n: dm._synth_get(n).cpu().to(torch.bfloat16) for n in dm._synth_names}      # (a checkpoint may be stored in fp16)
    tsd = {n: tm._synth_get(n).cpu().to(torch.bfloat16) for n in tm._synth_names}
    for sd in (dsd, tsd):
        if "model.decoder.embed_tokens.weight" in sd:
            sd["lm_head.weight"] = sd["model.decoder.embed_tokens.weight"]
    t_copy = time.time() - t0
    print(f"[cpu_baseline] weights on host in {t_copy:.1f} s, {ncores} threads", file=sys.stderr, flush=True)
    od, ot = oracle.RefCausalLM(dcfg, dsd), oracle.RefCausalLM(tcfg, tsd)
    prompt = prompt_for(0, tcfg.vocab_size, args.cpu_prompt_len)
    best = None
    n_tok = 1
    budget = float(args.cpu_budget_s)
    spent = 0.0
    while True:
        torch.manual_seed(2000)
        w0, p0 = time.time(), time.process_time()
        out, d = oracle.speculative_sampling(prompt, od, ot, 2, None, n_tok, gamma=args.gamma,
                                             top_k=args.top_k, top_p=args.top_p, details=True)
        wall, cpu = time.time() - w0, time.process_time() - p0
        spent += wall
        new = int(out.shape[1] - prompt.shape[1])
        iters = d["target_call_times"]
        print(f"[cpu_baseline] max_len {n_tok}: {new} tokens, {iters} iterations, {wall:.1f} s wall", file=sys.stderr, flush=True)
        run = dict(n_tok=n_tok, new=new, iters=iters, wall=wall, cpu=cpu, acc=d["acc_len"])
        if best is not None and iters > best["iters"]:
            per_iter = (wall - best["wall"]) / max(1, iters - best["iters"])
        else:
            per_iter = None
        run["per_iter"] = per_iter
        best = run
        if per_iter is None:
            nxt = n_tok + 2
            est = wall * 1.3
        else:
            room = budget - spent
            more = int(max(0.0, room - (wall - per_iter * iters)) / max(per_iter, 1e-3))
            nxt = min(args.cpu_max_len, more)
            est = (wall - per_iter * iters) + per_iter * nxt
        if nxt <= n_tok or spent + est > budget or n_tok >= args.cpu_max_len:
            break
        n_tok = nxt
    r = best
    per_iter_txt = f"{r['per_iter']:.2f} s per draft+verify iteration after the prefill; " if r["per_iter"] else ""
    return {
        "value": r["new"] / r["wall"], "unit": "tokens/s", "cores": ncores, "kind": "port",
        "sample": (f"oracle.speculative_sampling (torch-CPU restatement), the same bf16 weights as the GPU run, "
                   f"prompt {args.cpu_prompt_len}, max_len {r['n_tok']} (the GPU run uses max_len {args.max_len}), gamma "
                   f"{args.gamma}: {r['new']} tokens in {r['wall']:.1f} s wall / {r['cpu']:.1f} s process_time over "
                   f"{r['iters']} iterations incl. both prefills; {per_iter_txt}weight copy {t_copy:.1f} s not timed"),
        "wall_s": r["wall"], "process_time_s": r["cpu"], "iterations": r["iters"], "max_len": r["n_tok"],
        "s_per_iteration": r["per_iter"],
        "mean_accept_len": float(np.mean(r["acc"])) if r["acc"] else 0.0,
    }


def acceptance_sweep(args, dcfg, tcfg, max_pos):
    """The headline workload (same architectures, shapes, prompt length, max_len, gamma, sampling parameters, native
    device-RNG loop) on the acceptance-dial pair: one 13b-shaped target, one 68m-shaped draft per sigma.  Reports
    accepted tokens/s and the mean accept length per point; the all-accept branch (rollback(n+2), 2-row draft step,
    bonus sample) and partial accepts are inside these timed regions."""
    from llmspeculativesampling_amd.engine import SpecDecModel
    from llmspeculativesampling_amd.noise import DeviceNoise
    from llmspeculativesampling_amd.sampling import speculative_sampling
    from llmspeculativesampling_amd.synth import dial_draft_transform, dial_target_transform
    base = SpecDecModel.synthetic(dcfg, seed=11, dtype=torch.bfloat16, max_pos=max_pos, transform=dial_draft_transform(0.0, 11))
    tgt = SpecDecModel.synthetic(tcfg, seed=12, dtype=torch.bfloat16, max_pos=max_pos,
                                 transform=dial_target_transform(base._synth_get, dcfg.hidden_size, tcfg.hidden_size))
    points = []
    for sg in [float(x) for x in args.sweep_sigmas.split(",") if x.strip() != ""]:
        drf = base if sg == 0.0 else SpecDecModel.synthetic(dcfg, seed=11, dtype=torch.bfloat16, max_pos=max_pos,
                                                            transform=dial_draft_transform(sg, 11))
        tot_new = tot_acc = tot_it = 0
        logs = {"draft_ms": [], "target": []}
        t_el = 0.0
        for i in range(1 + args.sweep_steps):                 # first call = warmup
            prompt = prompt_for(500 + i, tcfg.vocab_size, args.prompt_len).cuda()
            torch.cuda.synchronize()
            t0 = time.time()
            out, d = speculative_sampling(prompt, drf, tgt, eos_token_id=-1, pad_token_id=None, max_len=args.max_len,
                                          gamma=args.gamma, top_k=args.top_k, top_p=args.top_p, details=True,
                                          rng=DeviceNoise(seed=3000 + i), _event_logs=logs if i else None)
            torch.cuda.synchronize()
            if i:
                t_el += time.time() - t0
                tot_new += int(out.shape[1]) - args.prompt_len
                tot_acc += int(sum(d["acc_len"]))
                tot_it += int(d["target_call_times"])
        ver = [ms for (ms, n_new, _) in logs["target"] if n_new <= args.gamma + 1]
        points.append({"sigma": sg, "value": tot_new / t_el, "unit": "tokens/s", "mean_accept_len": tot_acc / max(1, tot_it),
                       "iterations": tot_it, "new_tokens": tot_new, "ms_per_iteration": t_el / max(1, tot_it) * 1e3,
                       "verify_avg_ms": float(np.mean(ver)) if ver else None,
                       "draft_phase_avg_ms": float(np.mean([m for m in logs["draft_ms"] if m < 50.0])) if logs["draft_ms"] else None})
        if drf is not base:
            del drf
    del tgt, base
    torch.cuda.empty_cache()
    return {"pair": "acceptance-dial pair (synth.py): llama-68m-shaped draft with sigma * noise on its lm_head, "
                    "Llama-2-13b-shaped target carrying the sigma=0 draft's embedding / head in its first 768 hidden dims; "
                    "all weights dense random tensors, all 25.7 GB streamed per verify",
            "steps_per_point": args.sweep_steps, "eos": "disabled (eos_token_id=-1) so every point decodes max_len tokens",
            "points": points}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    # N > 1 without a launcher's environment: this process becomes the launcher (no GPU call before or in it)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a mislabelled n_gpus", file=sys.stderr)
        return 2
    if os.environ.get("SPECDEC_BENCH_STUB"):
        return stub_worker(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # rehearsal on a one-GPU box: SPECDEC_DIST_BACKEND=gloo puts every rank on cuda:0 and the collectives on the host
    backend = os.environ.get("SPECDEC_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > 1 and torch.cuda.device_count() < world:
        print(f"[bench] {world} ranks need {world} GPUs, {torch.cuda.device_count()} visible", file=sys.stderr)
        return 2
    torch.cuda.set_device(local if backend == "nccl" else 0)
    comm_dev = "cuda" if backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _stdout_to_stderr():
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            dist.barrier()                             # (the first collective: connections are made here)

    from llmspeculativesampling_amd.config import load_config
    from llmspeculativesampling_amd.engine import SpecDecModel
    from llmspeculativesampling_amd.noise import DeviceNoise, HostTorchNoise
    from llmspeculativesampling_amd.sampling import speculative_sampling, speculative_sampling_batch

    dcfg, tcfg = load_config(args.draft), load_config(args.target)
    c3_prompts = None
    if args.prompt_lens == "synthetic-c3":
        from llmspeculativesampling_amd.harness import synthetic_prompts
        c3_prompts = synthetic_prompts(100, tcfg.vocab_size, seed=5)
    max_plen = max(int(p.shape[1]) for p in c3_prompts) if c3_prompts else args.prompt_len
    max_pos = max_plen + args.max_len + args.gamma + 8

    def prompt_of(stream):
        return c3_prompts[stream % len(c3_prompts)] if c3_prompts else prompt_for(stream, tcfg.vocab_size, args.prompt_len)
    t0 = time.time()
    TP = args.tp
    # real checkpoints from a LOCAL directory when SPECDEC_MODEL_DIR holds them (never the hub), else
    # random-init from the committed config JSONs (accept-len ~0 by construction; `data` says which)
    ckpt_d, ckpt_t = find_checkpoint(args.draft), find_checkpoint(args.target)
    use_ckpt = bool(ckpt_d and ckpt_t and TP == 1)
    if use_ckpt:
        dm = SpecDecModel.from_pretrained_dir(ckpt_d, dtype=torch.bfloat16, max_pos=max_pos)
        dcfg = dm.cfg
    else:
        dm = SpecDecModel.synthetic(dcfg, seed=1, dtype=torch.bfloat16, max_pos=max_pos)
    if TP > 1:
        # config 5: the target is ONE model sharded over all ranks (two RCCL all-reduces per layer); every rank runs the
        # same stream with the same RNG seed, the draft is replicated
        assert TP == world, f"--tp {TP} needs --gpus {TP}"
        from llmspeculativesampling_amd import tp as tpmod

        def bcast(b):
            box = [b]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        group = tpmod.TPGroup.rccl(rank, world, bcast)
        tm = tpmod.synthetic_shard(tcfg, rank, world, seed=2, group=group, dtype=torch.bfloat16, max_pos=max_pos)
    elif use_ckpt:
        tm = SpecDecModel.from_pretrained_dir(ckpt_t, dtype=torch.bfloat16, max_pos=max_pos)
        tcfg = tm.cfg
    else:
        tm = SpecDecModel.synthetic(tcfg, seed=2, dtype=torch.bfloat16, max_pos=max_pos)
    if args.kv_dtype == "fp8":
        tm.kv_dtype = "fp8"
    torch.cuda.synchronize()
    t_build = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    BS = args.batch_streams

    def run_step(stream, logs=None):
        if BS > 1:
            prompts = [prompt_of(stream * BS + j).cuda() for j in range(BS)]
            outs, ds = speculative_sampling_batch(prompts, dm, tm, eos_token_id=2, pad_token_id=None, max_len=args.max_len,
                                                  gamma=args.gamma, top_k=args.top_k, top_p=args.top_p, details=True,
                                                  seeds=[2000 + stream * BS + j for j in range(BS)],
                                                  _timing=logs if isinstance(logs, dict) else None)
            d = {"acc_len": [a for x in ds for a in x["acc_len"]], "target_call_times": sum(x["target_call_times"] for x in ds)}
            new = sum(int(o.shape[1]) - int(p.shape[1]) for o, p in zip(outs, prompts))
            return outs, d, new
        prompt = prompt_of(stream).cuda()
        if args.rng == "device":
            nz = DeviceNoise(seed=2000 + stream)
        else:
            torch.manual_seed(2000 + stream)
            nz = HostTorchNoise(prompt.device)
        out, d = speculative_sampling(prompt, dm, tm, eos_token_id=2, pad_token_id=None, max_len=args.max_len,
                                      gamma=args.gamma, top_k=args.top_k, top_p=args.top_p, details=True, rng=nz,
                                      _event_logs=logs)
        return [out], d, int(out.shape[1]) - int(prompt.shape[1])

    # streams: rank r takes s = r, r+world, ... (round-robin)
    srank, sworld = (0, 1) if TP > 1 else (rank, world)      # tensor parallel: one stream per step for the whole group
    for i in range(args.warmup):
        run_step(srank + 10_000 * (i + 1))
    if dist is not None and TP == 1:
        # set-up, not work: the gather's communicator (ncclCommInitRank takes from a few hundred ms to seconds) and RCCL's
        # first-collective warm-up belong in front of the timed region, whatever --warmup is
        from llmspeculativesampling_amd.dist import gather_streams
        gather_streams([torch.zeros((1, 4), dtype=torch.int64, device=comm_dev)], world, 8, device=comm_dev)
    barrier()
    t0 = time.time()
    new_tokens, acc_sum, n_iters = 0, 0, 0
    logs = {"draft_ms": [], "target": []} if args.rng == "device" else ([], [])
    outs = []
    for i in range(args.steps):
        o_list, d, n_new_tok = run_step(srank + i * sworld, logs)
        new_tokens += n_new_tok
        acc_sum += int(sum(d["acc_len"]))
        n_iters += int(d["target_call_times"])
        outs.extend(o_list)
    if dist is not None and TP == 1:
        # throughput-mode gather of the generated ids (KB-scale; the only collective on the path)
        from llmspeculativesampling_amd.dist import gather_streams
        width = max_plen + args.max_len + args.gamma + 1
        all_streams = gather_streams(outs, args.steps * world * BS, width, device=comm_dev)
        assert len(all_streams) == args.steps * world * BS
    barrier()
    elapsed = time.time() - t0
    stats = torch.tensor([elapsed, float(new_tokens), float(acc_sum), float(n_iters)], dtype=torch.float64, device=comm_dev)
    if dist is not None:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = stats[1:].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        if TP > 1:
            tot /= world                              # every rank decoded the same streams: count them once
        elapsed, (new_tokens, acc_sum, n_iters) = float(tmax[0]), [float(x) for x in tot]
    value = new_tokens / elapsed

    # ---- verify-step roofline from the HIP events recorded inside the timed region (torch's current stream is
    # the stream every kernel was launched on).  One "launch" = one verify step = one target forward over
    # gamma+1 rows + norm_probs, a fixed chain of kernels; algorithmic bytes per SURVEY.md 8(d).
    ver_ms, ver_S, pre_ms = [], [], []
    if isinstance(logs, dict) and BS > 1:
        for (e0, e1, nstr, ctx) in logs.get("verify", []):
            if nstr == BS:
                ver_ms.append(e0.elapsed_time(e1))
                ver_S.append(ctx)
        drf_ms = []
    elif isinstance(logs, dict):
        for (ms, n_new, upto) in logs["target"]:
            (ver_ms if n_new == args.gamma + 1 else pre_ms).append(ms)
            if n_new == args.gamma + 1:
                ver_S.append(upto)
        # an iteration's draft phase = gamma draft steps; the first iteration of a call also carries the draft's prefill
        # (its target entry feeds more than gamma+1 rows): left out, like the target prefill is left out of the verify mean
        drf_ms = [ms / args.gamma for ms, (_, n_new, _) in zip(logs["draft_ms"], logs["target"]) if n_new == args.gamma + 1]
    else:
        for (e0, e1, n_new, upto) in logs[1]:
            (ver_ms if n_new == args.gamma + 1 else pre_ms).append(e0.elapsed_time(e1))
            if n_new == args.gamma + 1:
                ver_S.append(upto)
        drf_ms = [e0.elapsed_time(e1) for (e0, e1, n_new, _) in logs[0] if n_new <= 2]
    t_ver = float(np.mean(ver_ms)) if ver_ms else float("nan")
    S_mean = float(np.mean(ver_S)) if ver_S else float(max_plen + args.max_len / 2)
    b_ver = algorithmic_verify_bytes(tm.cfg if TP > 1 else tcfg, args.gamma, S_mean, kvbytes=1 if args.kv_dtype == "fp8" else 2)
    if BS > 1:       # one pass over the weights serves BS streams; KV and logits scale with the stream count
        w_only = tcfg.n_params(streamed_only=True) * 2
        from llmspeculativesampling_amd.engine import MAX_ROWS_PER_FORWARD
        per_pass = max(1, MAX_ROWS_PER_FORWARD // (args.gamma + 1))          # whole streams per pass (sd_spec_batch_generate)
        b_ver = w_only * ((BS + per_pass - 1) // per_pass) + (b_ver - w_only) * BS
    achieved = b_ver / (t_ver * 1e-3) / 1e9 if ver_ms else float("nan")
    # HBM traffic per verify step from the PMC counters: they need their own rocprofv3 passes (FETCH_SIZE and
    # WRITE_SIZE do not fit one pass and cannot be combined with the timed run), so the committed summary of the
    # latest such run is read here (tools/pmc_traffic.py; gfx950 half-count correction applied, calibrated on a
    # GEMM whose bytes are known).
    traffic = None
    try:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
        if cands and args.target == "llama-2-13b" and args.gamma == 4 and BS == 1:
            traffic = json.load(open(os.path.join(ROOT, "profiles", cands[-1])))["traffic_bytes_per_verify"]
    except Exception:
        traffic = None
    roofline = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
        "kernel": "verify step (target forward over gamma+1 rows: weight-streaming GEMM chain - gemm_bf16_stream and its norm-on-load / residual-epilogue forms - + attention / fused attention+O + norm_probs)",
        "algorithmic_bytes_per_launch": b_ver, "avg_launch_ms": t_ver, "launches_timed": len(ver_ms),
        "mean_context": S_mean,
        "draft_step_avg_ms": float(np.mean(drf_ms)) if drf_ms else None,
        "target_prefill_avg_ms": float(np.mean(pre_ms)) if pre_ms else None,
    }

    # ---- per-op-class split of one verify step (events around every launch; outside the timed region)
    if args.profile_classes and rank == 0 and BS == 1 and TP == 1:
        ses = tm.new_session(max_pos)
        toks = prompt_for(0, tcfg.vocab_size, args.prompt_len + args.gamma + 1).cuda()[0].to(torch.int32)
        done = 0
        while done < args.prompt_len:
            m = min(64, args.prompt_len - done)
            ses.forward(toks[done:done + m], 0)
            done += m
        for _ in range(2):
            ses.rollback(args.prompt_len)
            ses.forward(toks[args.prompt_len:], args.gamma + 1)
        torch.cuda.synchronize()
        ses.profile(True)
        reps = 3
        for _ in range(reps):
            ses.rollback(args.prompt_len)
            ses.forward(toks[args.prompt_len:], args.gamma + 1)
        prof = ses.profile_read()
        ses.profile(False)
        wbytes = tm.weight_bytes
        roofline["op_classes_ms_per_verify"] = {k: v[0] / reps for k, v in prof.items()}
        roofline["op_classes_launches_per_verify"] = {k: v[1] // reps for k, v in prof.items()}
        if "gemm" in prof:
            g_ms = prof["gemm"][0] / reps
            roofline["gemm_kernel"] = {"name": "the weight-streaming GEMM class: gemm_bf16_stream<MT=1,UNROLL=4,EPI,NTW=1> and its norm-on-load / residual-epilogue forms gemm_bf16_stream_xn, gemm_bf16_stream_fin (17-80 rows: gemm_bf16_rows)", "weight_bytes_per_verify": wbytes,
                                       "ms_per_verify": g_ms, "achieved_GBs": wbytes / (g_ms * 1e-3) / 1e9,
                                       "frac": wbytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    result = {
        "metric": "accepted tokens/sec (speculative_sampling, llama-68m -> Llama-2-13b, gamma=4)",
        "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if TP == 1 else "strong",
        "vs_baseline": None, "dtype": "bf16",
        "data": "checkpoint" if use_ckpt else "synthetic", "scaling_note": None if TP == 1 else
        "tensor-parallel target: the ranks decode the SAME stream together (strong scaling of one stream), per-rank roofline",
        "config": {"workload": f"{args.draft} -> {args.target}, gamma={args.gamma}, bf16, prompt " +
                               (f"{args.prompt_len}" if not c3_prompts else "lengths U{32..512} (100 synthetic prompts, seed 5: SURVEY 8(d) C3)") + ", "
                               f"max_len {args.max_len}, top_k {args.top_k}, top_p {args.top_p}, {BS} stream(s) per step per GPU"
                               f"{' decoded in lockstep through shared weight passes' if BS > 1 else ''}, "
                               f"rng={args.rng}; " + (f"local checkpoints {ckpt_d} -> {ckpt_t}, synthetic prompts" if use_ckpt else
                                                      "random-init weights (accept-len ~0 by construction)") +
                               ("; mode: throughput (stream-batched verify)" if BS > 1 else "; mode: single stream (BASELINE configs[1])"),
                   "streams": args.steps * (1 if TP > 1 else world) * BS,
                   "parallelism": (f"streams sharded over {world} GPU(s), no data-path collective" if TP == 1 else
                                   f"target tensor-parallel over {TP} GPUs (2 RCCL all-reduces per layer), draft replicated"),
                   "kv_dtype": args.kv_dtype},
        "mean_accept_len": acc_sum / max(1.0, n_iters), "iterations": n_iters, "new_tokens": new_tokens,
        "roofline": roofline, "model_build_s": t_build,
    }

    if (args.accept_sweep and rank == 0 and world == 1 and BS == 1 and args.kv_dtype == "model" and dcfg.arch == "llama"
            and tcfg.arch == "llama" and not use_ckpt):
        try:
            result["acceptance_sweep"] = acceptance_sweep(args, dcfg, tcfg, max_pos)
        except Exception as e:
            result["acceptance_sweep"] = {"error": f"{type(e).__name__}: {e}"}

    if args.cpu_baseline and rank == 0 and world == 1:
        try:
            cb = cpu_baseline(args, dcfg, tcfg, dm, tm)
            # the same bounded workload (same prompt, same max_len) on the GPU, so the two numbers are on one workload
            n_tok = int(cb.get("max_len", args.cpu_max_len))
            prompt = prompt_for(0, tcfg.vocab_size, args.cpu_prompt_len).cuda()
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                w0 = time.time()
                out, d = speculative_sampling(prompt, dm, tm, eos_token_id=2, pad_token_id=None, max_len=n_tok,
                                              gamma=args.gamma, top_k=args.top_k, top_p=args.top_p, details=True,
                                              rng=DeviceNoise(seed=2000))
                torch.cuda.synchronize()
                wall = time.time() - w0
                if best is None or wall < best[0]:
                    best = (wall, int(out.shape[1]) - args.cpu_prompt_len, d["target_call_times"])
            cb["gpu_same_workload"] = {"value": best[1] / best[0], "unit": "tokens/s", "wall_s": best[0], "new_tokens": best[1],
                                       "iterations": best[2], "max_len": n_tok, "prompt_len": args.cpu_prompt_len,
                                       "note": "same prompt / max_len / gamma / top_k / top_p as cpu_baseline, both prefills included; "
                                               "device Philox RNG (the CPU run draws from torch's generator)"}
            if cb.get("value"):
                cb["gpu_over_cpu_same_workload"] = cb["gpu_same_workload"]["value"] / cb["value"]
            result["cpu_baseline"] = cb
        except Exception as e:          # never lose the GPU numbers to a host-side failure
            result["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": None, "kind": "port",
                                      "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
