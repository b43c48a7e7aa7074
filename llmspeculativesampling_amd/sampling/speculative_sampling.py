"""Drop-in for reference sampling/speculative_sampling.py:1876-2076 (``speculative_sampling``).

Same signature, return values, ``details`` keys, RNG draw order, EOS rule and exceptions.  The
loop body differs in mechanism only: tokens, both KV caches and both probability histories stay
on the device; each iteration is gamma x (draft forward, norm_probs, sample), one target forward
over gamma+1 rows, norm_probs, then one accept-scan + resample launch pair and a single
device->host read of the 208-byte result block.
"""
from __future__ import annotations

import ctypes as C
from time import process_time_ns

import numpy as np
import torch

from .._lib import lib, check, SdAcceptResult
from ..engine import as_specdec_model, _stream, check_token_ids, same_device
from ..noise import DeviceNoise, HostTorchNoise
from .kvcache_model import KVCacheModel


def _make_noise(rng, device):
    if rng is None or rng == "host":
        return HostTorchNoise(device)
    if rng == "device":
        return DeviceNoise(seed=int(torch.initial_seed()))
    return rng


@torch.no_grad()
def speculative_sampling(prefix: torch.Tensor, approx_model, target_model, eos_token_id, pad_token_id,
                         max_len: int, gamma: int = 4, temperature: float = 1, top_k: int = 0, top_p: float = 0,
                         verbose: bool = False, random_seed: int = None, details: bool = False, *, rng=None,
                         _event_logs=None):
    """reference speculative_sampling.py:1876-2076.

    ``rng`` (extension): "host" (default) draws every variate from torch's CPU generator in the
    reference's order, so token ids equal the reference's CPU path under the same seed; "device"
    uses on-device Philox (throughput mode); or a noise object from ``llmspeculativesampling_amd.noise``.
    """
    assert prefix.shape[0] == 1, "input batch size must be 1"
    draft_m, target_m = as_specdec_model(approx_model), as_specdec_model(target_model)
    if draft_m.cfg.is_encoder_decoder or target_m.cfg.is_encoder_decoder:
        raise NotImplementedError("encoder-decoder models are out of scope")
    same_device(draft_m, target_m)
    dev = target_m.device
    V = target_m.cfg.vocab_size
    assert draft_m.cfg.vocab_size == V, "draft and target must share a vocabulary"
    assert 1 <= gamma <= 16

    seq_len0 = prefix.shape[1]
    T = seq_len0 + max_len
    cap = T + gamma + 2
    host_seq = [int(t) for t in prefix[0].tolist()]
    if seq_len0 < T:
        try:                                          # nn.Embedding's IndexError surfaces inside the reference's
            check_token_ids(host_seq, V)              # try block and leaves as RuntimeError('s') (:1933, :2044-2046)
        except IndexError as e:
            print(e)
            raise RuntimeError("s") from e
    ori_eos_cnt = sum(1 for t in host_seq if t == eos_token_id)
    noise = _make_noise(rng, dev)

    draft = KVCacheModel(draft_m, temperature, top_k, top_p, max_seq=cap, noise=noise, full_history=False)
    target = KVCacheModel(target_m, temperature, top_k, top_p, max_seq=cap, noise=noise, full_history=False)
    draft._ensure(cap)
    target._ensure(cap)
    if _event_logs is not None:                       # bench.py: HIP-event brackets around every forward
        draft.event_log, target.event_log = _event_logs
    seq32 = torch.zeros(cap + 1, dtype=torch.int32, device=dev)
    seq32[:seq_len0] = prefix[0].to(device=dev, dtype=torch.int32)
    res_dev = torch.zeros(C.sizeof(SdAcceptResult), dtype=torch.uint8, device=dev)
    res_host = torch.zeros(C.sizeof(SdAcceptResult), dtype=torch.uint8).pin_memory()
    tok_host = torch.zeros(gamma + 2, dtype=torch.int32).pin_memory()
    serr = torch.zeros(gamma + 2, dtype=torch.int32, device=dev)
    q_hist, p_hist = draft._probs, target._probs
    ld = q_hist.stride(0)
    st = _stream()
    # p - q, max_fn and the residual draw run in the rows' dtype when both models keep 16-bit rows (OPT)
    res_mode = target_m.norm_mode if target_m.norm_mode == draft_m.norm_mode else 0
    res_dtype = target_m.probs_dtype if res_mode else torch.float32

    if getattr(noise, "on_device", False) and not verbose:
        return _native_device_loop(prefix, draft, target, seq32, host_seq, ori_eos_cnt, eos_token_id, T, gamma,
                                   temperature, top_k, top_p, random_seed, details, noise, res_dev, res_host, tok_host,
                                   _event_logs)

    approx_time = target_time = other_time = 0
    approx_calls = target_calls = 0
    acc_rate, acc_len = [], []
    out_tokens = host_seq
    r_const = None
    try:
        while len(host_seq) < T:
            tick = process_time_ns()
            L = len(host_seq)
            # ---- draft: gamma steps, tokens never leave the device (kvcache_model.py:279-293)
            for i in range(gamma):
                draft.forward_sample(seq32, L + i, noise, serr[i])
            approx_calls += 1
            approx_time += process_time_ns() - tick
            tick = process_time_ns()
            # ---- target: one forward over the gamma+1 uncached rows (all rows on the first call)
            n_new = L + gamma - target.cache_len
            target.forward_rows(seq32, L + gamma, min(n_new, gamma + 1))
            # generate(x, 1) samples from the last row and throws the token away (kvcache_model.py:283):
            # the draw is part of the RNG contract even though its result is unused
            if noise.on_device:
                noise.next_draws(1)
            else:
                noise.skip_exponential(V, target_m.probs_dtype)
            target_calls += 1
            target_time += process_time_ns() - tick
            tick = process_time_ns()
            # ---- accept scan (speculative_sampling.py:1964-1991) + residual / bonus sample (:2005-2023)
            if noise.on_device:
                if random_seed:
                    # reseed-before-every-r quirk (:1976-1977): all r are one and the same draw and the
                    # stream restarts after it, whatever was accepted
                    noise.reseed(random_seed)
                    if r_const is None:
                        g = torch.Generator().manual_seed(int(random_seed))
                        r_const = torch.rand(1, generator=g).repeat(gamma).to(dev)
                    check(lib.sd_accept_scan(p_hist.data_ptr(), q_hist.data_ptr(), ld, seq32.data_ptr(), L, gamma,
                                             r_const.data_ptr(), 0, 0, res_dev.data_ptr(), st), "sd_accept_scan")
                else:
                    check(lib.sd_accept_scan(p_hist.data_ptr(), q_hist.data_ptr(), ld, seq32.data_ptr(), L, gamma,
                                             None, noise.seed, noise.next_draws(gamma), res_dev.data_ptr(), st),
                          "sd_accept_scan")
                check(lib.sd_resample(p_hist.data_ptr(), q_hist.data_ptr(), ld, V, seq32.data_ptr(), L, gamma, None,
                                      noise.seed, noise.next_draws(1), res_dev.data_ptr(), None, res_mode, st), "sd_resample")
            else:
                r, token = noise.uniforms(gamma, random_seed)
                check(lib.sd_accept_scan(p_hist.data_ptr(), q_hist.data_ptr(), ld, seq32.data_ptr(), L, gamma,
                                         r.data_ptr(), 0, 0, res_dev.data_ptr(), st), "sd_accept_scan")
                if token is not None:
                    # the reference stops drawing uniforms at the first reject: learn l, re-align the host stream
                    res_host.copy_(res_dev, non_blocking=True)
                    torch.cuda.current_stream().synchronize()
                    l_now = SdAcceptResult.from_buffer_copy(res_host.numpy().tobytes()).n_accepted
                    noise.realign(token, min(l_now + 1, gamma))
                e = noise.exponential(V, res_dtype)
                check(lib.sd_resample(p_hist.data_ptr(), q_hist.data_ptr(), ld, V, seq32.data_ptr(), L, gamma,
                                      e.data_ptr(), 0, 0, res_dev.data_ptr(), None, res_mode, st), "sd_resample")
            res_host.copy_(res_dev, non_blocking=True)
            tok_host.copy_(seq32[L:L + gamma + 2], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            res = SdAcceptResult.from_buffer_copy(res_host.numpy().tobytes())
            if bool(serr[:gamma].any()) or (res.flags & 2):
                raise RuntimeError("prob error")
            draft.check_errors(L - 1, L + gamma - 1)
            target.check_errors(L - 1, L + gamma)
            l, n, t = res.n_accepted, res.n, res.next_token
            for i in range(gamma):                    # statistic over all gamma drafted positions (:1966-1971)
                acc_rate.append(min(1.0, float(res.p_at[i]) / float(res.q_at[i])))
            acc_len.append(l)
            assert n >= L - 1, f"n {n}, prefix_len {L}"
            drafted = tok_host[:l].tolist()
            host_seq = host_seq + drafted + [t]
            draft.rollback(n + 1)
            target.rollback(n + 1 if l < gamma else n + 2)
            out_tokens = host_seq
            if verbose:
                print(f"accepted {l} of {gamma}: {drafted} + {t}")
            # ---- EOS rule over the whole sequence (:2033-2041)
            eos_total = sum(1 for x in host_seq if x == eos_token_id)
            if eos_total > ori_eos_cnt:
                seen, cut = 0, len(host_seq)
                for idx, x in enumerate(host_seq):
                    if x == eos_token_id:
                        seen += 1
                        if seen == ori_eos_cnt + 1:
                            cut = idx + 1
                            break
                out_tokens = host_seq[:cut]
                break
            other_time += process_time_ns() - tick
    except Exception as e:                            # (:2044-2046); the cause stays attached to the traceback
        print(e)
        raise RuntimeError("s") from e

    out = torch.tensor([out_tokens], dtype=torch.int64, device=prefix.device)
    if verbose:
        print(f"generated tokens numbers {len(host_seq) - seq_len0}, acc len {acc_len}")
    if details:
        return out, {
            "approx_time": approx_time, "target_time": target_time, "other_time": other_time,
            "acc_len": acc_len, "acc_rate": np.mean(acc_rate),
            "target_call_times": target_calls, "approx_call_times": approx_calls,
            "target_model_time": target.forward_time_dict["_model_time"],
            "target_pre_cache_time": target.forward_time_dict["prepare_cache_time"],
            "target_post_prob_time": target.forward_time_dict["norm_prob_time"],
        }
    return out


def _native_device_loop(prefix, draft, target, seq32, host_seq, ori_eos_cnt, eos_token_id, T, gamma, temperature,
                        top_k, top_p, random_seed, details, noise, res_dev, res_host, tok_host, timing_log):
    """Device-RNG mode: every iteration is ONE call into libspecdec (sd_spec_iteration) that enqueues the gamma draft
    steps, the target forward, the accept scan and the resample; the host waits on the stream once per iteration and
    reads 208 + 4*(gamma+2) bytes from pinned memory.  Same loop semantics as the Python-orchestrated path above
    (reference speculative_sampling.py:1934-2046)."""
    dev = target._model.device
    V = target._model.cfg.vocab_size
    seq_len0 = prefix.shape[1]
    err_words = torch.zeros(3 * gamma + 1, dtype=torch.int32, device=dev)
    sp = C.c_void_p()
    check(lib.sd_spec_create(draft._session.handle, target._session.handle, gamma, float(temperature), int(top_k or 0),
                             float(top_p or 0.0), seq32.data_ptr(), draft._probs.data_ptr(), target._probs.data_ptr(),
                             draft._probs.stride(0), draft._session.logits.data_ptr(), draft._session.logits.stride(0),
                             target._session.logits.data_ptr(), target._session.logits.stride(0), err_words.data_ptr(),
                             res_dev.data_ptr(), target._norm_ws.data_ptr(), C.byref(sp)), "sd_spec_create")
    timed = timing_log is not None or details          # HIP-event brackets around the draft and the verify phase
    if timed:
        check(lib.sd_spec_timing(sp, 1), "sd_spec_timing")
    r_const = None
    if random_seed:
        g = torch.Generator().manual_seed(int(random_seed))
        r_const = torch.rand(1, generator=g).repeat(gamma).to(dev)
    st = _stream()
    approx_time = target_time = other_time = 0
    calls = 0
    acc_rate, acc_len = [], []
    draft_len = target_len = 0
    # the loop itself runs inside libspecdec (sd_spec_generate): one call, one stream wait per iteration in native code,
    # the interpreter only sees the finished sequence and the per-iteration statistics
    cap = T + gamma + 2
    seq_host = np.zeros(cap, dtype=np.int32)
    seq_host[:len(host_seq)] = host_seq
    max_iters = max(1, T - len(host_seq))
    acc_arr = np.zeros(max_iters, dtype=np.int32)
    p_arr = np.zeros(max_iters * gamma, dtype=np.float32)
    q_arr = np.ones(max_iters * gamma, dtype=np.float32)
    dms_arr = np.zeros(max_iters, dtype=np.float32)
    tms_arr = np.zeros(max_iters, dtype=np.float32)
    c_len, c_dl, c_tl = C.c_int(len(host_seq)), C.c_int(0), C.c_int(0)
    c_seed, c_draw = C.c_uint64(noise.seed), C.c_uint64(noise.draw)
    c_iters, c_err = C.c_int(0), C.c_int(0)
    out_tokens = host_seq
    try:
        tick = process_time_ns()
        check(lib.sd_spec_generate(sp, seq_host.ctypes.data, C.byref(c_len), T, int(eos_token_id), int(ori_eos_cnt),
                                   C.byref(c_seed), C.byref(c_draw), int(random_seed or 0),
                                   r_const.data_ptr() if r_const is not None else None, C.byref(c_dl), C.byref(c_tl),
                                   res_host.data_ptr(), max_iters, acc_arr.ctypes.data, p_arr.ctypes.data, q_arr.ctypes.data,
                                   dms_arr.ctypes.data if timed else None, tms_arr.ctypes.data if timed else None,
                                   C.byref(c_iters), C.byref(c_err), st), "sd_spec_generate")
        other_time += process_time_ns() - tick           # host CPU time of the loop (enqueue + waits + bookkeeping)
        noise.seed, noise.draw = c_seed.value, c_draw.value
        if c_err.value == 1:
            raise RuntimeError("prob error")
        if c_err.value == 2:
            raise RuntimeError("norm logits error")
        calls = c_iters.value
        draft_len, target_len = c_dl.value, c_tl.value
        host_seq = seq_host[:c_len.value].tolist()
        acc_len = acc_arr[:calls].tolist()
        # python double ratio of the two float32 values, as the reference's .item() division (:1966-1971)
        acc_rate = np.minimum(1.0, p_arr[:calls * gamma].astype(np.float64) / q_arr[:calls * gamma].astype(np.float64)).tolist()
        if timed:
            # the reference's approx_time / target_time are host process_time deltas around generate() (:1937-1962);
            # here the host only enqueues, so the device time of the two phases (HIP events) is what is reported
            approx_time = int(sum(int(v * 1e6) for v in dms_arr[:calls]))
            target_time = int(sum(int(v * 1e6) for v in tms_arr[:calls]))
            if timing_log is not None:
                Lc, tl = seq_len0, 0
                for i in range(calls):
                    timing_log["draft_ms"].append(float(dms_arr[i]))
                    timing_log["target"].append((float(tms_arr[i]), Lc + gamma - tl, Lc + gamma))
                    tl = Lc + int(acc_arr[i])                    # n + 1 with n = L + l - 1
                    Lc += int(acc_arr[i]) + 1
        out_tokens = host_seq
        if sum(1 for x in host_seq if x == eos_token_id) > ori_eos_cnt:
            seen, cut = 0, len(host_seq)
            for idx, x in enumerate(host_seq):
                if x == eos_token_id:
                    seen += 1
                    if seen == ori_eos_cnt + 1:
                        cut = idx + 1
                        break
            out_tokens = host_seq[:cut]
    except Exception as e:
        print(e)
        lib.sd_spec_destroy(sp)
        raise RuntimeError("s") from e
    lib.sd_spec_destroy(sp)
    draft._session.cache_len, target._session.cache_len = draft_len, target_len
    out = torch.tensor([out_tokens], dtype=torch.int64, device=prefix.device)
    if details:
        return out, {
            "approx_time": approx_time, "target_time": target_time, "other_time": other_time,
            "acc_len": acc_len, "acc_rate": np.mean(acc_rate),
            "target_call_times": calls, "approx_call_times": calls,
            # the verify phase is one fused launch chain (forward + norm_probs, no cache preparation): all of it is model time
            "target_model_time": target_time, "target_pre_cache_time": 0, "target_post_prob_time": 0,
        }
    return out
