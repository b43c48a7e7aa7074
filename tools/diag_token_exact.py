#!/usr/bin/env python3
"""Diagnostic for tests/test_gpu_production_parity.py::test_mfma_path_token_for_token...: where the HIP bf16 path and the
oracle part ways on an exact-GEMM pair, is it a rounding-boundary event or a defect?  Teacher-forces the oracle's token
sequence through both implementations and compares every logit; prints the oracle's accept margins."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from philox_replay import PhiloxOracleNoise
import test_gpu_production_parity as T
from llmspeculativesampling_amd import _lib, engine, noise as N
import llmspeculativesampling_amd.sampling as S

def st(): return torch.cuda.current_stream().cuda_stream

name, frac, gamma, seed = [c for c in T.TOKEN_EXACT_CASES if c[0] == (sys.argv[1] if len(sys.argv) > 1 else "g4_b")][0]
V = 512
from llmspeculativesampling_amd.config import ModelConfig
dcfg = ModelConfig(arch="llama", vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=128, rms_norm_eps=1e-5)
tcfg = ModelConfig(arch="llama", vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=128, rms_norm_eps=1e-5)
dsd = T._fewbit_sparse_sd(dcfg, seed); tsd = T._fewbit_sparse_sd(tcfg, seed + 100)
shared = T._perturb_fewbit(dsd, seed + 200, frac); hd = tcfg.head_dim
for k, v in shared.items():
    if k in tsd and tsd[k].shape == v.shape: tsd[k] = v
    elif k in tsd and k.endswith(("k_proj.weight", "v_proj.weight")): tsd[k] = v[:hd]
prompt = torch.from_numpy(np.random.default_rng(seed).integers(3, V, size=(1, 9)))
kw = dict(gamma=gamma, top_k=20, top_p=0.9)

class Log(PhiloxOracleNoise):
    def __init__(self, *a):
        super().__init__(*a); self.us = []
    def uniform(self):
        r = super().uniform(); self.us.append(float(r)); return r
nz = Log(_lib.lib, 4242 + seed, gamma, st)
od, ot = oracle.RefCausalLM(dcfg, dsd), oracle.RefCausalLM(tcfg, tsd)
want, wd = oracle.speculative_sampling(prompt, od, ot, -1, None, 16, details=True, noise=nz, **kw)
dm = engine.SpecDecModel.from_state_dict(dcfg, dsd, dtype=torch.bfloat16)
tm = engine.SpecDecModel.from_state_dict(tcfg, tsd, dtype=torch.bfloat16)
got, gd = S.speculative_sampling(prompt.cuda(), dm, tm, -1, None, 16, details=True, rng=N.DeviceNoise(4242 + seed), **kw)
w, g = want[0].tolist(), got[0].cpu().tolist()
first = next((i for i, (a, b) in enumerate(zip(w, g)) if a != b), None)
print("oracle acc_len", wd["acc_len"], "hip", gd["acc_len"], "first differing position", first, "uniforms", [round(u, 4) for u in nz.us])
# teacher forcing: the oracle's sequence through both, one full forward and an incremental replay
for label, cfg, sd, m, om in (("draft", dcfg, dsd, dm, od), ("target", tcfg, tsd, tm, ot)):
    ids = want[:, :-1]
    full = om(ids).logits.float()[0]
    ses = m.new_session(64)
    hl = ses.forward(ids[0].to(torch.int32).cuda(), min(64, ids.shape[1])).cpu()
    d = (hl - full[-hl.shape[0]:]).abs()
    nz_ = d > 0
    ulp = (d / (full[-hl.shape[0]:].abs().clamp_min(1e-3) * 2.0 ** -8))
    print(f"{label}: full forward of {ids.shape[1]} rows: {int(nz_.sum())} of {d.numel()} logits differ; max |diff| {float(d.max()):.4g} "
          f"(= {float(ulp[nz_].max()) if nz_.any() else 0:.2f} bf16 ulps); rows with a difference: {sorted(set(nz_.nonzero()[:, 0].tolist()))}")
    # probabilities of those rows
    for r in sorted(set(nz_.nonzero()[:, 0].tolist()))[:4]:
        po = oracle.norm_logits(full[r:r + 1], 1.0, 20, 0.9)[0]
        ph = oracle.norm_logits(hl[r:r + 1], 1.0, 20, 0.9)[0]
        print(f"   row {r}: TV between the two probability rows {0.5 * float((po - ph).abs().sum()):.4f}; "
              f"support differs in {int(((po > 0) != (ph > 0)).sum())} tokens")
