#!/usr/bin/env python3
"""Where the fused attention + O-projection launch (fused_kernels.h) spends its time, PER WORKGROUP: every workgroup of a
stamped launch (SD_AO_STAMPS=1, set here) leaves wall_clock64 at its milestones plus XCC_ID / HW_ID, so a late attention
workgroup can be tied to its head, XCD, shader engine and CU.  13b layer shape, `rows` rows after a
190-token prefix; `--layers L` runs L layers back to back and reports the LAST layer's launch (warm pipeline, as in the
bench) instead of a cold single layer."""
import argparse
import ctypes as C
import os
import sys

os.environ["SD_AO_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from llmspeculativesampling_amd import _lib, engine
from llmspeculativesampling_amd.config import ModelConfig

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=1)
ap.add_argument("--rows", type=int, default=5)
ap.add_argument("--prefix", type=int, default=190)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--flush", type=int, default=1, help="stream 1 GiB through the caches before every repetition")
ap.add_argument("--prefetch", type=int, default=0, help="after the flush, read the whole KV arena once (torch reduction) - is a K / V "
                "slice that was touched before the layer's QKV GEMM still close (Infinity Cache) when attention runs?")
args = ap.parse_args()

cfg = ModelConfig(arch="llama", vocab_size=32000, hidden_size=5120, intermediate_size=13824, num_hidden_layers=args.layers,
                  num_attention_heads=40, num_key_value_heads=40, max_position_embeddings=512, rms_norm_eps=1e-5)
m = engine.SpecDecModel.synthetic(cfg, seed=9, dtype=torch.bfloat16, max_pos=400)
ids = torch.from_numpy(np.random.default_rng(4).integers(3, cfg.vocab_size, size=(400,))).to(torch.int32).cuda()
ses = m.new_session(400)
ses.forward(ids[:args.prefix], 0)
junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
n_groups = (args.rows + 7) // 8
n_att = 40 * n_groups
n_wgs = n_att + (5120 // 16 + 1) // 2
for rep in range(args.reps):
    if args.flush:
        junk.add_(1.0)
    ses.rollback(args.prefix)
    if args.prefetch:
        ses.kv.view(torch.int32).sum()
    ses.forward(ids[args.prefix:args.prefix + args.rows], args.rows)
    torch.cuda.synchronize()
    st = (C.c_longlong * (8 * n_wgs))()
    _lib.check(_lib.lib.sd_session_ao_stamps(ses.handle, st, n_wgs), "stamps")
    rec = np.array(list(st), dtype=np.int64).reshape(n_wgs, 8)
    t0 = rec[:, 0].min()
    us = (rec[:, :6] - t0) / 100.0
    ids6 = rec[:, 6]
    xcc, hw = ids6 & 0xff, ids6 >> 8
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    att = np.arange(n_wgs) < n_att
    print(f"== rep {rep}: {args.layers} layer(s), {args.rows} rows, {n_att} attention + {n_wgs - n_att} O workgroups; "
          f"launch span {us[:, :5].max():.2f} us")
    order = np.argsort(us[:n_att, 5])
    print("   attention workgroups by arrival: wg head grp | xcc se sh cu | start scores softmax pv stores counted")
    for w in order:
        print(f"   {w:3d} {int(rec[w, 7] >> 8) & 0xffffff:3d} {int(rec[w, 7]) & 0xff:1d} | {xcc[w]:1d} {se[w]:1d} {sh[w]:1d} {cu[w]:2d} | "
              + " ".join(f"{x:6.2f}" for x in us[w, :6]))
    late = us[:n_att, 5] > np.median(us[:n_att, 5]) + 1.5
    for name, key in (("xcc", xcc), ("se", se), ("cu", cu)):
        vals = sorted(set(key[:n_att].tolist()))
        print(f"   late (> median + 1.5 us) per {name}: " + " ".join(f"{v}:{int((late & (key[:n_att] == v)).sum())}/{int((key[:n_att] == v).sum())}" for v in vals))
    # CUs shared between an attention workgroup and an O workgroup (same xcc, se, sh, cu)?
    where = {}
    for w in range(n_wgs):
        where.setdefault((int(xcc[w]), int(se[w]), int(sh[w]), int(cu[w])), []).append(w)
    shared = [v for v in where.values() if len(v) > 1]
    print(f"   CUs holding more than one workgroup of the launch: {len(shared)}" + (f" e.g. {shared[:6]}" if shared else ""))
    o = us[n_att:]
    print(f"   O workgroups: weights landed {np.median(o[:, 1]):.2f} (max {o[:, 1].max():.2f}), counter seen {np.median(o[:, 2]):.2f}, "
          f"MFMAs done {np.median(o[:, 3]):.2f} (max {o[:, 3].max():.2f}), stored {np.median(o[:, 4]):.2f} (max {o[:, 4].max():.2f})")
