#!/usr/bin/env python3
"""Where the fused attention + O-projection launch (fused_kernels.h) spends its time: wall_clock64 stamps of attention
workgroup 0 and of two O workgroups, 13b layer shape, 5 rows after a 190-token prefix.  SD_AO_STAMPS=1 is set here."""
import ctypes as C, os, sys
os.environ["SD_AO_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from llmspeculativesampling_amd import _lib, engine
from llmspeculativesampling_amd.config import ModelConfig
cfg = ModelConfig(arch="llama", vocab_size=32000, hidden_size=5120, intermediate_size=13824, num_hidden_layers=1,
                  num_attention_heads=40, num_key_value_heads=40, max_position_embeddings=512, rms_norm_eps=1e-5)
m = engine.SpecDecModel.synthetic(cfg, seed=9, dtype=torch.bfloat16, max_pos=400)
ids = torch.from_numpy(np.random.default_rng(4).integers(3, cfg.vocab_size, size=(400,))).to(torch.int32).cuda()
ses = m.new_session(400)
ses.forward(ids[:190], 0)
# evict caches between runs: stream a big buffer
junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
for rep in range(4):
    junk.add_(1.0)
    ses.rollback(190)
    ses.forward(ids[190:195], 5)
    torch.cuda.synchronize()
    st = (C.c_longlong * 144)()
    _lib.check(_lib.lib.sd_session_ao_stamps(ses.handle, st), "stamps")
    t = [x / 100.0 for x in st]                       # us
    t0 = t[0]
    print(f"rep {rep}: attention wg0: start 0.0, arrived {t[1]-t0:.2f} us")
    heads = [((st[16 + 2 * h] / 100.0) - t0, (st[16 + 2 * h + 1] / 100.0) - t0) for h in range(40)]
    print("   attention heads (stores issued -> counted), by head:", " ".join("%.1f->%.1f" % (a, b) for a, b in heads))
    for name, b in (("O tile 0", 2), ("O tile N/32", 8)):
        print(f"   {name}: start {t[b]-t0:.2f}, weights landed {t[b+1]-t0:.2f}, counter seen {t[b+2]-t0:.2f}, "
              f"MFMAs done {t[b+3]-t0:.2f}, slab stored {t[b+4]-t0:.2f}")
