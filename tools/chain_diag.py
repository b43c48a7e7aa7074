"""Diagnostic: chained layer launch (SD_CHAIN=1) against the launch-per-op path, per layer and per position."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llmspeculativesampling_amd import engine
from llmspeculativesampling_amd.config import ModelConfig
from llmspeculativesampling_amd.synth import make_state_dict

CFG = dict(arch="opt", vocab_size=16384, hidden_size=1024, ffn_dim=4096, num_hidden_layers=3, num_attention_heads=16,
           max_position_embeddings=256, do_layer_norm_before=True, word_embed_proj_dim=1024)


def run(m, ids, flag, steps):
    os.environ["SD_CHAIN"] = flag
    ses = m.new_session(160)
    ses.forward(ids[:40], 0)
    got, pos = [], 40
    for q in steps:
        got.append(ses.forward(ids[pos:pos + q], q).clone())
        pos += q
    return torch.cat(got), ses.kv[:, :, :, :pos].clone()


def main():
    dtype = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else torch.bfloat16
    cfg = ModelConfig(**CFG)
    sd = make_state_dict(cfg, 91, dtype=dtype)
    m = engine.SpecDecModel.from_state_dict(cfg, sd, dtype=dtype)
    ids = torch.from_numpy(np.random.default_rng(23).integers(3, cfg.vocab_size, size=(1, 120))).to(torch.int32).cuda()[0]
    steps = (5, 1, 5, 16, 3, 5, 5, 2, 5, 9)
    ref = run(m, ids, "0", steps)
    for rep in range(3):
        got = run(m, ids, "1", steps)
        lg = (got[0].float() - ref[0].float()).abs()
        print(f"rep {rep}: logits max diff {float(lg.max()):.3e}, rows differing {int((lg.amax(1) > 0).sum())} of {lg.shape[0]}")
        kv = (got[1].float() - ref[1].float()).abs()            # [L][2][H][pos][D]
        for l in range(kv.shape[0]):
            d = kv[l].amax(dim=(0, 1, 3))                        # per position
            bad = torch.nonzero(d > 0).flatten().tolist()
            print(f"   layer {l}: max {float(kv[l].max()):.3e} positions {bad[:20]}")


if __name__ == "__main__":
    main()
