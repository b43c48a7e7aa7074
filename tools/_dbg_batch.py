import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
from llmspeculativesampling_amd.config import ModelConfig
from llmspeculativesampling_amd.synth import make_state_dict
from llmspeculativesampling_amd import engine
cfg = ModelConfig(arch="llama", vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=4, max_position_embeddings=256, rms_norm_eps=1e-5)
sd = make_state_dict(cfg, 31)
m = engine.SpecDecModel.from_state_dict(cfg, sd, dtype=torch.float32)
rng = np.random.default_rng(8)
def bufs(ses, rows):
    sc = ses.scratch
    H = 64; sz = 64*H*4
    f = lambda off: sc[off:off+rows*H*4].view(torch.float32).view(rows, H).clone()
    return dict(x=f(0), h=f(sz), q=f(2*sz), attn=f(3*sz))
lens, new, nlog = [17,18],[1,1],[1,1]
seqs = [torch.from_numpy(rng.integers(3, cfg.vocab_size, size=(L + n,)).astype(np.int32)).cuda() for L, n in zip(lens, new)]
solo = [m.new_session(96) for _ in lens]; both = [m.new_session(96) for _ in lens]
sb = []
for ses, ses2, sq, L, n, nl in zip(solo, both, seqs, lens, new, nlog):
    ses.forward(sq[:L], 0); ses2.forward(sq[:L], 0)
    ses.forward(sq[L:L + n], nl)
    sb.append(bufs(ses, 1))
got = engine.batch_forward(both, seqs, new, nlog)
bb = bufs(both[0], 2)
for k in ("q","attn","x","h"):
    for i in range(2):
        print(k, "stream", i, "diff", float((bb[k][i] - sb[i][k][0]).abs().max()))
