"""Readers for the fixtures under tests/golden/ (written by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    blobs = np.load(os.path.join(GOLD, name + ".npz"))
    return meta, blobs


def logits_row(seed, V, scale=4.0, dtype=torch.float32):
    rng = np.random.default_rng([seed, V])
    return torch.from_numpy((rng.standard_normal(V, dtype=np.float32) * np.float32(scale))[None]).to(dtype)


def events(blobs, prefix):
    """Rebuild the ordered noise stream [("exp", tensor(1,V)) | ("uni", tensor(1)) | ("seed", int)]."""
    kinds = blobs[prefix + "_kinds"]
    exp, uni, seed = blobs[prefix + "_exp"], blobs[prefix + "_uni"], blobs[prefix + "_seed"]
    ie = iu = isd = 0
    out = []
    for k in kinds:
        if k == 0:
            out.append(("exp", torch.from_numpy(exp[ie][None].copy())))
            ie += 1
        elif k == 1:
            out.append(("uni", torch.tensor([uni[iu]], dtype=torch.float32)))
            iu += 1
        else:
            out.append(("seed", int(seed[isd])))
            isd += 1
    return out


def events_ragged(blobs, prefix):
    """G7 streams: Exp(1) draws of shape (rows, V) - (width, V) for the batched draft / target samples, (1, V) for
    the residual-or-bonus sample - stored flat with their row counts."""
    kinds = blobs[prefix + "_kinds"]
    flat, rows, size = blobs[prefix + "_expflat"], blobs[prefix + "_exprows"], blobs[prefix + "_expsize"]
    uni, seed = blobs[prefix + "_uni"], blobs[prefix + "_seed"]
    ie = iu = isd = off = 0
    out = []
    for k in kinds:
        if k == 0:
            n = int(size[ie])
            out.append(("exp", torch.from_numpy(flat[off:off + n].reshape(int(rows[ie]), -1).copy())))
            off += n
            ie += 1
        elif k == 1:
            out.append(("uni", torch.tensor([uni[iu]], dtype=torch.float32)))
            iu += 1
        else:
            out.append(("seed", int(seed[isd])))
            isd += 1
    return out


def dense_from_sparse(V, idx, val):
    p = np.zeros(V, dtype=np.float32)
    p[idx] = val
    return p


DT = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}


def model_pair(case):
    """(draft_cfg, draft_sd, target_cfg, target_sd) of a G5 trace, rebuilt from seeds."""
    from llmspeculativesampling_amd.config import load_config
    from llmspeculativesampling_amd.synth import make_state_dict, perturb_state_dict
    dcfg, tcfg = load_config(case["draft_cfg"]), load_config(case["target_cfg"])
    dsd = make_state_dict(dcfg, case["draft_seed"])
    spec = case["target_spec"]
    if spec[0] == "same":
        tsd = dsd
    elif spec[0] == "perturb":
        tsd = perturb_state_dict(dsd, spec[1], spec[2])
    else:
        tsd = make_state_dict(tcfg, spec[1])
    return dcfg, dsd, tcfg, tsd
