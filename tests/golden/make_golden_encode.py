#!/usr/bin/env python3
"""Golden vectors for LADiffVae.encode from the REFERENCE module (build container only; see make_golden.py)."""
import os, sys, types
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/src"); sys.modules["clip"] = types.ModuleType("clip")
import make_golden as mg                     # noqa: E402  (reference module builders + ABL)
from ladiff_amd import synthetic as syn      # noqa: E402

torch.set_num_threads(8)
with torch.no_grad():
    for name, C, lens in (("vae_encode_humanml", 263, [60, 120, 196]), ("vae_encode_kit", 251, [33, 100])):
        vae = mg.build_vae(C)
        rs = np.random.RandomState(77 + C)
        feats = torch.from_numpy(rs.standard_normal((len(lens), max(lens), C)).astype(np.float32))
        torch.manual_seed(5)
        latent, dist, counts = vae.encode(feats, lens)
        mu, std = dist.loc, dist.scale
        eps = (latent - mu) / std                               # the draw rsample() made (valid rows)
        for i, c in enumerate(counts.tolist()):
            eps[c:, i] = 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), features=feats.numpy(), lengths=np.array(lens), mu=mu.numpy(),
                            std=std.numpy(), latent=latent.numpy(), eps=eps.numpy(), counts=counts.numpy())
        print(name, tuple(latent.shape), counts.tolist())
