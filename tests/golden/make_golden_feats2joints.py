#!/usr/bin/env python3
"""Golden vectors for feats2joints from the REFERENCE's recover_from_ric (build container only; see make_golden.py)."""
import os, sys, types
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src"); sys.modules["clip"] = types.ModuleType("clip")
from ladiff.data.humanml.scripts.motion_process import recover_from_ric   # noqa: E402

torch.manual_seed(0)
for name, C, J, B, F in (("feats2joints_humanml", 263, 22, 3, 196), ("feats2joints_kit", 251, 21, 2, 60)):
    rs = np.random.RandomState(41 + C)
    feats = torch.from_numpy(rs.standard_normal((B, F, C)).astype(np.float32))
    feats[0, F // 2:] = 0                                    # a padded tail, as vae.decode leaves it
    mean = torch.from_numpy((0.1 * rs.standard_normal(C)).astype(np.float32))
    std = torch.from_numpy((0.05 + 0.2 * rs.random_sample(C)).astype(np.float32))
    joints = recover_from_ric(feats * std + mean, J)         # = HumanML3DDataModule.feats2joints, HumanML3D.py:44-48
    np.savez_compressed(os.path.join(HERE, name + ".npz"), feats=feats.numpy(), mean=mean.numpy(), std=std.numpy(),
                        njoints=np.int64(J), joints=joints.numpy())
    print(name, tuple(joints.shape))
