#!/usr/bin/env python3
"""Golden vectors for the T2M evaluator encoders and the TM2T metric helpers from the REFERENCE modules
(`t2m_motionenc.py`, `t2m_textenc.py`, `metrics/utils.py`: torch / numpy / scipy only, importable in the build container).
Run here only: `python tests/golden/make_golden_t2m.py`; fixtures hold inputs and expected outputs, weights come from the seed."""
import importlib.util, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ladiff_amd import synthetic as syn      # noqa: E402

REF = "/root/reference/src/ladiff/models"


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


menc = load(f"{REF}/architectures/t2m_motionenc.py", "ref_t2m_motionenc")
tenc = load(f"{REF}/architectures/t2m_textenc.py", "ref_t2m_textenc")
mutil = load(f"{REF}/metrics/utils.py", "ref_metric_utils")
torch.set_num_threads(8)

with torch.no_grad():
    for name, C, lens in (("t2m_humanml", 263, [196, 180, 120, 64, 24, 8]), ("t2m_kit", 251, [196, 100, 40])):
        mv_sd, mo_sd, tx_sd = syn.t2m_weights(C)
        move = menc.MovementConvEncoder(C - 4, 512, 512).eval(); move.load_state_dict(mv_sd, strict=True)
        motion = menc.MotionEncoderBiGRUCo(512, 1024, 512).eval(); motion.load_state_dict(mo_sd, strict=True)
        text = tenc.TextEncoderBiGRUCo(300, 15, 512, 512).eval(); text.load_state_dict(tx_sd, strict=True)
        rs = np.random.RandomState(90 + C)
        B, Fm = len(lens), max(lens)
        feats = torch.from_numpy(rs.standard_normal((B, Fm, C)).astype(np.float32))
        for i, l in enumerate(lens):
            feats[i, l:] = 0                                   # the datamodule pads with zeros
        m_lens = torch.tensor(lens) // 4                       # ladiff.py:1255 (sorted descending like the collate does)
        mov = move(feats[..., :-4])
        emb = motion(mov, m_lens)
        cap = torch.tensor(sorted(rs.randint(2, 21, size=B).tolist(), reverse=True))
        L = int(cap.max())
        word = torch.from_numpy(rs.standard_normal((B, L, 300)).astype(np.float32))
        pos = torch.zeros(B, L, 15); pos[torch.arange(B)[:, None], torch.arange(L)[None], torch.from_numpy(rs.randint(0, 15, (B, L)))] = 1
        temb = text(word, pos, cap)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), feats=feats.numpy(), lengths=np.array(lens), movements=mov.numpy(),
                            motion_emb=emb.numpy(), word_embs=word.numpy(), pos_onehot=pos.numpy(), cap_lens=cap.numpy(),
                            text_emb=temb.numpy())
        print(name, tuple(mov.shape), tuple(emb.shape), tuple(temb.shape), float(emb.abs().max()), float(temb.abs().max()))

    # metric helpers on random embeddings (N = 352 sequences of 512-d co-embeddings)
    rs = np.random.RandomState(5)
    N = 352
    t = torch.from_numpy(rs.standard_normal((N, 512)).astype(np.float32))
    gen = t + 9.0 * torch.from_numpy(rs.standard_normal((N, 512)).astype(np.float32))
    gt = t + 6.0 * torch.from_numpy(rs.standard_normal((N, 512)).astype(np.float32))
    order = rs.permutation(N)
    d1, d2 = rs.choice(N, 300, replace=False), rs.choice(N, 300, replace=False)
    out = {}
    for tag, mot in (("", gen), ("gt_", gt)):
        match, topk = 0.0, torch.zeros(3)
        for i in range(N // 32):
            a, b = t[order][i * 32:(i + 1) * 32], mot[order][i * 32:(i + 1) * 32]
            dist = mutil.euclidean_distance_matrix(a, b).nan_to_num()
            match += dist.trace().item()
            topk += mutil.calculate_top_k(torch.argsort(dist, dim=1), top_k=3).sum(axis=0)
        out[tag + "Matching_score"] = match / (N // 32 * 32)
        for k in range(3):
            out[f"{tag}R_precision_top_{k + 1}"] = float(topk[k] / (N // 32 * 32))
    mu, cov = mutil.calculate_activation_statistics_np(gen[order].numpy())
    gmu, gcov = mutil.calculate_activation_statistics_np(gt[order].numpy())
    out["FID"] = float(mutil.calculate_frechet_distance_np(gmu, gcov, mu, cov))
    g_, t_ = gen[order].numpy(), gt[order].numpy()
    out["Diversity"] = float(np.linalg.norm(g_[d1] - g_[d2], axis=1).mean())       # calculate_diversity_np with its draws fixed
    out["gt_Diversity"] = float(np.linalg.norm(t_[d1] - t_[d2], axis=1).mean())
    np.savez_compressed(os.path.join(HERE, "tm2t_metrics.npz"), text=t.numpy(), gen=gen.numpy(), gt=gt.numpy(), order=order,
                        div_first=d1, div_second=d2, **{k: np.array(v) for k, v in out.items()})
    print(out)
