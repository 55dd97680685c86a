"""GPU parity of the T2M evaluator encoders against goldens captured from the reference's modules
(tests/golden/make_golden_t2m.py) and of `LADIFF.t2m_eval` against the oracle chain."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from ladiff_amd import (LADIFF, DDIMScheduler, LADiffDenoiser, LADiffVae, MotionEncoderBiGRUCo, MovementConvEncoder,
                        TextEncoderBiGRUCo, TM2TMetrics, _lib, synthetic as syn)
from oracle import ladiff_oracle as orc
from conftest import load_golden
from test_abi import ABL, DEN_KW, VAE_KW

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def maxdiff(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


def make_evaluators(nfeats):
    mv, mo, tx = syn.t2m_weights(nfeats)
    move = MovementConvEncoder(nfeats - 4, 512, 512); move.load_state_dict(mv, strict=True)
    motion = MotionEncoderBiGRUCo(512, 1024, 512); motion.load_state_dict(mo, strict=True)
    text = TextEncoderBiGRUCo(300, 15, 512, 512); text.load_state_dict(tx, strict=True)
    return move.to(DEV), motion.to(DEV), text.to(DEV)


@pytest.mark.parametrize("name,nfeats", [("t2m_humanml", 263), ("t2m_kit", 251)])
def test_encoders_match_reference_golden(name, nfeats):
    g = load_golden(name)
    move, motion, text = make_evaluators(nfeats)
    feats = g["feats"].to(DEV)
    mov = move(feats[..., :-4])                                  # the strided view the reference passes (ladiff.py:1264)
    assert mov.shape == g["movements"].shape and maxdiff(mov, g["movements"]) < 2e-5
    assert maxdiff(move(feats[..., :-4].contiguous()), g["movements"]) < 2e-5
    emb = motion(mov, g["lengths"] // 4)
    assert maxdiff(emb, g["motion_emb"]) < 5e-5
    temb = text(g["word_embs"].to(DEV), g["pos_onehot"].to(DEV), g["cap_lens"])
    assert maxdiff(temb, g["text_emb"]) < 5e-5


def test_gru_lengths_in_any_order_and_odd_frame_counts():
    """pack_padded_sequence wants lengths sorted; the kernel does not care.  Frame counts that are not multiples of 4."""
    mv, mo, tx = syn.t2m_weights(263)
    move, motion, text = make_evaluators(263)
    gen = torch.Generator().manual_seed(3)
    feats = torch.randn(5, 75, 263, generator=gen)
    lens = torch.tensor([9, 18, 4, 18, 1])
    mov_o = orc.t2m_movement_encoder(mv, feats)
    assert mov_o.shape == (5, 18, 512)
    mov = move(feats.to(DEV)[..., :-4])
    assert maxdiff(mov, mov_o) < 2e-5
    assert maxdiff(motion(mov, lens), orc.t2m_motion_encoder(mo, mov_o, lens)) < 5e-5
    with pytest.raises(RuntimeError, match="greater than 0"):
        motion(mov, torch.tensor([9, 18, 0, 18, 1]))
    with pytest.raises(_lib.LadiffHipError):
        move(feats[..., :-4])                                     # CPU tensor: no fallback


def test_t2m_eval_end_to_end_against_oracle():
    """prompts -> 5-step guided DDIM -> decode -> renorm -> evaluator embeddings -> TM2T metrics, HIP path vs oracle chain."""
    lens = [60, 196, 120, 24]
    B = len(lens)
    den = LADiffDenoiser(ABL, **DEN_KW); den.load_state_dict(syn.denoiser_weights())
    vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263))
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                        clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    rs = np.random.RandomState(2)
    mean = torch.from_numpy(rs.standard_normal(263).astype(np.float32)) * 0.1
    std = torch.from_numpy(rs.uniform(0.5, 1.5, 263).astype(np.float32))
    mean_e = torch.from_numpy(rs.standard_normal(263).astype(np.float32)) * 0.1
    std_e = torch.from_numpy(rs.uniform(0.5, 1.5, 263).astype(np.float32))

    def renorm(f, mean=mean, std=std, mean_e=mean_e, std_e=std_e):              # HumanML3D.py renorm4t2m: de-normalise, re-normalise
        d = f.device
        return (f * std.to(d) + mean.to(d) - mean_e.to(d)) / std_e.to(d)

    dm = SimpleNamespace(renorm4t2m=renorm, mean=mean, std=std, njoints=22,
                         feats2joints=lambda f: orc.feats2joints(f, mean, std, 22))
    text_all = syn.text_embeddings(B, seed=61)                                    # [2B, 1, 768], unconditional half first
    model = LADIFF(None, dm, denoiser=den.to(DEV).eval(), vae=vae.to(DEV).eval(), scheduler=sch, guidance_scale=7.5,
                   num_inference_timesteps=5, eta=0.0, text_encoder=lambda texts: text_all.to(DEV))
    move, motion, text = make_evaluators(263)
    model.set_t2m_evaluators(text, move, motion, unit_len=4)
    noise = syn.init_noise(lens, seed=62)
    gen = torch.Generator().manual_seed(63)
    motions = torch.randn(B, max(lens), 263, generator=gen)
    for i, l in enumerate(lens):
        motions[i, l:] = 0
    cap = torch.tensor([12, 9, 7, 3])
    word = torch.randn(B, 12, 300, generator=gen)
    pos = torch.nn.functional.one_hot(torch.randint(0, 15, (B, 12), generator=gen), 15).float()
    batch = {"text": ["a"] * B, "length": lens, "motion": motions, "word_embs": word, "pos_ohot": pos, "text_len": cap}
    orig = model._diffusion_reverse
    model._diffusion_reverse = lambda emb, lengths: orig(emb, lengths, init_noise=noise.to(DEV))
    rs_set = model.t2m_eval(batch)

    mv, mo, tx = syn.t2m_weights(263)
    z_o, feats_o = orc.sample_motions(syn.denoiser_weights(), syn.vae_weights(263), text_all, lens, noise, 5, "ddim")
    order = np.argsort(lens)[::-1].copy()
    f_o, m_o = renorm(feats_o)[order], renorm(motions)[order]
    ml = torch.tensor(lens)[order] // 4
    lat_rm = orc.t2m_motion_encoder(mo, orc.t2m_movement_encoder(mv, f_o), ml)
    lat_m = orc.t2m_motion_encoder(mo, orc.t2m_movement_encoder(mv, m_o), ml)
    lat_t = orc.t2m_text_encoder(tx, word, pos, cap)[order]
    assert maxdiff(rs_set["m_rst"], f_o) < 2e-3
    assert maxdiff(rs_set["lat_m"], lat_m) < 1e-4 and maxdiff(rs_set["lat_t"], lat_t) < 1e-4
    assert maxdiff(rs_set["lat_rm"], lat_rm) < 2e-3               # carries the sampling path's own tolerance
    assert rs_set["joints_rst"].shape == (B, 196, 22, 3)
    m = TM2TMetrics(top_k=1, R_size=2, diversity_times=2)
    m.update(rs_set["lat_t"], rs_set["lat_rm"], rs_set["lat_m"], lens)
    out = m.compute()
    assert np.isfinite(list(out.values())).all() and out["gt_Matching_score"] > 0
