"""Pins oracle/ladiff_oracle.py against vectors captured from the reference (tests/golden/make_golden.py)."""
import json
import os

import pytest
import torch

from ladiff_amd import synthetic as syn
from oracle import ladiff_oracle as orc

from conftest import GOLDEN, load_golden

TOL = 2e-5   # fp32 CPU restatement vs fp32 CPU reference: same math, different op order


def maxdiff(a, b):
    return (a.double() - b.double()).abs().max().item()


@pytest.fixture(scope="module")
def den_sd():
    return syn.denoiser_weights()


def test_weight_generator_is_stable():
    want = json.load(open(os.path.join(GOLDEN, "weights.sha256.json")))
    assert syn.state_dict_sha256(syn.denoiser_weights()) == want["denoiser_seed1234"]
    assert syn.state_dict_sha256(syn.vae_weights(263)) == want["vae263_seed1235"]
    assert syn.state_dict_sha256(syn.vae_weights(251)) == want["vae251_seed1235"]


def test_timestep_sinusoid():
    g = load_golden("timestep_embedding")
    assert maxdiff(orc.timestep_sinusoid(g["t"]), g["out"]) < 1e-6


def test_cross_attention_general_n(den_sd):
    g = load_golden("cross_attention_n4")
    p = orc.sub(den_sd, "encoder.input_blocks.0.ca_block")
    out = orc.linear_cross_attention(g["x"], g["xf"], g["emb"], p, g["pad"])
    assert maxdiff(out, g["out"]) < TOL


@pytest.mark.parametrize("t", [981, 1])
def test_denoiser_forward(den_sd, t):
    g = load_golden(f"denoiser_forward_t{t}")
    eps = orc.denoiser_forward(den_sd, g["sample"], g["t"], g["text"], g["counts"])
    assert maxdiff(eps, g["eps"]) < TOL


@pytest.mark.parametrize("name,nfeats", [("vae_decode_c1", 263), ("vae_decode_mixed_kit", 251),
                                         ("vae_decode_ragged", 263)])
def test_vae_decode(name, nfeats):
    g = load_golden(name)
    feats = orc.vae_decode(syn.vae_weights(nfeats), g["z"], g["lengths"].tolist())
    assert feats.shape == g["feats"].shape
    assert maxdiff(feats, g["feats"]) < TOL


@pytest.mark.parametrize("tag,sched", [("ddim5", "ddim"), ("ddim50", "ddim"), ("ddpm10", "ddpm")])
def test_sampling_loop(den_sd, tag, sched):
    g = load_golden(f"loop_{tag}")
    z, feats = orc.sample_motions(den_sd, syn.vae_weights(263), g["text"], g["lengths"].tolist(),
                                  g["init_noise"], int(g["n_steps"]), sched,
                                  step_noise=g.get("step_noise"))
    # 50 guided steps on random-init weights amplify rounding (latents reach |x|~300): scale the bound
    scale = max(1.0, g["latents"].abs().max().item())
    assert maxdiff(z, g["latents"]) < 5e-6 * scale
    assert maxdiff(feats, g["feats"]) < 1e-4


def test_scheduler_known_answers():
    """A3 is unpinned (no diffusers here): check the published closed-form properties only."""
    d = orc.DDIM()
    d.set_timesteps(50)
    assert d.timesteps[0].item() == 981 and d.timesteps[-1].item() == 1 and len(d.timesteps) == 50
    d.set_timesteps(20)
    assert d.timesteps[0].item() == 951 and d.timesteps[-1].item() == 1
    assert abs(d.alphas_cumprod[0].item() - (1 - 0.00085)) < 1e-7
    assert abs(d.betas[-1].item() - 0.012) < 1e-7
    p = orc.DDPM()
    p.set_timesteps(1000)
    assert p.timesteps[0].item() == 999 and p.timesteps[-1].item() == 0
    # DDIM with eta=0 and eps=0 only rescales x by sqrt(a_prev/a_t)
    x = torch.ones(1, 1, 4)
    d.set_timesteps(50)
    y = d.step(torch.zeros_like(x), 981, x)
    assert torch.allclose(y, x * (d.alphas_cumprod[961] / d.alphas_cumprod[981]) ** 0.5, atol=1e-6)


@pytest.mark.parametrize("name", ["feats2joints_humanml", "feats2joints_kit"])
def test_feats2joints(name):
    """Next row of the scope table, pinned to the reference's recover_from_ric."""
    g = load_golden(name)
    j = orc.feats2joints(g["feats"], g["mean"], g["std"], int(g["njoints"]))
    assert j.shape == g["joints"].shape and maxdiff(j, g["joints"]) < 1e-6


@pytest.mark.parametrize("name,nfeats", [("vae_encode_humanml", 263), ("vae_encode_kit", 251)])
def test_vae_encode(name, nfeats):
    """Next row of the scope table, pinned to the reference's LADiffVae.encode."""
    g = load_golden(name)
    mu, std, latent = orc.vae_encode(syn.vae_weights(nfeats), g["features"], g["lengths"].tolist(), g["eps"])
    assert maxdiff(mu, g["mu"]) < TOL and maxdiff(std, g["std"]) < TOL and maxdiff(latent, g["latent"]) < TOL


@pytest.mark.parametrize("name", ["clip_small", "clip_small_eos", "clip_full"])
def test_clip_text_features_against_transformers(name):
    """the restated CLIP text tower vs transformers' CLIPModel.get_text_features (make_golden_clip.py)."""
    g = load_golden(name)
    vocab, layers = int(g["vocab"]), int(g["layers"])
    sd = syn.clip_weights(vocab, layers)
    ids = g["ids"]
    out = orc.clip_text_features(sd, ids, layers)
    assert (out - g["text_features"]).abs().max().item() < 2e-5
    L = int(ids.argmax(-1).max()) + 1                      # the causal mask makes positions behind the EOS irrelevant
    assert torch.equal(orc.clip_text_features(sd, ids[:, :L], layers), out)


# ---------------------------------------------------------------- T2M evaluators + TM2T metrics
@pytest.mark.parametrize("name,nfeats", [("t2m_humanml", 263), ("t2m_kit", 251)])
def test_t2m_evaluator_encoders_against_reference(name, nfeats):
    g = load_golden(name)
    mv, mo, tx = syn.t2m_weights(nfeats)
    assert maxdiff(orc.t2m_movement_encoder(mv, g["feats"]), g["movements"]) < 1e-5
    assert maxdiff(orc.t2m_motion_encoder(mo, g["movements"], g["lengths"] // 4), g["motion_emb"]) < 2e-5
    assert maxdiff(orc.t2m_text_encoder(tx, g["word_embs"], g["pos_onehot"], g["cap_lens"]), g["text_emb"]) < 2e-5


def test_tm2t_metrics_against_reference_helpers():
    """Oracle restatement and the shipped host-side `TM2TMetrics` vs numbers computed with the reference's metrics/utils.py."""
    from ladiff_amd.evaluators import TM2TMetrics
    g = load_golden("tm2t_metrics")
    want = {k: float(g[k]) for k in TM2TMetrics().metrics}
    got = orc.tm2t_metrics(g["text"], g["gen"], g["gt"], g["order"], g["div_first"], g["div_second"])
    m = TM2TMetrics()
    for lo in range(0, 352, 88):                                   # four update() calls, as test batches arrive
        m.update(g["text"][lo:lo + 88], g["gen"][lo:lo + 88], g["gt"][lo:lo + 88], [196] * 88)
    shipped = m.compute(order=g["order"].numpy(), div_first=g["div_first"].numpy(), div_second=g["div_second"].numpy())
    assert m.count_seq == 352 and m.count == 352 * 196
    for k, v in want.items():
        tol = 1e-6 * max(1.0, abs(v)) if "R_precision" not in k else 1e-7      # hit counts are exact; the reference divides in fp32
        assert abs(got[k] - v) <= tol, k
        assert abs(shipped[k] - v) <= tol, k
    assert 0 < want["R_precision_top_1"] < want["R_precision_top_3"] < 1       # a case where ranking actually matters
    assert TM2TMetrics().compute(sanity_flag=True)["FID"] == 0.0
