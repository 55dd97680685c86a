#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE implementation.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
It imports the reference's own modules (`ladiff.models.architectures.*`, CPU fp32), loads the
deterministic synthetic weights of `ladiff_amd.synthetic` (strict load = schema check), runs
seeded inputs and stores INPUTS + EXPECTED OUTPUTS only (weights are regenerated from the seed;
their SHA-256 is stored so generator drift is detected).

    python tests/golden/make_golden.py

The sampling-loop fixtures drive the reference denoiser / VAE modules with the restated loop
and schedulers of oracle/ladiff_oracle.py (`LADIFF` itself needs pytorch_lightning /
torchmetrics / omegaconf / diffusers, none of which exist offline).
"""
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
sys.modules["clip"] = types.ModuleType("clip")   # mdiff_transformer.py:10 imports it, never uses it here

from ladiff.models.architectures.ladiff_denoiser import LADiffDenoiser  # noqa: E402
from ladiff.models.architectures.ladiff_vae import LADiffVae            # noqa: E402
from ladiff.models.architectures.tools.embeddings import get_timestep_embedding  # noqa: E402

from ladiff_amd import synthetic as syn                                  # noqa: E402
from oracle import ladiff_oracle as orc                                  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

ABL = SimpleNamespace(SKIP_CONNECT=True, VAE_TYPE="actor", DIFF_PE_TYPE="mld", PE_TYPE="mld", IDEA="ard",
                      MD_TRANS=True, TEST_EFFICIENCY=False, MLP_DIST=False, DVAE=False, PERCENTAGE_NOISED=0.0,
                      MAX_IT=5, FRAME_PER_LATENT=48, JOINT_DISTRO_FIX=False, LAD=True)


def build_denoiser():
    den = LADiffDenoiser(ABL, nfeats=263, condition="text", latent_dim=[7, 256], ff_size=1024, num_layers=9,
                         num_heads=4, dropout=0.1, normalize_before=False, activation="gelu",
                         flip_sin_to_cos=True, return_intermediate_dec=False, position_embedding="learned",
                         arch="trans_enc", freq_shift=0, guidance_scale=7.5, guidance_uncondp=0.1,
                         text_encoded_dim=768, nclasses=10).eval()
    den.load_state_dict(syn.denoiser_weights(), strict=True)
    return den


def build_vae(nfeats):
    vae = LADiffVae(ABL, nfeats=nfeats, latent_dim=[7, 256], ff_size=1024, num_layers=9, num_heads=4,
                    dropout=0.1, arch="encoder_decoder", normalize_before=False, activation="gelu",
                    position_embedding="learned").eval()
    vae.load_state_dict(syn.vae_weights(nfeats=nfeats), strict=True)
    return vae


def save(name, **arrays):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB  " +
          " ".join(f"{k}{list(np.shape(v))}" for k, v in out.items()))


@torch.no_grad()
def main():
    den = build_denoiser()
    rs = np.random.RandomState(101)

    # (1a) timestep sinusoid, tools/embeddings.py:245-285
    ts = torch.tensor([1, 481, 981])
    save("timestep_embedding", t=ts,
         out=get_timestep_embedding(ts, 768, flip_sin_to_cos=True, downscale_freq_shift=0))

    # (1b) linear cross-attention with N=4 text tokens (general-N path), layer input_blocks.0
    ca = den.encoder.input_blocks[0].ca_block
    x = torch.from_numpy(rs.standard_normal((3, 5, 256)).astype(np.float32))
    xf = torch.from_numpy(rs.standard_normal((3, 4, 256)).astype(np.float32))
    emb = torch.from_numpy(rs.standard_normal((3, 256)).astype(np.float32))
    pad = torch.tensor([[False] * 5, [False, False, True, True, True], [False, False, False, True, True]])
    save("cross_attention_n4", x=x, xf=xf, emb=emb, pad=pad, out=ca(x, xf, emb, src_key_padding_mask=pad))

    # (2) full denoiser forward, B2=8, mixed max_iter_elements, t in {981, 1}; per-block outputs too
    B2 = 8
    sample = torch.from_numpy((3.0 * rs.standard_normal((B2, 5, 256))).astype(np.float32))
    text = torch.from_numpy(rs.standard_normal((B2, 1, 768)).astype(np.float32))
    counts = torch.tensor([5, 2, 3, 1, 5, 2, 3, 1])
    blocks = list(den.encoder.input_blocks) + [den.encoder.middle_block] + list(den.encoder.output_blocks)
    for t in (981, 1):
        taps = []
        hooks = [b.register_forward_hook(lambda m, i, o: taps.append(o.permute(1, 0, 2).clone())) for b in blocks]
        eps = den(sample=sample, timestep=torch.tensor(t), encoder_hidden_states=text,
                  lengths=[196] * B2, max_iter_elements=counts)[0]
        for h in hooks:
            h.remove()
        save(f"denoiser_forward_t{t}", sample=sample, text=text, counts=counts, t=np.int64(t), eps=eps,
             blocks=torch.stack(taps))

    # (3) vae.decode: c1 (B=8, F=60, C=263) and mixed KIT (B=3, {60,120,196}, C=251)
    vae = build_vae(263)
    z = torch.from_numpy(rs.standard_normal((5, 8, 256)).astype(np.float32))
    z[2:] = 0
    save("vae_decode_c1", z=z, lengths=np.array([60] * 8), feats=vae.decode(z, [60] * 8))
    lens = [60, 120, 196]
    z = torch.from_numpy(rs.standard_normal((5, 3, 256)).astype(np.float32))
    for i, m in enumerate(syn.max_iter_elements(lens)):
        z[m:, i] = 0
    kit = build_vae(251)
    dblocks = list(kit.decoder.input_blocks) + [kit.decoder.middle_block] + list(kit.decoder.output_blocks)
    taps = []
    hk = dblocks[0].register_forward_hook(lambda m, i, o: taps.append(o.permute(1, 0, 2).clone()))
    feats = kit.decode(z, lens)
    hk.remove()
    save("vae_decode_mixed_kit", z=z, lengths=np.array(lens), feats=feats, block0=taps[0])
    # ragged short batch on HumanML3D shapes (F not a multiple of anything, 1-latent sample)
    lens = [37, 49, 5, 101]
    z = torch.from_numpy(rs.standard_normal((5, 4, 256)).astype(np.float32))
    for i, m in enumerate(syn.max_iter_elements(lens)):
        z[m:, i] = 0
    save("vae_decode_ragged", z=z, lengths=np.array(lens), feats=vae.decode(z, lens))

    # (4) sampling loop with the reference modules as callables, B=4, lengths {60,120,196,196}
    lens = [60, 120, 196, 196]
    text = syn.text_embeddings(4, seed=21)
    noise = syn.init_noise(lens, seed=22)

    def ref_denoise(xin, t, txt, counts2):
        return den(sample=xin, timestep=torch.as_tensor(t), encoder_hidden_states=txt, lengths=lens * 2,
                   max_iter_elements=torch.as_tensor(counts2))[0]

    for tag, sch, n, step_noise in (("ddim5", orc.DDIM(), 5, None), ("ddim50", orc.DDIM(), 50, None),
                                    ("ddpm10", orc.DDPM(), 10, syn.ddpm_noise(10, 4, seed=23))):
        zf = orc.diffusion_reverse(ref_denoise, sch, text, lens, noise, n, 7.5, 0.0, step_noise)
        save(f"loop_{tag}", text=text, lengths=np.array(lens), init_noise=noise, n_steps=np.int64(n),
             timesteps=sch.timesteps, latents=zf, feats=vae.decode(zf, lens),
             **({} if step_noise is None else {"step_noise": step_noise}))

    with open(os.path.join(HERE, "weights.sha256.json"), "w") as f:
        json.dump({"denoiser_seed1234": syn.state_dict_sha256(syn.denoiser_weights()),
                   "vae263_seed1235": syn.state_dict_sha256(syn.vae_weights(263)),
                   "vae251_seed1235": syn.state_dict_sha256(syn.vae_weights(251))}, f, indent=1)


if __name__ == "__main__":
    main()
