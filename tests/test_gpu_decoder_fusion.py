"""The decoder layer's self-attention out_proj + norm1 + cross-attention + norm2 as ONE kernel (csrc/dec_cross.hip,
dec_out_cross_kernel: Wo and the sample's folded keys / values as MFMA fragments in registers, f16x3 mode, from 4,096 frame rows up)
against the two launches it replaces and against the CPU oracle - for every latent-token count 1 .. 8, padded and ragged rows.
Reference: TransformerDecoderLayer.forward_post, operator/cross_attention.py:367-376, :407-409."""
import pytest
import torch

from ladiff_amd import LADiffVae, _lib, synthetic as syn
from oracle import ladiff_oracle as orc
from test_abi import ABL, VAE_KW

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vae():
    m = LADiffVae(ABL, **VAE_KW)
    m.load_state_dict(syn.vae_weights(263), strict=True)
    return m.to(DEV).eval()


@pytest.mark.parametrize("T,fpl", [(1, 196), (2, 98), (3, 66), (4, 49), (5, 48), (6, 33), (7, 28), (8, 25)])
@pytest.mark.parametrize("ragged", [False, True])
def test_fused_out_proj_cross_attention_kernel(vae, T, fpl, ragged):
    """T latent tokens per sample (frame_per_latent chosen so that a 196-frame motion uses all T), 48 samples of mixed lengths
    (> 4,096 rows: the fused kernel runs), rows of a pass that straddle the end of a sample, a sample shorter than one 32-row pass."""
    L = _lib.lib()
    B = 48
    lens = [196, 1, 33, 196, 64, 150, 97, 196, 31, 32, 128, 196] * 4            # 5,280 frames: past the 4,096-row threshold also as ragged rows
    z = torch.randn(T, B, 256, generator=torch.Generator().manual_seed(10 + T))
    counts = [-(-l // fpl) for l in lens]
    for i, c in enumerate(counts):
        z[c:, i] = 0
    old = (vae.frame_per_latent, vae.length_aware, vae.precision, vae.max_it)
    try:
        vae.frame_per_latent, vae.length_aware, vae.precision = fpl, ragged, "f16x3"
        with torch.no_grad():
            assert L.ladiff_debug_set_decoder_fusion(1) == 0
            fused = vae.decode(z.to(DEV), lens)
            again = vae.decode(z.to(DEV), lens)
            assert L.ladiff_debug_set_decoder_fusion(1 + 64) == 0
            two = vae.decode(z.to(DEV), lens)
    finally:
        L.ladiff_debug_set_decoder_fusion(1)
        vae.frame_per_latent, vae.length_aware, vae.precision, vae.max_it = old
    ref = orc.vae_decode(syn.vae_weights(263), z, lens, frame_per_latent=fpl)
    scale = max(1.0, ref.abs().max().item())
    assert torch.isfinite(fused).all() and torch.equal(fused, again)
    d2 = (fused - two).abs().max().item()
    dr = (fused.cpu() - ref).abs().max().item()
    print(f"T={T} ragged={ragged}: fused vs two launches {d2:.2e}, fused vs oracle {dr:.2e}")
    assert 0.0 < d2 < 2e-4 * scale           # the cross-attention products on the f16x3 MFMA instead of fp32 FMAs: 2^-16 per product
    assert dr < 5e-4 * scale
    for i, l in enumerate(lens):
        if l < fused.shape[1]:
            assert fused[i, l:].abs().max().item() == 0.0
