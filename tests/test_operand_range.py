"""How far the matrix-product operands of the path are from fp16's range (CPU; the oracle is the instrument).

The split arithmetic mode of the library ("f16x3", csrc/common.h) carries every GEMM / attention operand as a pair of fp16 halves: exact to
22 bits for |x| <= 65504, saturating beyond 131008.  This test runs the ORACLE (not the product) with its `linear` and attention products
wrapped to record the largest operand magnitude: the 50-step guided loop + decode on the synthetic random-init weights - the hard case,
latents reach |x| ~ 280 there - and the CLIP tower.  It documents the margin DESIGN.md 1 states and fails if a change of the synthetic
weights ever moved an operand within 8x of the fp16 limit."""
import torch

from ladiff_amd import synthetic as syn
from oracle import ladiff_oracle as orc

F16_MAX = 65504.0


def _record(fn):
    seen = {"x": 0.0, "w": 0.0, "attn": 0.0}
    real_linear, real_softmax = orc.linear, torch.softmax

    def linear(x, w, b=None):
        seen["x"] = max(seen["x"], x.abs().max().item())
        seen["w"] = max(seen["w"], w.abs().max().item())
        y = real_linear(x, w, b)
        seen["attn"] = max(seen["attn"], y.abs().max().item())      # q | k | v (attention operands) are outputs of `linear`
        return y

    orc.linear = linear
    try:
        with torch.no_grad():
            fn()
    finally:
        orc.linear = real_linear
    return seen


def test_loop_and_decode_operands_are_far_inside_fp16_range():
    lens = [196, 60, 120, 150]
    text, noise = syn.text_embeddings(len(lens)), syn.init_noise(lens)
    den, vae = syn.denoiser_weights(), syn.vae_weights(263)
    seen = _record(lambda: orc.sample_motions(den, vae, text, lens, noise, 50, "ddim"))
    print(f"loop + decode, 50 steps: largest GEMM input {seen['x']:.1f}, largest weight {seen['w']:.3f}, largest GEMM output (q | k | v, hidden) {seen['attn']:.1f}")
    assert max(seen.values()) < F16_MAX / 8, seen


def test_clip_tower_operands_are_far_inside_fp16_range():
    ids = syn.clip_token_ids(8, empty_first=4)
    sd = syn.clip_weights()
    seen = _record(lambda: orc.clip_text_features(sd, ids, 12))
    print(f"CLIP tower: largest GEMM input {seen['x']:.1f}, largest weight {seen['w']:.3f}, largest GEMM output {seen['attn']:.1f}")
    assert max(seen.values()) < F16_MAX / 8, seen
