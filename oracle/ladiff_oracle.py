"""ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by the product path (`ladiff_amd/`).

CPU restatement, in plain PyTorch (fp32 by default, fp64 on request), of the LADiff sampling
hot path: DDIM/DDPM reverse loop with classifier-free guidance over the length-aware
denoiser, then the LA-VAE decoder.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module, and only as the checker / the
timed CPU baseline.

Parity status
-------------
* Denoiser forward (rows A5-A14 of SURVEY.md §8a) and VAE decode (A15-A18) are PINNED:
  `tests/golden/make_golden.py` imports the reference modules from /root/reference in the
  build container, runs them on seeded inputs, and commits the input/output vectors under
  `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this file against them (<=2e-5).
* The sampling loop (A2, A4) is pinned the same way with the *reference* denoiser / VAE
  modules driven by the loop below.
* The schedulers (A3) are PARITY UNPINNED: `diffusers` is not vendored in the reference, is not
  version-pinned (`src/requirements.txt:23`) and is not installed here.  The formulas below restate
  diffusers' published DDIM/DDPM `step()` semantics for the options the reference sets in
  `src/configs/modules/scheduler.yaml:5-14` and `modules_novae/scheduler.yaml:16-29`; only
  known-answer properties (timestep lists, alpha endpoints) are checked.

* The CLIP text tower (`clip_text_features`) lives in the third-party `transformers` package
  (unpinned, `src/requirements.txt:9`); it is PINNED against transformers 5.15's `CLIPModel.get_text_features`
  as installed in the build container, loaded with the synthetic weights (`tests/golden/make_golden_clip.py`).

* The T2M evaluator encoders are PINNED to the reference's own modules (t2m_motionenc.py, t2m_textenc.py,
  importable here: torch only) on synthetic weights (`tests/golden/make_golden_t2m.py`); the TM2T metric formulas are
  pinned to the reference's `metrics/utils.py` helpers (same script) - `TM2TMetrics` itself needs torchmetrics, which is
  not installed, so its accumulation loop is restated from `metrics/tm2t.py:77-153`.

Everything is written batch-first ([B, T, D]); the reference is sequence-first, which only
changes strides, not arithmetic.  Each function cites the reference lines it follows.
"""
import math

import torch
import torch.nn.functional as F

EPS_LN = 1e-5
NUM_HEADS = 4


# --------------------------------------------------------------------------- helpers
def block_names(num_layers=9):
    nb = (num_layers - 1) // 2
    return ([f"input_blocks.{i}" for i in range(nb)] + ["middle_block"] +
            [f"output_blocks.{i}" for i in range(nb)])


def sub(sd, prefix):
    """View of a state dict under `prefix.`"""
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


def cast(sd, dtype):
    return {k: v.to(dtype) for k, v in sd.items()}


def linear(x, w, b=None):
    return F.linear(x, w, b)


def layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, EPS_LN)


def max_iter_elements(lengths, frame_per_latent=48):
    """ladiff.py:379 / ladiff_vae.py:292: ceil(len / FRAME_PER_LATENT)."""
    return [int(math.ceil(l / frame_per_latent)) for l in lengths]


def lengths_to_mask(lengths, max_len=None):
    """utils/temos_utils.py:10-17.  True = frame is real."""
    lengths = torch.as_tensor(lengths)
    max_len = int(max_len if max_len else lengths.max())
    return torch.arange(max_len)[None, :] < lengths[:, None]


def count_mask(counts, n):
    """ladiff_denoiser.py:164-171 / ladiff_vae.py:152-159.  True = latent row is real."""
    counts = torch.as_tensor(counts)
    return torch.arange(n)[None, :] < counts[:, None]


def mha(q_in, k_in, v_in, p, key_padding_mask=None, num_heads=NUM_HEADS):
    """torch.nn.MultiheadAttention forward (eval), batch-first.

    q_in [B,Lq,D], k_in/v_in [B,Lk,D]; key_padding_mask [B,Lk] True = ignore.
    Packed in_proj = [Wq;Wk;Wv], scale 1/sqrt(dh) on q, -inf masking,
    softmax over keys, out_proj.
    """
    d = q_in.shape[-1]
    dh = d // num_heads
    w, b = p["in_proj_weight"], p["in_proj_bias"]
    q = linear(q_in, w[:d], b[:d])
    k = linear(k_in, w[d:2 * d], b[d:2 * d])
    v = linear(v_in, w[2 * d:], b[2 * d:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    q = q.view(B, Lq, num_heads, dh).transpose(1, 2) * (1.0 / math.sqrt(dh))
    k = k.view(B, Lk, num_heads, dh).transpose(1, 2)
    v = v.view(B, Lk, num_heads, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)                                  # [B,H,Lq,Lk]
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, Lq, d)
    return linear(o, p["out_proj.weight"], p["out_proj.bias"])


# --------------------------------------------------------------------------- denoiser
def timestep_sinusoid(t, dim=768, max_period=10000.0):
    """tools/embeddings.py:245-285 with flip_sin_to_cos=True, freq_shift=0 -> [cos | sin]."""
    half = dim // 2
    expo = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - 0)
    arg = t.reshape(-1, 1).to(torch.float32) * torch.exp(expo)[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def time_embedding(sd, t, dtype=torch.float32):
    """Timesteps + TimestepEmbedding (tools/embeddings.py:288-322): Linear, SiLU, Linear."""
    e = timestep_sinusoid(t).to(dtype)
    h = F.silu(linear(e, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"]))
    return linear(h, sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])


def stylization(h, emb, p):
    """StylizationBlock.forward (mdiff_transformer.py:152-163). h [B,T,D], emb [B,D]."""
    e = linear(F.silu(emb), p["emb_layers.1.weight"], p["emb_layers.1.bias"])[:, None, :]
    scale, shift = torch.chunk(e, 2, dim=2)
    h = layer_norm(h, p["norm.weight"], p["norm.bias"]) * (1 + scale) + shift
    return linear(F.silu(h), p["out_layers.2.weight"], p["out_layers.2.bias"])


def sa_block(seq, p, key_padding_mask):
    """mdiff_transformer.TransformerEncoderLayer.forward_post (:54-67), relu feed-forward."""
    a = mha(seq, seq, seq, sub(p, "self_attn"), key_padding_mask)
    seq = layer_norm(seq + a, p["norm1.weight"], p["norm1.bias"])
    f = linear(F.relu(linear(seq, p["linear1.weight"], p["linear1.bias"])),
               p["linear2.weight"], p["linear2.bias"])
    return layer_norm(seq + f, p["norm2.weight"], p["norm2.bias"])


def linear_cross_attention(x, xf, emb, p, pad_mask, num_heads=NUM_HEADS):
    """LinearTemporalCrossAttention.forward (mdiff_transformer.py:219-247).

    x [B,T,D] latents, xf [B,N,D] text tokens, emb [B,D] time, pad_mask [B,T] True = padded row.
    """
    B, T, D = x.shape
    N = xf.shape[1]
    H = num_heads
    q = linear(layer_norm(x, p["norm.weight"], p["norm.bias"]), p["query.weight"], p["query.bias"])
    xn = layer_norm(xf, p["text_norm.weight"], p["text_norm.bias"])
    k = linear(xn, p["key.weight"], p["key.bias"])
    v = linear(xn, p["value.weight"], p["value.bias"]).view(B, N, H, -1)
    q = torch.softmax(q.view(B, T, H, -1), dim=-1)
    k = torch.softmax(k.view(B, N, H, -1), dim=1)
    att = torch.einsum("bnhd,bnhl->bhdl", k, v)
    if pad_mask is not None:
        q = q * (~pad_mask).to(q.dtype)[:, :, None, None]       # :224-225, :242-243
    y = torch.einsum("bnhd,bhdl->bnhl", q, att).reshape(B, T, D)
    return x + stylization(y, emb, sub(p, "proj_out"))


def ffn_block(x, emb, p):
    """FFN.forward (mdiff_transformer.py:259-262): GELU(erf) MLP then StylizationBlock, residual."""
    y = linear(F.gelu(linear(x, p["linear1.weight"], p["linear1.bias"])),
               p["linear2.weight"], p["linear2.bias"])
    return x + stylization(y, emb, sub(p, "proj_out"))


def denoiser_layer(x, xf, emb, p, pad_mask):
    """LinearTemporalDiffusionTransformerDecoderLayer.forward (mdiff_transformer.py:294-321)."""
    B, T, _ = x.shape
    seq = torch.cat([x, xf, emb[:, None, :]], dim=1)            # [B,T+N+1,D]  :308-311
    kpm = None
    if pad_mask is not None:
        kpm = torch.cat([pad_mask, torch.zeros(B, seq.shape[1] - T, dtype=torch.bool)], dim=1)
    x = sa_block(seq, sub(p, "sa_block"), kpm)[:, :T]           # :312-313
    x = linear_cross_attention(x, xf, emb, sub(p, "ca_block"), pad_mask)
    return ffn_block(x, emb, sub(p, "ffn"))


def skip_encoder(x, xf, emb, p, pad_mask, num_layers=9):
    """SkipTransformerEncoder.forward, MD_trans branch (cross_attention.py:69-85)."""
    nb = (num_layers - 1) // 2
    xs = []
    for i in range(nb):
        x = denoiser_layer(x, xf, emb, sub(p, f"input_blocks.{i}"), pad_mask)
        xs.append(x)
    x = denoiser_layer(x, xf, emb, sub(p, "middle_block"), pad_mask)
    for i in range(nb):
        x = torch.cat([x, xs.pop()], dim=-1)
        x = linear(x, p[f"linear_blocks.{i}.weight"], p[f"linear_blocks.{i}.bias"])
        x = denoiser_layer(x, xf, emb, sub(p, f"output_blocks.{i}"), pad_mask)
    return layer_norm(x, p["norm.weight"], p["norm.bias"])


def denoiser_forward(sd, sample, timestep, encoder_hidden_states, max_iter_elems=None,
                     num_layers=9):
    """LADiffDenoiser.forward, text / trans_enc / MD_TRANS branch (ladiff_denoiser.py:153-295).

    sample [B2,T,D]; timestep scalar; encoder_hidden_states [B2,N,E]; max_iter_elems [B2] or None.
    Returns [B2,T,D] (the reference returns it wrapped in a 1-tuple).
    """
    dtype = sample.dtype
    B, T, D = sample.shape
    pad_mask = None
    if max_iter_elems is not None:
        pad_mask = ~count_mask(max_iter_elems, T)                # :164-171, :254
    t = torch.as_tensor(timestep).reshape(1).expand(B)           # :184
    emb = time_embedding(sd, t, dtype)                           # [B,D]  :185-188
    xf = linear(F.relu(encoder_hidden_states), sd["emb_proj.1.weight"], sd["emb_proj.1.bias"])  # :198
    x = sample + sd["query_pos.pe"][:T, 0][None]                 # :251, position_encoding.py:158
    return skip_encoder(x, xf, emb, sub(sd, "encoder"), pad_mask, num_layers)


# --------------------------------------------------------------------------- LA-VAE decoder
def decoder_layer(tgt, memory, p, tgt_pad, mem_pad):
    """cross_attention.TransformerDecoderLayer.forward_post (:358-413), gelu feed-forward."""
    a = mha(tgt, tgt, tgt, sub(p, "self_attn"), tgt_pad)
    tgt = layer_norm(tgt + a, p["norm1.weight"], p["norm1.bias"])
    c = mha(tgt, memory, memory, sub(p, "multihead_attn"), mem_pad)
    tgt = layer_norm(tgt + c, p["norm2.weight"], p["norm2.bias"])
    f = linear(F.gelu(linear(tgt, p["linear1.weight"], p["linear1.bias"])),
               p["linear2.weight"], p["linear2.bias"])
    return layer_norm(tgt + f, p["norm3.weight"], p["norm3.bias"])


def skip_decoder(x, memory, p, tgt_pad, mem_pad, num_layers=9):
    """SkipTransformerDecoder.forward (cross_attention.py:113-153)."""
    nb = (num_layers - 1) // 2
    xs = []
    for i in range(nb):
        x = decoder_layer(x, memory, sub(p, f"input_blocks.{i}"), tgt_pad, mem_pad)
        xs.append(x)
    x = decoder_layer(x, memory, sub(p, "middle_block"), tgt_pad, mem_pad)
    for i in range(nb):
        x = torch.cat([x, xs.pop()], dim=-1)
        x = linear(x, p[f"linear_blocks.{i}.weight"], p[f"linear_blocks.{i}.bias"])
        x = decoder_layer(x, memory, sub(p, f"output_blocks.{i}"), tgt_pad, mem_pad)
    return layer_norm(x, p["norm.weight"], p["norm.bias"])


def vae_decode(sd, z, lengths, frame_per_latent=48, num_layers=9, latent_counts=None):
    """LADiffVae.decode, encoder_decoder / mld-PE branch (ladiff_vae.py:288-362).

    z [max_it,B,D] (sequence-first, as the reference passes it), lengths list -> [B,max(len),C].
    `latent_counts` overrides ceil(len/48) (used for latentwise_gen="fw", :295).
    """
    dtype = z.dtype
    mask = lengths_to_mask(lengths)                              # [B,F]
    counts = max_iter_elements(lengths, frame_per_latent) if latent_counts is None else latent_counts
    B, Fr = mask.shape
    mem = z.permute(1, 0, 2)                                     # [B,max_it,D]
    mem_pad = ~count_mask(counts, mem.shape[1])
    q = torch.zeros(B, Fr, z.shape[-1], dtype=dtype) + sd["query_pos_decoder.pe"][:Fr, 0][None]  # :299,:334
    out = skip_decoder(q, mem, sub(sd, "decoder"), ~mask, mem_pad, num_layers)
    out = linear(out, sd["final_layer.weight"], sd["final_layer.bias"])   # :356
    out = out * mask[:, :, None].to(dtype)                       # :358
    return out


# --------------------------------------------------------------------------- schedulers (A3, restated; parity unpinned)
class _SchedulerBase:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        self.num_train_timesteps = num_train_timesteps
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                    dtype=torch.float32) ** 2    # 'scaled_linear'
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.init_noise_sigma = 1.0
        self.timesteps = None


class DDIM(_SchedulerBase):
    """diffusers.DDIMScheduler with clip_sample=False, set_alpha_to_one=False, steps_offset=1."""

    def __init__(self, steps_offset=1, **kw):
        super().__init__(**kw)
        self.steps_offset = steps_offset
        self.final_alpha_cumprod = self.alphas_cumprod[0]        # set_alpha_to_one: false

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        self.timesteps = torch.arange(n - 1, -1, -1, dtype=torch.int64) * ratio + self.steps_offset

    def step(self, eps, t, x, eta=0.0, noise=None):
        t = int(t)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t].to(x.dtype)
        a_p = (self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod).to(x.dtype)
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        var = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
        sigma = eta * var ** 0.5
        direction = (1 - a_p - sigma ** 2) ** 0.5 * eps
        out = a_p ** 0.5 * x0 + direction
        if eta > 0:
            out = out + sigma * noise
        return out


class DDPM(_SchedulerBase):
    """diffusers.DDPMScheduler, variance_type fixed_small, clip_sample False.

    `prev_timestep`: "t-1" = diffusers <= 0.14 (the reference's era: alpha_prod_t_prev = alphas_cumprod[t-1],
    beta_t = betas[t]); "schedule" = later diffusers (prev_t = t - num_train // num_inference,
    beta_t = 1 - a_t / a_prev).  The two coincide when num_inference_steps == num_train_timesteps, which is how the
    reference runs DDPM (modules_novae/scheduler.yaml:16-29)."""

    def __init__(self, prev_timestep="t-1", **kw):
        super().__init__(**kw)
        self.prev_timestep = prev_timestep

    def set_timesteps(self, n):
        n = min(self.num_train_timesteps, n)
        self.num_inference_steps = n
        self.timesteps = torch.arange(0, self.num_train_timesteps, self.num_train_timesteps // n,
                                      dtype=torch.int64).flip(0)

    def step(self, eps, t, x, noise=None):
        t = int(t)
        a_t = self.alphas_cumprod[t].to(x.dtype)
        if self.prev_timestep == "schedule":
            prev = t - self.num_train_timesteps // self.num_inference_steps
            a_p = (self.alphas_cumprod[prev] if prev >= 0 else torch.tensor(1.0)).to(x.dtype)
            b_t = 1 - a_t / a_p
        else:
            a_p = (self.alphas_cumprod[t - 1] if t > 0 else torch.tensor(1.0)).to(x.dtype)
            b_t = self.betas[t].to(x.dtype)
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        out = (a_p ** 0.5 * b_t / (1 - a_t)) * x0 + ((1 - b_t) ** 0.5 * (1 - a_p) / (1 - a_t)) * x
        if t > 0:
            var = torch.clamp((1 - a_p) / (1 - a_t) * b_t, min=1e-20)
            out = out + var ** 0.5 * noise
        return out


# --------------------------------------------------------------------------- per-step noise drawn on the device
# The product draws the stochastic schedulers' noise where it is consumed (csrc/noise_gen.h) instead of reading a tensor that
# `scheduler.step` would have drawn with randn_tensor (ladiff.py:492; diffusers' DDPMScheduler.step / DDIMScheduler.step with eta > 0).
# This is the numpy restatement of that generator: the same integers (Philox4x32-10, pinned to the Random123 known-answer vectors in
# tests/test_noise.py), the same fp32 Box-Muller; ln / cos / sin come from numpy instead of the device library (last-bit differences).
def philox4x32_10(counter, key):
    """counter [..., 4] uint32, key [..., 2] uint32 (broadcastable) -> [..., 4] uint32 (Salmon et al., SC'11)."""
    import numpy as np
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k = [np.asarray(key[..., i], dtype=np.uint64) for i in range(2)]
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & m32, p1 >> np.uint64(32), p1 & m32
        c = [hi1 ^ c[1] ^ k[0], lo1, hi0 ^ c[3] ^ k[1], lo0]
        k = [(k[0] + np.uint64(0x9E3779B9)) & m32, (k[1] + np.uint64(0xBB67AE85)) & m32]
    return np.stack(c, axis=-1).astype(np.uint32)


def device_noise(seed, first_prompt, first_step, n_steps, B, T, dim=256):
    """[n_steps, B, T, dim] float32: what csrc/noise_gen.h draws for schedule positions first_step .., global prompts first_prompt .."""
    import numpy as np
    st, b, t, ch = np.meshgrid(np.arange(n_steps, dtype=np.uint64) + np.uint64(first_step),
                               np.arange(B, dtype=np.uint64) + np.uint64(first_prompt),
                               np.arange(T, dtype=np.uint64), np.arange(dim // 4, dtype=np.uint64), indexing="ij")
    counter = np.stack([t * np.uint64(64) + ch, b & np.uint64(0xFFFFFFFF), st, np.zeros_like(st)], axis=-1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    x = philox4x32_10(counter, key)
    u = ((x >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    out = np.empty(x.shape, dtype=np.float32)
    for h in range(2):
        r = np.sqrt(np.float32(-2.0) * np.log(u[..., 2 * h]), dtype=np.float32)
        th = np.float32(6.2831854820251465) * u[..., 2 * h + 1]
        out[..., 2 * h] = r * np.cos(th, dtype=np.float32)
        out[..., 2 * h + 1] = r * np.sin(th, dtype=np.float32)
    return out.reshape(n_steps, B, T, dim)


# --------------------------------------------------------------------------- sampling loop (A2, A4)
def diffusion_reverse(denoise_fn, scheduler, text_emb, lengths, init_noise, n_steps,
                      guidance_scale=7.5, eta=0.0, step_noise=None, frame_per_latent=48, test_efficiency=False):
    """LADIFF._diffusion_reverse, live branch (ladiff.py:379-390, 407-417, 470-500, 562-566).

    denoise_fn(sample[2B,T,D], t, text[2B,N,E], counts[2B]) -> eps[2B,T,D]
    text_emb [2B,N,E] with the unconditional half FIRST (ladiff.py:258-264);
    init_noise [B,T,D] stands in for torch.randn (:380-385); padded rows are zeroed here (:389-390).
    step_noise [n_steps,B,T,D] feeds DDPM / eta>0 variance noise.  Returns [T,B,D].
    guidance_scale <= 1: no classifier-free guidance (`do_classifier_free_guidance` False, ladiff.py:472-490): text_emb is
    [B,N,E] and the network runs on the B latents.  test_efficiency (TEST_EFFICIENCY ablation): T = counts[0] latent rows
    (:381), no zeroing of the initial noise (:386-390), no masks in the denoiser (ladiff_denoiser.py:254) - the final
    zeroing (:559-566) still applies.
    """
    counts = max_iter_elements(lengths, frame_per_latent)
    B, T, _ = init_noise.shape
    valid = count_mask(counts, T)
    cfg = guidance_scale > 1.0
    latents = init_noise if test_efficiency else init_noise * valid[:, :, None].to(init_noise.dtype)
    latents = latents * scheduler.init_noise_sigma
    scheduler.set_timesteps(n_steps)
    counts2 = None if test_efficiency else torch.tensor(counts + counts if cfg else counts)
    for i, t in enumerate(scheduler.timesteps):
        if cfg:
            model_in = torch.cat([latents, latents], dim=0)
            eps = denoise_fn(model_in, t, text_emb, counts2)
            eps_u, eps_c = eps.chunk(2)
            eps = eps_u + guidance_scale * (eps_c - eps_u)
        else:
            eps = denoise_fn(latents, t, text_emb, counts2)
        nz = None if step_noise is None else step_noise[i]
        if isinstance(scheduler, DDIM):
            latents = scheduler.step(eps, t, latents, eta=eta, noise=nz)
        else:
            latents = scheduler.step(eps, t, latents, noise=nz)
    latents = latents.permute(1, 0, 2).clone()                   # :500
    latents = latents * valid.t()[:, :, None].to(latents.dtype)  # :562-566
    return latents


def sample_motions(den_sd, vae_sd, text_emb, lengths, init_noise, n_steps=50, scheduler="ddim",
                   guidance_scale=7.5, eta=0.0, step_noise=None, dtype=torch.float32, test_efficiency=False):
    """Whole hot path: _diffusion_reverse + vae.decode (ladiff.py:266, :283). -> (z[T,B,D], feats[B,F,C])."""
    den_sd, vae_sd = cast(den_sd, dtype), cast(vae_sd, dtype)
    sch = DDIM() if scheduler == "ddim" else (DDPM() if scheduler == "ddpm" else scheduler)
    fn = lambda x, t, txt, counts: denoiser_forward(den_sd, x, t, txt, counts)
    z = diffusion_reverse(fn, sch, text_emb.to(dtype), lengths, init_noise.to(dtype), n_steps,
                          guidance_scale, eta, None if step_noise is None else step_noise.to(dtype),
                          test_efficiency=test_efficiency)
    # TEST_EFFICIENCY: the decoder gets no memory mask either (ladiff_vae.py:292-297 builds it only when not test_efficiency)
    feats = vae_decode(vae_sd, z, lengths, latent_counts=[z.shape[0]] * len(lengths) if test_efficiency else None)
    return z, feats


# --------------------------------------------------------------------------- feats2joints
def feats2joints(features, mean, std, njoints):
    """HumanML3DDataModule.feats2joints (data/HumanML3D.py:44-48): de-normalise, then recover_from_ric
    (data/humanml/scripts/motion_process.py:362-381, :415-430; qinv/qrot quaternion.py:16-20, :54-73).
    features [B,F,C] -> joints [B,F,njoints,3].  Pinned by tests/golden/feats2joints_*.npz."""
    data = features * std + mean
    rot_vel = data[..., 0]
    ang = torch.zeros_like(rot_vel)
    ang[..., 1:] = rot_vel[..., :-1]
    ang = torch.cumsum(ang, dim=-1)
    qw, qy = torch.cos(ang), -torch.sin(ang)                 # qinv(r_rot_quat) = (cos, 0, -sin, 0)

    def qrot_y(vx, vy, vz):                                  # qvec = (0, qy, 0)
        ux, uy, uz = qy * vz, torch.zeros_like(vx), -qy * vx
        wx, wy, wz = qy * uz, torch.zeros_like(vx), -qy * ux
        return vx + 2 * (qw * ux + wx), vy + 2 * (qw * uy + wy), vz + 2 * (qw * uz + wz)

    vx = torch.zeros_like(rot_vel); vz = torch.zeros_like(rot_vel)
    vx[..., 1:] = data[..., :-1, 1]
    vz[..., 1:] = data[..., :-1, 2]
    px, _, pz = qrot_y(vx, torch.zeros_like(vx), vz)
    rx, rz = torch.cumsum(px, dim=-1), torch.cumsum(pz, dim=-1)
    ry = data[..., 3]
    loc = data[..., 4:4 + (njoints - 1) * 3].reshape(*data.shape[:-1], njoints - 1, 3)
    qw, qy = qw[..., None], qy[..., None]
    ox, oy, oz = qrot_y(loc[..., 0], loc[..., 1], loc[..., 2])
    pos = torch.stack([ox + rx[..., None], oy, oz + rz[..., None]], dim=-1)
    root = torch.stack([rx, ry, rz], dim=-1)[..., None, :]
    return torch.cat([root, pos], dim=-2)


# --------------------------------------------------------------------------- LA-VAE encoder
def detr_encoder_layer(x, p, key_pad):
    """cross_attention.TransformerEncoderLayer.forward_post (:293-307), gelu feed-forward."""
    a = mha(x, x, x, sub(p, "self_attn"), key_pad)
    x = layer_norm(x + a, p["norm1.weight"], p["norm1.bias"])
    f = linear(F.gelu(linear(x, p["linear1.weight"], p["linear1.bias"])), p["linear2.weight"], p["linear2.bias"])
    return layer_norm(x + f, p["norm2.weight"], p["norm2.bias"])


def vae_encode(sd, features, lengths, eps, max_it=5, frame_per_latent=48, num_layers=9):
    """LADiffVae.encode, LAD / mld-PE branch (ladiff_vae.py:162-286).  features [B,F,C] -> (mu, std, latent), each
    [max_it,B,D]; `eps` [max_it,B,D] stands in for Normal.rsample's draw.  Pinned by tests/golden/vae_encode_*.npz."""
    B, Fr, _ = features.shape
    mask = lengths_to_mask(lengths, Fr)                                                  # :178
    x = linear(features, sd["skel_embedding.weight"], sd["skel_embedding.bias"])         # :182
    tok = sd["global_motion_token"][None].expand(B, -1, -1)                              # :189
    counts = max_iter_elements(lengths, frame_per_latent)                                # :198
    dm = count_mask(counts, max_it)
    aug = torch.cat([dm, dm, mask], dim=1)                                               # :203-209
    xseq = torch.cat([tok, x], dim=1)
    xseq = xseq + sd["query_pos_encoder.pe"][:xseq.shape[1], 0][None]                    # :219
    p = sub(sd, "encoder")
    nb = (num_layers - 1) // 2
    xs, h = [], xseq
    for i in range(nb):                                                                  # cross_attention.py:48-67
        h = detr_encoder_layer(h, sub(p, f"input_blocks.{i}"), ~aug)
        xs.append(h)
    h = detr_encoder_layer(h, sub(p, "middle_block"), ~aug)
    for i in range(nb):
        h = linear(torch.cat([h, xs.pop()], dim=-1), p[f"linear_blocks.{i}.weight"], p[f"linear_blocks.{i}.bias"])
        h = detr_encoder_layer(h, sub(p, f"output_blocks.{i}"), ~aug)
    h = layer_norm(h, p["norm.weight"], p["norm.bias"])
    dist = h[:, :2 * max_it].permute(1, 0, 2)                                            # :221
    mu, logvar = dist[:max_it], dist[max_it:]                                            # :258-259
    std = logvar.exp().pow(0.5)                                                          # :262
    latent = (mu + std * eps) * dm.t()[:, :, None].to(mu.dtype)                          # :264-268
    return mu, std, latent


# --------------------------------------------------------------------------- CLIP text tower
def clip_text_features(sd, input_ids, num_layers=12, num_heads=12):
    """`MldTextEncoder.forward`, "clip" branch (`mld_clip.py:51-78`): `text_model.get_text_features(input_ids)` with no
    attention mask -> [B, 768]  (the reference unsqueezes to [B, 1, 768] at `:78`).  Restates transformers'
    CLIPTextTransformer: token + position embedding, pre-LN layers under a causal mask, quick_gelu MLP, final LayerNorm,
    the row at argmax(input_ids) (= first EOS with the CLIP vocabulary), text_projection without bias."""
    B, S = input_ids.shape
    x = sd["text_model.embeddings.token_embedding.weight"][input_ids] + \
        sd["text_model.embeddings.position_embedding.weight"][:S]
    W = x.shape[-1]
    dh = W // num_heads
    causal = torch.full((S, S), float("-inf"), dtype=x.dtype).triu(1)
    for i in range(num_layers):
        p = f"text_model.encoder.layers.{i}."
        h = F.layer_norm(x, (W,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], EPS_LN)
        q = linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * dh ** -0.5
        k = linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = (t.view(B, S, num_heads, dh).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) + causal, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, S, W)
        x = x + linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (W,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], EPS_LN)
        h = linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        h = h * torch.sigmoid(1.702 * h)                                           # quick_gelu
        x = x + linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    x = F.layer_norm(x, (W,), sd["text_model.final_layer_norm.weight"], sd["text_model.final_layer_norm.bias"], EPS_LN)
    pooled = x[torch.arange(B), input_ids.argmax(dim=-1)]
    return pooled @ sd["text_projection.weight"].t()


# --------------------------------------------------------------------------- T2M evaluators + TM2T metrics
def t2m_movement_encoder(sd, feats):
    """`MovementConvEncoder.forward` (t2m_motionenc.py:21-25) on `feats[..., :-4]` (ladiff.py:1264): [B,F,C] -> [B,F/4,512]."""
    x = feats[..., :-4].permute(0, 2, 1)
    x = F.leaky_relu(F.conv1d(x, sd["main.0.weight"], sd["main.0.bias"], stride=2, padding=1), 0.2)
    x = F.leaky_relu(F.conv1d(x, sd["main.3.weight"], sd["main.3.bias"], stride=2, padding=1), 0.2)
    return linear(x.permute(0, 2, 1), sd["out_net.weight"], sd["out_net.bias"])


def gru_bidir_last(sd, x, lens, h0):
    """Final hidden states of a bidirectional one-layer nn.GRU run over `pack_padded_sequence(x, lens)`: a sample's state
    advances over its own first `len` steps only (forward 0..len-1, backward len-1..0).  -> [B, 2H]"""
    B, T, _ = x.shape
    outs = []
    for d, sfx in enumerate(("", "_reverse")):
        wi, wh = sd[f"gru.weight_ih_l0{sfx}"], sd[f"gru.weight_hh_l0{sfx}"]
        bi, bh = sd[f"gru.bias_ih_l0{sfx}"], sd[f"gru.bias_hh_l0{sfx}"]
        H = wh.shape[1]
        h = h0[d].expand(B, H).clone()
        gi_all = linear(x, wi, bi)
        for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
            gi, gh = gi_all[:, t], linear(h, wh, bh)
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            hn = (1 - z) * n + z * h
            h = torch.where((t < torch.as_tensor(lens)).unsqueeze(1), hn, h)
        outs.append(h)
    return torch.cat(outs, dim=-1)


def _coemb_head(sd, x):
    x = linear(x, sd["output_net.0.weight"], sd["output_net.0.bias"])
    x = F.leaky_relu(layer_norm(x, sd["output_net.1.weight"], sd["output_net.1.bias"]), 0.2)
    return linear(x, sd["output_net.3.weight"], sd["output_net.3.bias"])


def t2m_motion_encoder(sd, movements, m_lens):
    """`MotionEncoderBiGRUCo.forward` (t2m_motionenc.py:51-64): [B,T,512], lengths -> [B,512]."""
    emb = linear(movements, sd["input_emb.weight"], sd["input_emb.bias"])
    return _coemb_head(sd, gru_bidir_last(sd, emb, m_lens, sd["hidden"][:, 0]))


def t2m_text_encoder(sd, word_embs, pos_onehot, cap_lens):
    """`TextEncoderBiGRUCo.forward` (t2m_textenc.py:32-48): [B,L,300], [B,L,15], lengths -> [B,512]."""
    inputs = word_embs + linear(pos_onehot, sd["pos_emb.weight"], sd["pos_emb.bias"])
    emb = linear(inputs, sd["input_emb.weight"], sd["input_emb.bias"])
    return _coemb_head(sd, gru_bidir_last(sd, emb, cap_lens, sd["hidden"][:, 0]))


def tm2t_metrics(text_emb, gen_emb, gt_emb, order, div_first, div_second, R_size=32, top_k=3):
    """`TM2TMetrics.compute` (metrics/tm2t.py:77-153) with its random draws made explicit: `order` = the shuffle of the
    sequences, `div_first` / `div_second` = the index pairs of the diversity estimate (metrics/utils.py:230-244).
    Embeddings are [N, 512] tensors; returns a dict of python floats."""
    import numpy as np
    import scipy.linalg
    text, gen, gt = (t[order].double() for t in (text_emb, gen_emb, gt_emb))
    N = text.shape[0]
    out = {}
    for tag, mot in (("", gen), ("gt_", gt)):
        match, hits = 0.0, torch.zeros(top_k)
        for i in range(N // R_size):
            a, b = text[i * R_size:(i + 1) * R_size], mot[i * R_size:(i + 1) * R_size]
            d = torch.sqrt(-2 * a @ b.T + (a * a).sum(1, keepdim=True) + (b * b).sum(1)).nan_to_num()   # utils.py:26-41
            match += d.trace().item()
            rank = torch.argsort(d, dim=1)
            hit = torch.zeros(R_size, dtype=torch.bool)
            for k in range(top_k):                                                                        # utils.py:62-75
                hit = hit | (rank[:, k] == torch.arange(R_size))
                hits[k] += hit.sum()
        R = N // R_size * R_size
        out[tag + "Matching_score"] = match / R
        for k in range(top_k):
            out[f"{tag}R_precision_top_{k + 1}"] = (hits[k] / R).item()
    g, t = gen.numpy(), gt.numpy()
    mu, cov, mu_t, cov_t = g.mean(0), np.cov(g, rowvar=False), t.mean(0), np.cov(t, rowvar=False)
    covmean = scipy.linalg.sqrtm(cov_t.dot(cov))                                                         # utils.py:161-211
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    diff = mu_t - mu
    out["FID"] = float(diff.dot(diff) + np.trace(cov_t) + np.trace(cov) - 2 * np.trace(covmean))
    out["Diversity"] = float(np.linalg.norm(g[div_first] - g[div_second], axis=1).mean())
    out["gt_Diversity"] = float(np.linalg.norm(t[div_first] - t[div_second], axis=1).mean())
    return out
