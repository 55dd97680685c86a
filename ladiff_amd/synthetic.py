"""Deterministic synthetic weights and inputs.

There are no checkpoints or datasets offline, so tests, goldens and the benchmark use
random-init weights of the reference architecture.  Weights are regenerated everywhere
from one `numpy.random.RandomState(seed)` stream, filled in *sorted key order*, so the
golden fixtures only have to store inputs/outputs, never weights.

Distributions follow what the reference's constructors leave behind
(`cross_attention.py:37-40` Xavier-uniform on every >=2-D tensor, `position_encoding.py:150-151`
U(0,1) PEs, `ladiff_vae.py:119-120` N(0,1) motion tokens) except that biases and LayerNorm
affine parameters get small non-zero values so that every bias / gamma / beta path is exercised
by the parity tests (the reference leaves most of them at exactly 0 / 1).
"""
import hashlib
import math

import numpy as np
import torch

from . import schema as _schema

WEIGHT_SEED = 1234   # base.yaml:2 SEED_VALUE
TEXT_SEED = 7
NOISE_SEED = 11
DDPM_NOISE_SEED = 13
FRAME_PER_LATENT = 48  # config_ladiff_humanml3d.yaml:59
MAX_IT = 5             # config_ladiff_humanml3d.yaml:58


def _fill(rs, name, shape):
    if name.endswith(".pe"):
        return rs.uniform(0.0, 1.0, size=shape)
    if name == "global_motion_token" or name == "hidden":
        return rs.standard_normal(size=shape)
    if name.endswith(("token_embedding.weight", "position_embedding.weight")):      # CLIP token / position tables: N(0, 0.02) as transformers initialises them
        return 0.02 * rs.standard_normal(size=shape)
    if len(shape) >= 2:
        fan_out, fan_in = shape[0], shape[1]
        a = math.sqrt(6.0 / (fan_in + fan_out))
        return rs.uniform(-a, a, size=shape)
    if name.endswith("weight"):          # LayerNorm gamma
        return 1.0 + 0.02 * rs.standard_normal(size=shape)
    return 0.02 * rs.standard_normal(size=shape)   # biases, LayerNorm beta


def make_state_dict(schema, seed=WEIGHT_SEED, dtype=torch.float32):
    """name -> CPU tensor for every key of `schema`, filled in sorted-key order."""
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(schema):
        out[name] = torch.from_numpy(np.ascontiguousarray(_fill(rs, name, schema[name]))).to(dtype)
    return {k: out[k] for k in schema}


def state_dict_sha256(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().to(torch.float32).contiguous().numpy().tobytes())
    return h.hexdigest()


def denoiser_weights(seed=WEIGHT_SEED, **kw):
    return make_state_dict(_schema.denoiser_schema(**kw), seed)


def vae_weights(nfeats=263, seed=WEIGHT_SEED + 1, **kw):
    return make_state_dict(_schema.vae_schema(nfeats=nfeats, **kw), seed)


def clip_weights(vocab_size=49408, num_layers=12, seed=WEIGHT_SEED + 2):
    return make_state_dict(_schema.clip_text_schema(vocab_size, num_layers), seed)


def clip_token_ids(batch, vocab_size=49408, seq_len=77, max_words=30, seed=TEXT_SEED + 1, empty_first=0):
    """Token ids shaped like the CLIP tokenizer's output with padding="max_length" (mld_clip.py:54-60):
    [BOS, w_1..w_n, EOS, EOS-padding...], BOS = vocab-2, EOS = pad = vocab-1 (49406 / 49407 for the real vocabulary).
    The first `empty_first` rows are the empty prompt "" of the classifier-free branch (ladiff.py:258-262)."""
    rs = np.random.RandomState(seed)
    ids = np.full((batch, seq_len), vocab_size - 1, dtype=np.int64)
    ids[:, 0] = vocab_size - 2
    for b in range(batch):
        n = 0 if b < empty_first else int(rs.randint(1, min(max_words, seq_len - 2) + 1))
        ids[b, 1:1 + n] = rs.randint(0, vocab_size - 2, size=n)
    return torch.from_numpy(ids)


def t2m_weights(nfeats=263, seed=WEIGHT_SEED + 3):
    """(movement, motion, text) evaluator state dicts; GRU / conv tensors get the Xavier-uniform fill of every >= 2-D tensor."""
    return (make_state_dict(_schema.t2m_movement_schema(nfeats - 4), seed),
            make_state_dict(_schema.t2m_motion_schema(), seed + 1),
            make_state_dict(_schema.t2m_text_schema(), seed + 2))


def max_iter_elements(lengths, frame_per_latent=FRAME_PER_LATENT):
    """ceil(len / FRAME_PER_LATENT)  (ladiff.py:379, ladiff_vae.py:292)."""
    return [int(math.ceil(l / frame_per_latent)) for l in lengths]


def text_embeddings(batch, dim=768, seed=TEXT_SEED):
    """[2B,1,dim] fp32; rows [0:B] play the unconditional ("") embeddings (ladiff.py:258-264)."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal((2 * batch, 1, dim)).astype(np.float32))


def init_noise(lengths, max_it=MAX_IT, dim=256, seed=NOISE_SEED, offset=0, total=None):
    """[B,max_it,dim] N(0,1) with rows >= ceil(len/48) zeroed (ladiff.py:380-390).

    `offset/total` slice a global batch: noise is drawn for `total` prompts and rows
    [offset, offset+B) are returned, so results do not depend on how prompts are sharded.
    """
    b = len(lengths)
    total = b if total is None else total
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((total, max_it, dim)).astype(np.float32)[offset:offset + b]
    x = torch.from_numpy(np.ascontiguousarray(x))
    for i, m in enumerate(max_iter_elements(lengths)):
        x[i, m:] = 0
    return x


def ddpm_noise(n_steps, batch, max_it=MAX_IT, dim=256, seed=DDPM_NOISE_SEED):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal((n_steps, batch, max_it, dim)).astype(np.float32))


def mixed_lengths(batch, choices=(60, 120, 196)):
    return (list(choices) * ((batch + len(choices) - 1) // len(choices)))[:batch]
