"""Parameter schemas (state-dict keys and shapes) of the two hot-path networks.

These are the drop-in contract of SURVEY.md Appendix A: a checkpoint written by the
reference (`demo.py:138-159` loads ``ckpt["state_dict"]`` strictly) must load into the
modules of this package, so names and shapes follow what the reference modules register:

* denoiser  - `src/ladiff/models/architectures/ladiff_denoiser.py:62-123` (time_embedding,
  emb_proj, query_pos/mem_pos, SkipTransformerEncoder of
  `mdiff_transformer.py:265-291` layers)
* LA-VAE    - `src/ladiff/models/architectures/ladiff_vae.py:66-123` (PEs, encoder stack,
  SkipTransformerDecoder of `cross_attention.py:332-353` layers, skel_embedding, final_layer)

The schemas are also what the C-ABI library consumes: `csrc/weights.h` lists the same
names in the same order (checked by tests/test_abi.py).
"""
from collections import OrderedDict
from types import SimpleNamespace

PE_MAX_LEN = 500  # position_encoding.py:140


def block_names(num_layers):
    """U-Net style block list of SkipTransformer{Encoder,Decoder} (cross_attention.py:27-33)."""
    assert num_layers % 2 == 1
    nb = (num_layers - 1) // 2
    return ([f"input_blocks.{i}" for i in range(nb)] + ["middle_block"] +
            [f"output_blocks.{i}" for i in range(nb)])


def _linear(out, prefix, n_out, n_in):
    out[prefix + ".weight"] = (n_out, n_in)
    out[prefix + ".bias"] = (n_out,)


def _norm(out, prefix, d):
    out[prefix + ".weight"] = (d,)
    out[prefix + ".bias"] = (d,)


def _mha(out, prefix, d):
    out[prefix + ".in_proj_weight"] = (3 * d, d)
    out[prefix + ".in_proj_bias"] = (3 * d,)
    _linear(out, prefix + ".out_proj", d, d)


def _stylization(out, prefix, d):
    # StylizationBlock, mdiff_transformer.py:137-150
    _linear(out, prefix + ".emb_layers.1", 2 * d, d)
    _norm(out, prefix + ".norm", d)
    _linear(out, prefix + ".out_layers.2", d, d)


def denoiser_schema(latent_dim=256, ff_size=1024, num_layers=9, text_encoded_dim=768,
                    sa_ff_size=1024):
    """Keys/shapes of LADiffDenoiser (text condition, trans_enc, SKIP_CONNECT, MD_TRANS)."""
    d = latent_dim
    s = OrderedDict()
    _linear(s, "time_embedding.linear_1", d, text_encoded_dim)
    _linear(s, "time_embedding.linear_2", d, d)
    if text_encoded_dim != d:
        _linear(s, "emb_proj.1", d, text_encoded_dim)
    s["query_pos.pe"] = (PE_MAX_LEN, 1, d)
    s["mem_pos.pe"] = (PE_MAX_LEN, 1, d)
    _norm(s, "encoder.norm", d)
    for blk in block_names(num_layers):
        p = f"encoder.{blk}"
        _norm(s, p + ".ca_block.norm", d)
        _norm(s, p + ".ca_block.text_norm", d)
        _linear(s, p + ".ca_block.query", d, d)
        _linear(s, p + ".ca_block.key", d, d)
        _linear(s, p + ".ca_block.value", d, d)
        _stylization(s, p + ".ca_block.proj_out", d)
        _linear(s, p + ".ffn.linear1", ff_size, d)
        _linear(s, p + ".ffn.linear2", d, ff_size)
        _stylization(s, p + ".ffn.proj_out", d)
        _mha(s, p + ".sa_block.self_attn", d)
        # sa_block feed-forward is hard-wired to 1024/relu (mdiff_transformer.py:287-288)
        _linear(s, p + ".sa_block.linear1", sa_ff_size, d)
        _linear(s, p + ".sa_block.linear2", d, sa_ff_size)
        _norm(s, p + ".sa_block.norm1", d)
        _norm(s, p + ".sa_block.norm2", d)
    for i in range((num_layers - 1) // 2):
        _linear(s, f"encoder.linear_blocks.{i}", d, 2 * d)
    return s


def _detr_encoder_layer(s, p, d, ff):
    _mha(s, p + ".self_attn", d)
    _linear(s, p + ".linear1", ff, d)
    _linear(s, p + ".linear2", d, ff)
    _norm(s, p + ".norm1", d)
    _norm(s, p + ".norm2", d)


def _detr_decoder_layer(s, p, d, ff):
    _mha(s, p + ".self_attn", d)
    _mha(s, p + ".multihead_attn", d)
    _linear(s, p + ".linear1", ff, d)
    _linear(s, p + ".linear2", d, ff)
    _norm(s, p + ".norm1", d)
    _norm(s, p + ".norm2", d)
    _norm(s, p + ".norm3", d)


def vae_schema(nfeats=263, latent_dim=256, ff_size=1024, num_layers=9, max_it=5):
    """Keys/shapes of LADiffVae (arch encoder_decoder, pe mld, MLP_DIST False)."""
    d = latent_dim
    s = OrderedDict()
    s["global_motion_token"] = (2 * max_it, d)
    s["query_pos_encoder.pe"] = (PE_MAX_LEN, 1, d)
    s["query_pos_decoder.pe"] = (PE_MAX_LEN, 1, d)
    for blk in block_names(num_layers):
        _detr_encoder_layer(s, f"encoder.{blk}", d, ff_size)
    for i in range((num_layers - 1) // 2):
        _linear(s, f"encoder.linear_blocks.{i}", d, 2 * d)
    _norm(s, "encoder.norm", d)
    for blk in block_names(num_layers):
        _detr_decoder_layer(s, f"decoder.{blk}", d, ff_size)
    for i in range((num_layers - 1) // 2):
        _linear(s, f"decoder.linear_blocks.{i}", d, 2 * d)
    _norm(s, "decoder.norm", d)
    _linear(s, "skel_embedding", d, nfeats)
    _linear(s, "final_layer", nfeats, d)
    return s


def vae_decode_keys(schema):
    """The subset `LADiffVae.decode` reads (ladiff_vae.py:334-356)."""
    return [k for k in schema
            if k.startswith("decoder.") or k.startswith("final_layer.") or k == "query_pos_decoder.pe"]


def clip_text_schema(vocab_size=49408, num_layers=12, width=768, max_positions=77):
    """Text side of transformers' CLIPModel for openai/clip-vit-large-patch14 (loaded by `mld_clip.py:29`), keys as in
    `CLIPModel.state_dict()`: what `get_text_features` touches (`mld_clip.py:75-76`).  SURVEY.md §8f-1."""
    s = OrderedDict()
    s["text_model.embeddings.token_embedding.weight"] = (vocab_size, width)
    s["text_model.embeddings.position_embedding.weight"] = (max_positions, width)
    for i in range(num_layers):
        p = f"text_model.encoder.layers.{i}"
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            _linear(s, f"{p}.self_attn.{n}", width, width)
        _norm(s, f"{p}.layer_norm1", width)
        _linear(s, f"{p}.mlp.fc1", 4 * width, width)
        _linear(s, f"{p}.mlp.fc2", width, 4 * width)
        _norm(s, f"{p}.layer_norm2", width)
    _norm(s, "text_model.final_layer_norm", width)
    s["text_projection.weight"] = (width, width)
    return s


def _gru(out, prefix, d_in, d_h):
    for sfx in ("", "_reverse"):
        out[f"{prefix}.weight_ih_l0{sfx}"] = (3 * d_h, d_in)
        out[f"{prefix}.weight_hh_l0{sfx}"] = (3 * d_h, d_h)
        out[f"{prefix}.bias_ih_l0{sfx}"] = (3 * d_h,)
        out[f"{prefix}.bias_hh_l0{sfx}"] = (3 * d_h,)


def _coemb_head(out, d_h, d_out):
    _linear(out, "output_net.0", d_h, 2 * d_h)
    _norm(out, "output_net.1", d_h)
    _linear(out, "output_net.3", d_out, d_h)


def t2m_movement_schema(input_size=259, hidden_size=512, output_size=512):
    """`MovementConvEncoder` (t2m_motionenc.py:6-19): the `movement_encoder` sub-dict of the evaluator checkpoint."""
    s = OrderedDict()
    s["main.0.weight"] = (hidden_size, input_size, 4)
    s["main.0.bias"] = (hidden_size,)
    s["main.3.weight"] = (output_size, hidden_size, 4)
    s["main.3.bias"] = (output_size,)
    _linear(s, "out_net", output_size, output_size)
    return s


def t2m_motion_schema(input_size=512, hidden_size=1024, output_size=512):
    """`MotionEncoderBiGRUCo` (t2m_motionenc.py:28-49): the `motion_encoder` sub-dict."""
    s = OrderedDict()
    _linear(s, "input_emb", hidden_size, input_size)
    _gru(s, "gru", hidden_size, hidden_size)
    _coemb_head(s, hidden_size, output_size)
    s["hidden"] = (2, 1, hidden_size)
    return s


def t2m_text_schema(word_size=300, pos_size=15, hidden_size=512, output_size=512):
    """`TextEncoderBiGRUCo` (t2m_textenc.py:6-30): the `text_encoder` sub-dict."""
    s = OrderedDict()
    _linear(s, "pos_emb", word_size, pos_size)
    _linear(s, "input_emb", hidden_size, word_size)
    _gru(s, "gru", hidden_size, hidden_size)
    _coemb_head(s, hidden_size, output_size)
    s["hidden"] = (2, 1, hidden_size)
    return s


# The shipped configuration (`configs/config_ladiff_humanml3d.yaml` model.ablation + `configs/modules/denoiser.yaml`,
# `configs/modules/motion_vae.yaml` as `get_model` instantiates them): what bench.py, smoke() and the tests build the two networks with.
ABL = SimpleNamespace(SKIP_CONNECT=True, VAE_TYPE="actor", DIFF_PE_TYPE="mld", PE_TYPE="mld", IDEA="ard",
                      MD_TRANS=True, TEST_EFFICIENCY=False, MLP_DIST=False, DVAE=False, PERCENTAGE_NOISED=0.0,
                      MAX_IT=5, FRAME_PER_LATENT=48, JOINT_DISTRO_FIX=False, LAD=True)
DEN_KW = dict(nfeats=263, condition="text", latent_dim=[7, 256], ff_size=1024, num_layers=9, num_heads=4,
              dropout=0.1, normalize_before=False, activation="gelu", flip_sin_to_cos=True,
              return_intermediate_dec=False, position_embedding="learned", arch="trans_enc", freq_shift=0,
              guidance_scale=7.5, guidance_uncondp=0.1, text_encoded_dim=768, nclasses=10)
VAE_KW = dict(nfeats=263, latent_dim=[7, 256], ff_size=1024, num_layers=9, num_heads=4, dropout=0.1,
              arch="encoder_decoder", normalize_before=False, activation="gelu", position_embedding="learned")
