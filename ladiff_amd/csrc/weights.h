// Weight tables: structs of device pointers laid out in the SAME order as the state-dict key lists
// returned by ladiff_{denoiser,decoder}_param_name (and ladiff_amd/schema.py), so a pointer array from
// the host maps onto them by a plain copy.  Every member is one `const float*`.
#pragma once
#include <string>
#include <vector>

#include "common.h"

namespace ladiff {

constexpr int NL = LADIFF_NUM_LAYERS;        // 9 = 4 input + middle + 4 output blocks (cross_attention.py:29-33)
constexpr int NSKIP = (NL - 1) / 2;
constexpr int FF = LADIFF_FF_SIZE;
constexpr int TEXT_DIM = LADIFF_TEXT_DIM;

struct LinearW { const float *w, *b; };
struct NormW { const float *g, *b; };
struct MhaW { const float *in_w, *in_b, *out_w, *out_b; };          // packed [Wq;Wk;Wv], nn.MultiheadAttention
struct StylW { LinearW emb; NormW norm; LinearW out; };             // StylizationBlock, mdiff_transformer.py:137-150

struct DenLayerW {                                                   // mdiff_transformer.py:265-291
    NormW ca_norm, ca_text_norm;
    LinearW ca_query, ca_key, ca_value;
    StylW ca_proj;
    LinearW ffn1, ffn2;
    StylW ffn_proj;
    MhaW sa_attn;
    LinearW sa_lin1, sa_lin2;
    NormW sa_norm1, sa_norm2;
};

struct DenoiserW {                                                   // ladiff_denoiser.py:62-123
    LinearW time1, time2, emb_proj;
    const float *query_pe, *mem_pe;
    NormW norm;
    DenLayerW layer[NL];
    LinearW skip[NSKIP];
};

struct DecLayerW {                                                   // cross_attention.py:332-353
    MhaW self_attn, cross_attn;
    LinearW lin1, lin2;
    NormW norm1, norm2, norm3;
};

struct DecoderW {                                                    // ladiff_vae.py:74-75, 96-106, 123
    const float* query_pe;
    DecLayerW layer[NL];
    LinearW skip[NSKIP];
    NormW norm;
    LinearW final_layer;
};

struct EncLayerW {                                                   // cross_attention.py:264-286 (DETR encoder layer)
    MhaW self_attn;
    LinearW lin1, lin2;
    NormW norm1, norm2;
};

struct EncoderW {                                                    // ladiff_vae.py:72-73, 79-89, 116-122 (encode side)
    const float* motion_token;                                       // global_motion_token [2*MAX_IT, 256]
    const float* query_pe;                                           // query_pos_encoder.pe
    EncLayerW layer[NL];
    LinearW skip[NSKIP];
    NormW norm;
    LinearW skel;                                                    // skel_embedding [256, nfeats]
};

struct ClipLayerW {                                                  // transformers CLIPEncoderLayer (mld_clip.py:29, :76)
    NormW ln1;
    LinearW q, k, v, o;
    NormW ln2;
    LinearW fc1, fc2;
};

struct ClipW {                                                       // text side of CLIPModel + text_projection
    const float* tok;                                                // token_embedding [vocab, 768]
    const float* pos;                                                // position_embedding [77, 768]
    NormW final_ln;
    const float* proj;                                               // text_projection.weight [768, 768], no bias
    ClipLayerW layer[CLIP_MAX_LAYERS];                               // a shallower model fills a prefix (rest NULL)
};

constexpr int DEN_NPARAMS = sizeof(DenoiserW) / sizeof(const float*);
constexpr int DEC_NPARAMS = sizeof(DecoderW) / sizeof(const float*);
constexpr int ENC_NPARAMS = sizeof(EncoderW) / sizeof(const float*);
constexpr int CLIP_NPARAMS = sizeof(ClipW) / sizeof(const float*);
constexpr int CLIP_HEAD_NPARAMS = 5, CLIP_LAYER_NPARAMS = sizeof(ClipLayerW) / sizeof(const float*);

const std::vector<std::string>& denoiser_param_names();
const std::vector<std::string>& decoder_param_names();
const std::vector<std::string>& encoder_param_names();
const std::vector<std::string>& clip_param_names();

}  // namespace ladiff
