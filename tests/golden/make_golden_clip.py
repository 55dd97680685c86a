#!/usr/bin/env python3
"""Golden vectors for the CLIP text tower from the third-party implementation the reference calls.

`MldTextEncoder` (mld_clip.py:29, :75-76) runs `transformers.AutoModel.from_pretrained(clip-vit-large-patch14)
.get_text_features(input_ids)`.  There are no pretrained weights offline, so this script builds transformers'
`CLIPModel` (the class AutoModel resolves to; text tower at the ViT-L/14 geometry, a token vision tower that is never
run) with the synthetic weights of `ladiff_amd.synthetic.clip_weights` and records ids -> text features.
Run in the build container only (`python tests/golden/make_golden_clip.py`); fixtures hold ids and outputs, no weights.
"""
import os, sys
import numpy as np
import torch
import transformers
from transformers import CLIPConfig, CLIPModel

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ladiff_amd import synthetic as syn      # noqa: E402

torch.set_num_threads(8)


def build(vocab, layers, eos_id):
    text = dict(vocab_size=vocab, hidden_size=768, intermediate_size=3072, projection_dim=768, num_hidden_layers=layers,
                num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                bos_token_id=vocab - 2, eos_token_id=eos_id, pad_token_id=1)
    vision = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, image_size=28,
                  patch_size=14, projection_dim=768)
    m = CLIPModel(CLIPConfig(text_config=text, vision_config=vision, projection_dim=768)).eval()
    sd = syn.clip_weights(vocab, layers)
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith(("vision_model.", "visual_projection.", "logit_scale")) or "position_ids" in k
               for k in res.missing_keys), res.missing_keys
    return m


with torch.no_grad():
    #  name, vocab, layers, eos_token_id in the config (2 = the hub config's legacy value -> argmax pooling), batch, empty
    for name, vocab, layers, eos_id, batch, empty in (("clip_small", 512, 2, 2, 6, 1), ("clip_small_eos", 512, 2, 511, 3, 0),
                                                      ("clip_full", 49408, 12, 2, 4, 1)):
        m = build(vocab, layers, eos_id)
        ids = syn.clip_token_ids(batch, vocab, max_words=75 if name == "clip_small" else 30, empty_first=empty)
        out = m.get_text_features(ids)
        out = out if torch.is_tensor(out) else out.pooler_output
        np.savez_compressed(os.path.join(HERE, name + ".npz"), ids=ids.numpy(), text_features=out.numpy(),
                            vocab=np.array(vocab), layers=np.array(layers),
                            transformers_version=np.array(transformers.__version__))
        print(name, tuple(out.shape), float(out.abs().max()), (ids.argmax(-1)).tolist())
