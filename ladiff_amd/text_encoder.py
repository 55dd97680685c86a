"""Drop-in for `ladiff.models.architectures.mld_clip.MldTextEncoder` ("clip" branch): prompts -> [B, 1, 768].

SURVEY.md §8f-1, the step in front of the sampling path (`ladiff.py:265`).  The reference wraps transformers'
`CLIPModel.get_text_features` (`mld_clip.py:29, :75-78`); here the text tower runs in libladiff_hip.so
(`ladiff_clip_text_encode`, csrc/clip.hip).  Tokenising is host string work and stays with the CLIP tokenizer the
caller supplies (the reference's `AutoTokenizer.from_pretrained(modelpath)` when the files exist).

Two exact host-side savings, both consequences of the causal mask / row independence:
* only positions up to the last EOS in the batch are evaluated (padding behind the EOS cannot influence it);
* identical id rows (the B copies of the empty prompt in the guidance batch, `ladiff.py:258-262`) are encoded once.
"""
import os

import numpy as np
import torch
from torch import nn

from . import _lib, schema
from .modules import _HipModule


class _ClipTextTower(_HipModule):
    """Parameter container with the keys of `transformers.CLIPModel.state_dict()` that `get_text_features` uses."""
    _KIND = "clip"

    def __init__(self, vocab_size, num_layers):
        super().__init__()
        self.vocab_size, self.num_layers = vocab_size, num_layers
        self._build(schema.clip_text_schema(vocab_size, num_layers))

    def _weight_table(self, kind=None):
        sd = dict(self.named_parameters())
        n = 5 + 16 * self.num_layers
        names = _lib.param_names("clip")[:n]
        key = _lib.WeightTable.key_of(names, sd)
        tab = self.__dict__.get("_wt_clip")
        if tab is None or tab.key != key:
            tab = _lib.WeightTable("clip", sd, n_names=n, no_split=(names[0], names[1], names[4]))
            self.__dict__["_wt_clip"] = tab
        return tab


class MldTextEncoder(nn.Module):
    """Same constructor and call as `mld_clip.py:13-90`; keyword-only extras for offline use (no checkpoint on disk)."""

    def __init__(self, modelpath: str = None, finetune: bool = False, last_hidden_state: bool = False,
                 latent_dim: list = [1, 256], *, tokenizer=None, vocab_size: int = 49408, num_layers: int = 12,
                 precision: str = "fp32") -> None:
        super().__init__()
        self.latent_dim = latent_dim
        if modelpath is not None and "bert" in modelpath:
            raise NotImplementedError('the "bert" text encoder (mld_clip.py:44-46) is not built; the shipped config uses CLIP')
        if modelpath is not None and "clip" not in modelpath:
            raise ValueError(f"Model {modelpath} not supported")          # mld_clip.py:47-48
        if finetune:
            raise NotImplementedError("finetune=True needs a backward pass; only the frozen encoder (mld_clip.py:32-35) is built")
        if last_hidden_state:
            raise NotImplementedError('last_hidden_state=True ("clip_hidden", mld_clip.py:79-82) is not built; '
                                      "the shipped config uses the pooled token")
        self.name = "clip"
        self.text_encoded_dim = 768
        self.precision = precision
        self.tokenizer = tokenizer
        state = None
        if modelpath is not None and os.path.isdir(modelpath):
            # host-side asset loading through the same third-party entry points the reference uses (mld_clip.py:28-29)
            from transformers import AutoModel, AutoTokenizer
            if self.tokenizer is None:
                self.tokenizer = AutoTokenizer.from_pretrained(modelpath)
            hf = AutoModel.from_pretrained(modelpath)
            cfg = hf.config.text_config
            if (cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size, cfg.hidden_act) != (768, 12, 3072, "quick_gelu"):
                raise NotImplementedError("only the CLIP ViT-L/14 text geometry (768 wide, 12 heads, quick_gelu) is built")
            vocab_size, num_layers = cfg.vocab_size, cfg.num_hidden_layers
            state = {k: v for k, v in hf.state_dict().items() if k in schema.clip_text_schema(vocab_size, num_layers)}
        self.max_length = getattr(self.tokenizer, "model_max_length", 77) if self.tokenizer is not None else 77
        self.text_model = _ClipTextTower(vocab_size, num_layers)
        if state is not None:
            self.text_model.load_state_dict(state, strict=True)
        for p in self.text_model.parameters():
            p.requires_grad = False

    @property
    def device(self):
        return next(self.text_model.parameters()).device

    def forward(self, texts):
        """List[str] -> [len(texts), 1, 768]   (mld_clip.py:51-86)."""
        if self.tokenizer is None:
            raise _lib.LadiffHipError("MldTextEncoder has no tokenizer: pass tokenizer= (the CLIP BPE tokenizer) or a "
                                      "modelpath holding its vocabulary; token ids can be fed to encode_ids directly")
        enc = self.tokenizer(texts, padding="max_length", truncation=True, max_length=self.max_length, return_tensors="pt")
        ids = enc.input_ids if hasattr(enc, "input_ids") else enc["input_ids"]
        return self.encode_ids(ids[:, :self.max_length]).unsqueeze(1)

    @torch.no_grad()
    def encode_ids(self, input_ids, full_length=False, dedup=True, ragged=True):
        """Token ids [B, S<=77] (int64; as the tokenizer delivers them: on the HOST - a device tensor works but costs a device -> host
        copy, i.e. a stream synchronisation) -> text features [B, 768] on the module's device.  Host work (as the reference's tokenizer
        side): identical prompts are evaluated once (`dedup`: the B empty prompts of the guidance batch), every prompt's EOS position
        is found (the argmax the reference's CLIPTextTransformer takes) and the rows are laid out ragged - prompt b at its own positions
        0 .. eos_b (`ragged`; False: every prompt padded to the batch's longest, `full_length`: to all S positions)."""
        tower = self.text_model
        dev = self.device
        ids = torch.as_tensor(input_ids).to(torch.int64)
        if ids.device.type != "cpu":
            ids = ids.cpu()
        if ids.dim() != 2 or ids.shape[1] < 1 or ids.shape[1] > 77:
            raise ValueError(f"input_ids must be [B, 1..77], got {tuple(ids.shape)}")
        # the host side works on numpy arrays: torch's CPU ops wake the whole intra-op thread pool for every one of these tiny tensors
        # (measured on the 128-thread GPU box: 4 ms for unique(dim=0) of [256,77], 5 ms for the row layout below; numpy: < 0.2 ms in all)
        a = np.ascontiguousarray(ids.numpy())
        if a.size and (int(a.min()) < 0 or int(a.max()) >= tower.vocab_size):
            raise IndexError("index out of range in self")          # what nn.Embedding raises inside the reference
        B, S = a.shape
        out = torch.empty(B, 768, dtype=torch.float32, device=dev)
        if B == 0:
            return out
        inverse = None
        if dedup:
            uniq, inv = np.unique(a, axis=0, return_inverse=True)
            if uniq.shape[0] != B:
                a, inverse = np.ascontiguousarray(uniq), inv.reshape(-1).astype(np.int64)
        n = a.shape[0]
        eos = a.argmax(axis=1)                                          # first position of the largest id (EOS), per prompt
        Lx = S if full_length else int(eos.max()) + 1
        L = _lib.lib()
        wt = tower._weight_table()
        split = wt.split_array() if _lib.is_split(self.precision) else None
        res = out if inverse is None else torch.empty(n, 768, dtype=torch.float32, device=dev)
        if ragged and not full_length:
            seq_len = (eos + 1).astype(np.int32)
            row_off = np.zeros(n + 1, dtype=np.int32)
            np.cumsum(seq_len, out=row_off[1:])
            total = int(row_off[-1])
            row_seq = np.repeat(np.arange(n, dtype=np.int32), seq_len)
            # one pinned staging slot, one asynchronous copy: [ids (int64) | seq_len | row_off | row_seq (int32)]; nothing here waits for the GPU
            meta = np.concatenate([seq_len, row_off, row_seq, np.zeros(1, dtype=np.int32)])
            meta = meta[:meta.size // 2 * 2]
            packed = np.concatenate([a.reshape(-1), meta.view(np.int64)])
            d_packed = self._stage(packed, dev)
            d_ids = d_packed[:n * S].view(n, S)
            d_meta = d_packed[n * S:].view(torch.int32)
            d_len, d_off, d_seq = d_meta[:n], d_meta[n:2 * n + 1], d_meta[2 * n + 1:2 * n + 1 + total]
            wsb = L.ladiff_clip_workspace_bytes_ragged(n, total)
            ws = _lib.workspace(wsb, dev)
            _lib.check(L.ladiff_clip_text_encode_ragged(wt.array, split, tower.num_layers, tower.vocab_size, _lib.ptr(d_ids, torch.int64), n, S, Lx,
                                                        _lib.ptr(d_len, torch.int32), _lib.ptr(d_off, torch.int32), _lib.ptr(d_seq, torch.int32),
                                                        total, _lib.ptr(res), _lib.ptr(ws), wsb, _lib.stream_ptr()))
            self._keep = d_packed                                       # alive until the next call (the stream may still read it)
        else:
            d_ids = self._stage(a.reshape(-1), dev).view(n, S)
            wsb = L.ladiff_clip_workspace_bytes(n, Lx)
            ws = _lib.workspace(wsb, dev)
            _lib.check(L.ladiff_clip_text_encode(wt.array, split, tower.num_layers, tower.vocab_size, _lib.ptr(d_ids, torch.int64), n, S, Lx,
                                                 _lib.ptr(res), _lib.ptr(ws), wsb, _lib.stream_ptr()))
            self._keep = d_ids
        if inverse is not None:
            out = res[self._stage(inverse, dev)]
        return out

    def _stage(self, host_tensor, dev, slots=16):
        """Host int64 numpy vector -> device copy through a ring of pinned staging slots carved from ONE pinned arena (a hipHostMalloc costs tens
        of milliseconds; a pageable source makes the copy wait for the stream): asynchronous, a slot is reused `slots` calls later, behind
        its copy's event."""
        n = int(host_tensor.size)
        ring = self.__dict__.get("_ring")
        if ring is None or ring["elems"] < n:
            if ring is not None:
                torch.cuda.synchronize(dev)                             # copies from the old arena may still be queued
            elems = max(2 * n, 32768)
            ring = {"i": 0, "elems": elems, "arena": torch.empty(slots * elems, dtype=torch.int64).pin_memory(), "events": [None] * slots}
            self.__dict__["_ring"] = ring
        k = ring["i"] % slots
        ring["i"] += 1
        if ring["events"][k] is not None:
            ring["events"][k].synchronize()                             # the copy issued `slots` calls ago has long run
        src = ring["arena"][k * ring["elems"]:k * ring["elems"] + n]
        src.numpy()[:] = host_tensor                                    # (a numpy int64 vector)
        d = src.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        ring["events"][k] = ev
        return d
