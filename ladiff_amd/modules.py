"""Drop-in `nn.Module`s for the two hot-path networks: same class names, constructor kwargs, state-dict keys and
call signatures as the reference's plugin classes, arithmetic in libladiff_hip.so.

    cfg.model.denoiser.target   : ladiff_amd.modules.LADiffDenoiser   (ref: ladiff.models.architectures.ladiff_denoiser.LADiffDenoiser)
    cfg.model.motion_vae.target : ladiff_amd.modules.LADiffVae        (ref: ladiff.models.architectures.ladiff_vae.LADiffVae)

The modules hold the parameters (so `load_state_dict(strict=True)` of a reference checkpoint works, demo.py:159)
but contain NO PyTorch math: `forward` / `decode` hand device pointers to the HIP library and raise if the
library or a GPU is missing.  Only the configuration the reference ships is built (text condition, trans_enc,
SKIP_CONNECT, MD_TRANS, mld PE, encoder_decoder VAE); other branches raise at construction.
"""
import math

import torch
from torch import nn

from . import _lib, schema
from .schedulers import timestep_sinusoid


class _ParamTree(nn.Module):
    """Registers parameters under nested dotted names so state-dict keys equal the reference's."""

    def _add_param(self, dotted, shape):
        head, _, rest = dotted.partition(".")
        if not rest:
            self.register_parameter(head, nn.Parameter(torch.zeros(*shape)))
            return
        if head not in self._modules:
            self.add_module(head, _ParamTree())
        self._modules[head]._add_param(rest, shape)

    def _build(self, spec):
        for name, shape in spec.items():
            self._add_param(name, shape)


def _xavier_(t):
    nn.init.xavier_uniform_(t)


def _linear_default_(w, b):
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1.0 / math.sqrt(w.shape[1])
    nn.init.uniform_(b, -bound, bound)


def _get(ablation, name, default=None):
    if isinstance(ablation, dict):
        return ablation.get(name, default)
    return getattr(ablation, name, default)


class _HipModule(_ParamTree):
    _KIND = None

    # S-format table only: tensors padded to a multiple of rows.  None since ABI 4: the decoder pads final_layer to whole 128-column GEMM
    # tiles inside the library (include/ladiff_hip.h, ladiff_vae_decode), so the table carries exactly the module's rows
    _PAD_ROWS = {}

    def _weight_table(self, kind=None):
        kind = kind or self._KIND
        sd = dict(self.named_parameters())
        names = _lib.param_names(kind)
        key = _lib.WeightTable.key_of(names, sd)
        cache = self.__dict__.setdefault("_wt", {})
        tab = cache.get(kind)
        if tab is None or tab.key != key:
            tab = _lib.WeightTable(kind, sd, pad_rows=self._PAD_ROWS.get(kind))
            cache[kind] = tab
        return tab


class LADiffDenoiser(_HipModule):
    """Length-aware latent denoiser; replaces `ladiff_denoiser.py:16-295` behind the same interface."""
    _KIND = "denoiser"

    def __init__(self, ablation, nfeats: int = 263, condition: str = "text", latent_dim: list = [1, 256],
                 ff_size: int = 1024, num_layers: int = 6, num_heads: int = 4, dropout: float = 0.1,
                 normalize_before: bool = False, activation: str = "gelu", flip_sin_to_cos: bool = True,
                 return_intermediate_dec: bool = False, position_embedding: str = "learned", arch: str = "trans_enc",
                 freq_shift: int = 0, guidance_scale: float = 7.5, guidance_uncondp: float = 0.1,
                 text_encoded_dim: int = 768, nclasses: int = 10, **kwargs) -> None:
        super().__init__()
        self.latent_dim = latent_dim[-1]
        self.text_encoded_dim = text_encoded_dim
        self.condition = condition
        self.arch = arch
        self.test_efficiency = bool(_get(ablation, "TEST_EFFICIENCY", False))
        # errors mirror ladiff_denoiser.py:84,97,151 where the reference has them; the rest are "not built"
        if condition not in ("text", "text_uncond"):
            raise TypeError(f"condition type {condition} not supported")
        if _get(ablation, "DIFF_PE_TYPE", "mld") != "mld":
            raise ValueError("Not Support PE type")
        if arch != "trans_enc":
            raise ValueError(f"Not supported architechure{arch}!")
        unsupported = []
        if not _get(ablation, "SKIP_CONNECT", True): unsupported.append("SKIP_CONNECT=False")
        if not _get(ablation, "MD_TRANS", True): unsupported.append("MD_TRANS=False")
        if _get(ablation, "VAE_TYPE", "actor") == "no": unsupported.append("VAE_TYPE='no'")
        if position_embedding not in ("learned", "v3"): unsupported.append(f"position_embedding={position_embedding!r}")
        if (self.latent_dim, ff_size, num_layers, num_heads, text_encoded_dim) != (256, 1024, 9, 4, 768):
            unsupported.append("sizes other than latent 256 / ff 1024 / 9 layers / 4 heads / text 768")
        if normalize_before or not flip_sin_to_cos or freq_shift != 0:
            unsupported.append("normalize_before / flip_sin_to_cos=False / freq_shift!=0")
        if unsupported:
            raise NotImplementedError("LADiffDenoiser (gfx950) builds the shipped config only; not built: "
                                      + ", ".join(unsupported))
        self._build(schema.denoiser_schema(self.latent_dim, ff_size, num_layers, text_encoded_dim))
        self.reset_parameters()
        self.precision = kwargs.get("precision", "fp32")     # "fp32" | "f16x3" (matrix products only)

    def reset_parameters(self):
        """Init as the reference's constructors leave it."""
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith(".pe"):
                    p.uniform_(0.0, 1.0)                      # position_encoding.py:150-151
                elif name.startswith("encoder."):
                    if p.dim() > 1:
                        _xavier_(p)                           # cross_attention.py:37-40 (overrides zero_module)
                    elif ".norm" in name and name.endswith("weight"):
                        p.fill_(1.0)
                    elif name.endswith("in_proj_bias") or "out_proj" in name or "out_layers" in name \
                            or name.endswith("ffn.linear2.bias") or ".norm" in name:
                        p.zero_()
                    else:
                        p.uniform_(-1.0 / 16.0, 1.0 / 16.0)   # nn.Linear default bias, fan_in 256
            for lin in ("time_embedding.linear_1", "time_embedding.linear_2", "emb_proj.1"):
                _linear_default_(self.get_parameter(lin + ".weight"), self.get_parameter(lin + ".bias"))

    # ------------------------------------------------------------------
    def forward(self, sample, timestep, encoder_hidden_states, enclat=None, enclat_future=None, lengths=None,
                latent_idx=None, max_iter_elements=None, **kwargs):
        """sample [B2,T,256], timestep 0-dim or [B2] (all equal), encoder_hidden_states [B2,1,768] -> (eps [B2,T,256],)"""
        if enclat is not None or enclat_future is not None:
            raise NotImplementedError("autoregressive conditioning (ARDIFF) is not built")
        n_text = int(encoder_hidden_states.shape[1])
        L = _lib.lib()
        dev = sample.device
        B2, T, Dm = sample.shape
        if Dm != self.latent_dim or T > _lib.MAX_LATENTS:
            raise ValueError(f"unsupported sample shape {tuple(sample.shape)}")
        t = torch.as_tensor(timestep).reshape(-1)
        if t.numel() > 1 and not bool((t == t[0]).all()):
            raise NotImplementedError("per-sample timesteps are not built (the sampling loop uses one t per call)")
        wt = self._weight_table()
        x = sample.detach().to(torch.float32).contiguous()
        text = encoder_hidden_states.detach().to(torch.float32).contiguous()
        counts = None
        if max_iter_elements is not None and not self.test_efficiency:
            counts = torch.as_tensor(max_iter_elements).to(device=dev, dtype=torch.int32).contiguous()
        sinus = timestep_sinusoid(t[:1].cpu(), self.text_encoded_dim).to(dev)
        tables = torch.empty(L.ladiff_denoiser_tables_floats(1), dtype=torch.float32, device=dev)
        cache = torch.empty(L.ladiff_denoiser_text_cache_floats(B2, 1, n_text), dtype=torch.float32, device=dev)
        wsb = L.ladiff_denoiser_workspace_bytes(B2, T, 1, n_text)
        ws = _lib.workspace(wsb, dev)
        step0 = torch.zeros(1, dtype=torch.int32, device=dev)
        eps = torch.empty_like(x)
        st = _lib.stream_ptr()
        _lib.check(L.ladiff_denoiser_time_tables(wt.array, _lib.ptr(sinus), 1, _lib.ptr(tables), _lib.ptr(ws), wsb, st))
        _lib.check(L.ladiff_denoiser_text_cache(wt.array, _lib.ptr(text), B2, n_text, _lib.ptr(tables), 1, _lib.ptr(cache),
                                                _lib.ptr(ws), wsb, st))
        _lib.check(L.ladiff_denoiser_forward(wt.array, wt.split_array() if _lib.is_split(self.precision) else None,
                                             _lib.ptr(tables), step0.data_ptr(), _lib.ptr(cache), n_text, 1, _lib.ptr(x),
                                             B2, 1, T, None if counts is None else counts.data_ptr(), _lib.ptr(eps),
                                             _lib.ptr(ws), wsb, st))
        return (eps.to(sample.dtype),)


class LADiffVae(_HipModule):
    """Length-aware VAE; `decode` replaces `ladiff_vae.py:288-362`, `encode` replaces `:162-286`."""
    _KIND = "decoder"

    def __init__(self, ablation, nfeats: int, latent_dim: list = [1, 256], ff_size: int = 1024, num_layers: int = 9,
                 num_heads: int = 4, dropout: float = 0.1, arch: str = "all_encoder", normalize_before: bool = False,
                 activation: str = "gelu", position_embedding: str = "learned", **kwargs) -> None:
        super().__init__()
        self.latent_size = latent_dim[0]
        self.latent_dim = latent_dim[-1]
        self.nfeats = nfeats
        self.arch = arch
        self.max_it = int(_get(ablation, "MAX_IT", 5))
        self.frame_per_latent = int(_get(ablation, "FRAME_PER_LATENT", 48))
        self.test_efficiency = bool(_get(ablation, "TEST_EFFICIENCY", False))
        self.length_aware = True        # decode only the valid frames of a mixed-length batch (same results, fewer rows)
        # Opt-in: decodes of fewer frame rows than this are replayed from a hipGraph over persistent buffers (0 = never, the default).
        # Measured on config c1 (8 x 60 frames, profiles/r3): 0.467 ms replayed against 0.458 ms launched directly - with the small-M
        # GEMM routing the ~110 launches are paced by the GPU (kernel + dependent-boundary time), not by the host, so the graph only
        # frees host time; and a batch whose lengths change from call to call would re-capture every time (milliseconds).
        self.graph_rows = 0
        self._dec_plans = {}
        self._graph_stream = None
        if _get(ablation, "PE_TYPE", "mld") != "mld":
            raise ValueError("Not Support PE type")
        if arch not in ("all_encoder", "encoder_decoder"):
            raise ValueError("Not support architecture!")
        unsupported = []
        if arch != "encoder_decoder": unsupported.append("arch='all_encoder'")
        if _get(ablation, "MLP_DIST", False): unsupported.append("MLP_DIST")
        if self.max_it == 0 or self.max_it > _lib.MAX_LATENTS: unsupported.append(f"MAX_IT={self.max_it}")
        if activation != "gelu" or normalize_before: unsupported.append("activation!='gelu' / normalize_before")
        if position_embedding not in ("learned", "v3"): unsupported.append(f"position_embedding={position_embedding!r}")
        if (self.latent_dim, ff_size, num_layers, num_heads) != (256, 1024, 9, 4):
            unsupported.append("sizes other than latent 256 / ff 1024 / 9 layers / 4 heads")
        if unsupported:
            raise NotImplementedError("LADiffVae (gfx950) builds the shipped config only; not built: "
                                      + ", ".join(unsupported))
        self._build(schema.vae_schema(nfeats, self.latent_dim, ff_size, num_layers, self.max_it))
        self.reset_parameters()
        self.precision = kwargs.get("precision", "fp32")     # "fp32" | "f16x3"

    def reset_parameters(self):
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith(".pe"):
                    p.uniform_(0.0, 1.0)
                elif name == "global_motion_token":
                    p.normal_()                                # ladiff_vae.py:119-120
                elif name.startswith(("encoder.", "decoder.")):
                    if p.dim() > 1:
                        _xavier_(p)
                    elif ".norm" in name and name.endswith("weight"):
                        p.fill_(1.0)
                    elif name.endswith("in_proj_bias") or "out_proj" in name or ".norm" in name:
                        p.zero_()
                    else:
                        fan_in = 1024 if name.endswith("linear2.bias") else (512 if "linear_blocks" in name else 256)
                        p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            for lin in ("skel_embedding", "final_layer"):
                _linear_default_(self.get_parameter(lin + ".weight"), self.get_parameter(lin + ".bias"))

    def encode(self, features, lengths=None, eps=None):
        """features [B,F,nfeats], lengths list[int] -> (latent [max_it,B,256], Normal(mu, std), max_iter_elements).

        `LADiffVae.encode` of the reference (ladiff_vae.py:162-286; LAD branch).  `eps` ([max_it,B,256], optional)
        replaces the standard-normal draw of `dist.rsample()` so that results can be reproduced."""
        L = _lib.lib()
        dev = features.device
        B, F, C = features.shape
        if lengths is None:
            lengths = [F] * B
        lengths = [int(l) for l in lengths]
        T = self.max_it
        if len(lengths) != B or C != self.nfeats:
            raise ValueError(f"features {tuple(features.shape)} do not match {len(lengths)} lengths / nfeats {self.nfeats}")
        if F + 2 * T > _lib.MAX_FRAMES:
            raise NotImplementedError(f"encode handles up to {_lib.MAX_FRAMES - 2 * T} frames")
        counts = [int(math.ceil(l / self.frame_per_latent)) for l in lengths]
        lens_t = _lib.device_ints(lengths, dev)
        counts_t = _lib.device_ints(counts, dev)
        if eps is None:
            eps = torch.randn(T, B, self.latent_dim, dtype=torch.float32, device=dev)     # Normal.rsample's draw
        eps = eps.detach().to(device=dev, dtype=torch.float32).contiguous()
        x = features.detach().to(torch.float32).contiguous()
        wt = self._weight_table("encoder")
        mu, std, latent = (torch.empty(T, B, self.latent_dim, dtype=torch.float32, device=dev) for _ in range(3))
        wsb = L.ladiff_encoder_workspace_bytes(B, F, T, C)
        ws = _lib.workspace(wsb, dev)
        _lib.check(L.ladiff_vae_encode(wt.array, wt.split_array() if _lib.is_split(self.precision) else None, _lib.ptr(x),
                                       lens_t.data_ptr(), counts_t.data_ptr(), _lib.ptr(eps), B, F, T, C, _lib.ptr(mu),
                                       _lib.ptr(std), _lib.ptr(latent), _lib.ptr(ws), wsb, _lib.stream_ptr()))
        dist = torch.distributions.Normal(mu, std)
        return latent.to(features.dtype), dist, torch.tensor(counts, dtype=torch.long)

    def __del__(self):
        try:
            for plan in self._dec_plans.values():
                _lib.lib().ladiff_decoder_graph_destroy(plan["graph"])
        except Exception:
            pass

    def _decode_graphed(self, zz, lengths, counts, counts_t, lens_t, ragged_rows, wt, wsplit):
        """Few frame rows: the decode is ~110 launches of a few microseconds each, paced by the host.  It is captured once per
        (shape, lengths, weights) into a hipGraph over persistent buffers and replayed (ladiff_vae_decode_graphed)."""
        from ctypes import byref, c_void_p
        L = _lib.lib()
        dev = zz.device
        T, B, _ = zz.shape
        F = max(lengths)
        cur = torch.cuda.current_stream(dev)
        if self._graph_stream is None or self._graph_stream.device != dev:
            self._graph_stream = torch.cuda.Stream(device=dev)
        run = cur if cur.cuda_stream != 0 else self._graph_stream           # capture is illegal on the null stream
        key = (B, F, T, tuple(lengths), tuple(counts), ragged_rows, str(dev), self.precision, bool(self.test_efficiency), run.cuda_stream)
        plan = self._dec_plans.get(key)
        if plan is None:
            while len(self._dec_plans) >= 4:
                old = self._dec_plans.pop(next(iter(self._dec_plans)))
                _lib.check(L.ladiff_decoder_graph_destroy(old["graph"]))
            h = c_void_p()
            _lib.check(L.ladiff_decoder_graph_create(byref(h)))
            wsb = L.ladiff_decoder_workspace_bytes(B, F, T, self.nfeats)
            off_t = None
            if ragged_rows:
                off = [0] * (B + 1)
                for i, l in enumerate(lengths):
                    off[i + 1] = off[i] + l
                off_t = torch.tensor(off, dtype=torch.int32, device=dev)
            plan = {"graph": h, "z": torch.empty_like(zz), "feats": torch.empty(B, F, self.nfeats, dtype=torch.float32, device=dev),
                    "ws": _lib.workspace(wsb, dev), "wsb": wsb, "off": off_t,
                    # private copies: the graph bakes these pointers in (the shared device_ints cache may evict its entries)
                    "lens": lens_t.clone(), "counts": None if counts_t is None else counts_t.clone()}
            self._dec_plans[key] = plan
        else:
            self._dec_plans[key] = self._dec_plans.pop(key)
        if run is not cur:
            run.wait_stream(cur)
        with torch.cuda.stream(run):
            plan["z"].copy_(zz)
            _lib.check(L.ladiff_vae_decode_graphed(
                plan["graph"], wt.array, wsplit, wt.generation, _lib.ptr(plan["z"]), plan["lens"].data_ptr(),
                None if plan["counts"] is None else plan["counts"].data_ptr(), None if plan["off"] is None else plan["off"].data_ptr(),
                ragged_rows, B, F, T, self.nfeats, _lib.ptr(plan["feats"]), _lib.ptr(plan["ws"]), plan["wsb"], run.cuda_stream))
            out = plan["feats"].clone()
        if run is not cur:
            cur.wait_stream(run)
        return out

    def decode(self, z, lengths, plot_att_map=None, latentwise_gen=None):
        """z [max_it,B,256], lengths list[int] -> feats [B, max(lengths), nfeats]; frames >= len are zero."""
        if plot_att_map:
            raise NotImplementedError("plot_att_map (matplotlib debug heat-maps, cross_attention.py:378-406) is not built")
        L = _lib.lib()
        dev = z.device
        T, B, Dm = z.shape
        lengths = [int(l) for l in lengths]
        if len(lengths) != B or Dm != self.latent_dim or T > _lib.MAX_LATENTS:
            raise ValueError(f"z {tuple(z.shape)} does not match {len(lengths)} lengths")
        F = max(lengths)
        if F > _lib.MAX_FRAMES:
            raise NotImplementedError(f"motions longer than {_lib.MAX_FRAMES} frames are not built (reference MAX_LEN 196)")
        if latentwise_gen == "fw":
            counts = list(range(1, self.max_it + 1))           # ladiff_vae.py:295
            if len(counts) != B:
                raise ValueError("latentwise_gen='fw' expects batch == MAX_IT")
        else:
            counts = [int(math.ceil(l / self.frame_per_latent)) for l in lengths]
        counts_t = None
        if not self.test_efficiency:
            counts_t = _lib.device_ints(counts, dev)
        lens_t = _lib.device_ints(lengths, dev)
        wt = self._weight_table()
        wsplit = wt.split_array() if _lib.is_split(self.precision) else None
        zz = z.detach().to(torch.float32).contiguous()
        rows = sum(lengths)
        ragged = self.length_aware and rows < B * F
        if self.graph_rows and (rows if ragged else B * F) < self.graph_rows:
            return self._decode_graphed(zz, lengths, counts, counts_t, lens_t, rows if ragged else 0, wt, wsplit).to(z.dtype)
        feats = torch.empty(B, F, self.nfeats, dtype=torch.float32, device=dev)
        wsb = L.ladiff_decoder_workspace_bytes(B, F, T, self.nfeats)
        ws = _lib.workspace(wsb, dev)
        if ragged:
            # mixed lengths: only the valid frames are computed (ragged rows, ladiff_vae_decode_ragged); same frames out
            off = [0] * (B + 1)
            for i, l in enumerate(lengths):
                off[i + 1] = off[i] + l
            _lib.check(L.ladiff_vae_decode_ragged(wt.array, wsplit, _lib.ptr(zz), lens_t.data_ptr(),
                                                  None if counts_t is None else counts_t.data_ptr(),
                                                  _lib.device_ints(off, dev).data_ptr(), rows, B, F, T, self.nfeats,
                                                  _lib.ptr(feats), _lib.ptr(ws), wsb, _lib.stream_ptr()))
        else:
            _lib.check(L.ladiff_vae_decode(wt.array, wsplit, _lib.ptr(zz), lens_t.data_ptr(),
                                           None if counts_t is None else counts_t.data_ptr(), B, F, T, self.nfeats,
                                           _lib.ptr(feats), _lib.ptr(ws), wsb, _lib.stream_ptr()))
        return feats.to(z.dtype)
