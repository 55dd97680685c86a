"""Drop-ins for the T2M evaluator networks and the TM2T metrics: what `LADIFF.t2m_eval` / `test.py` use
to turn generated motions into R-precision / FID / matching-score / diversity numbers.

    ladiff.models.architectures.t2m_motionenc.MovementConvEncoder   -> ladiff_amd.evaluators.MovementConvEncoder
    ladiff.models.architectures.t2m_motionenc.MotionEncoderBiGRUCo  -> ladiff_amd.evaluators.MotionEncoderBiGRUCo
    ladiff.models.architectures.t2m_textenc.TextEncoderBiGRUCo      -> ladiff_amd.evaluators.TextEncoderBiGRUCo
    ladiff.models.metrics.tm2t.TM2TMetrics                          -> ladiff_amd.evaluators.TM2TMetrics

The three networks keep the reference's constructor arguments and state-dict keys (the `movement_encoder` /
`motion_encoder` / `text_encoder` sub-dicts of the evaluator checkpoint load with `strict=True`, `ladiff.py:205-212`) and run
in libladiff_hip.so (`csrc/evaluator.hip`).  The metrics are host-side numpy / scipy exactly as in the reference
(`metrics/tm2t.py` moves everything to the CPU first); they do not depend on torchmetrics.
"""
import numpy as np
import torch

from . import _lib, schema
from .modules import _HipModule


class _Frozen(_HipModule):
    def _freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        return self

    def _table(self):
        return self._weight_table()


class MovementConvEncoder(_Frozen):
    """`t2m_motionenc.py:6-25`: Conv1d(k4,s2,p1)+LeakyReLU x2, Linear; [B, F, input_size] -> [B, (F//2)//2, output_size]."""
    _KIND = "t2m_movement"

    def __init__(self, input_size, hidden_size, output_size):
        super().__init__()
        if (hidden_size, output_size) != (512, 512):
            raise NotImplementedError("only the shipped evaluator geometry (512, 512) is built (configs/modules/evaluators.yaml)")
        self.input_size = input_size
        self._build(schema.t2m_movement_schema(input_size, hidden_size, output_size))
        self._freeze()

    @torch.no_grad()
    def forward(self, inputs):
        L = _lib.lib()
        if inputs.dim() != 3 or inputs.shape[-1] != self.input_size:
            raise ValueError(f"expected [B, F, {self.input_size}], got {tuple(inputs.shape)}")
        x = inputs.detach().to(torch.float32)
        B, F, C = x.shape
        # `feats[..., :-4]` (ladiff.py:1264) is a view with row stride nfeats: read it in place
        if not x.is_cuda:
            raise _lib.LadiffHipError("libladiff_hip works on GPU tensors only; got a CPU tensor (no CPU fallback exists)")
        if x.stride(2) == 1 and x.stride(0) == F * x.stride(1) and x.stride(1) >= C:
            ld = x.stride(1)
        else:
            x = x.contiguous(); ld = C
        out = torch.empty(B, (F // 2) // 2, 512, dtype=torch.float32, device=x.device)
        wsb = L.ladiff_t2m_movement_workspace_bytes(B, F, C)
        ws = _lib.workspace(wsb, x.device)
        _lib.check(L.ladiff_t2m_movement_encode(self._table().array, x.data_ptr(), ld, B, F, C, _lib.ptr(out), _lib.ptr(ws), wsb,
                                                _lib.stream_ptr()))
        return out


def _lens(m_lens, B, device):
    t = torch.as_tensor(m_lens).reshape(-1).to(torch.int64)
    if t.numel() != B:
        raise ValueError("one length per sample")
    if int(t.min()) < 1:
        raise RuntimeError("Length of all samples has to be greater than 0")          # pack_padded_sequence's message
    return t.to(device=device, dtype=torch.int32).contiguous()


class MotionEncoderBiGRUCo(_Frozen):
    """`t2m_motionenc.py:28-64`: Linear, bidirectional GRU over the packed sequences, co-embedding head -> [B, output_size]."""
    _KIND = "t2m_motion"

    def __init__(self, input_size, hidden_size, output_size):
        super().__init__()
        if (input_size, hidden_size, output_size) != (512, 1024, 512):
            raise NotImplementedError("only the shipped evaluator geometry (512, 1024, 512) is built")
        self.hidden_size = hidden_size
        self._build(schema.t2m_motion_schema(input_size, hidden_size, output_size))
        self._freeze()

    @torch.no_grad()
    def forward(self, inputs, m_lens):
        L = _lib.lib()
        x = inputs.detach().to(torch.float32).contiguous()
        B, T, _ = x.shape
        lens = _lens(m_lens, B, x.device)
        if int(lens.max()) > T:
            raise ValueError("a length exceeds the sequence")
        out = torch.empty(B, 512, dtype=torch.float32, device=x.device)
        wsb = L.ladiff_t2m_motion_workspace_bytes(B, T)
        ws = _lib.workspace(wsb, x.device)
        _lib.check(L.ladiff_t2m_motion_encode(self._table().array, _lib.ptr(x), lens.data_ptr(), B, T, _lib.ptr(out), _lib.ptr(ws),
                                              wsb, _lib.stream_ptr()))
        return out


class TextEncoderBiGRUCo(_Frozen):
    """`t2m_textenc.py:6-48`: POS embedding + word embedding, Linear, bidirectional GRU, co-embedding head -> [B, output_size]."""
    _KIND = "t2m_text"

    def __init__(self, word_size, pos_size, hidden_size, output_size):
        super().__init__()
        if (word_size, pos_size, hidden_size, output_size) != (300, 15, 512, 512):
            raise NotImplementedError("only the shipped evaluator geometry (300, 15, 512, 512) is built")
        self.hidden_size = hidden_size
        self._build(schema.t2m_text_schema(word_size, pos_size, hidden_size, output_size))
        self._freeze()

    @torch.no_grad()
    def forward(self, word_embs, pos_onehot, cap_lens):
        L = _lib.lib()
        w = word_embs.detach().to(torch.float32).contiguous()
        p = pos_onehot.detach().to(device=w.device, dtype=torch.float32).contiguous()
        B, T, _ = w.shape
        lens = _lens(cap_lens, B, w.device)
        out = torch.empty(B, 512, dtype=torch.float32, device=w.device)
        wsb = L.ladiff_t2m_text_workspace_bytes(B, T)
        ws = _lib.workspace(wsb, w.device)
        _lib.check(L.ladiff_t2m_text_encode(self._table().array, _lib.ptr(w), _lib.ptr(p), lens.data_ptr(), B, T, _lib.ptr(out),
                                            _lib.ptr(ws), wsb, _lib.stream_ptr()))
        return out


class TM2TMetrics:
    """Matching score, R-precision@1..k, FID and diversity of text / generated-motion / real-motion co-embeddings
    (`metrics/tm2t.py`).  `update` caches batches, `compute` evaluates on the host.  The two random draws of the reference
    (sequence shuffle, diversity pairs) come from `numpy.random.Generator(seed)` so that a run is repeatable."""

    def __init__(self, top_k=3, R_size=32, diversity_times=300, seed=0, **kwargs):
        self.name = "matching, fid, and diversity scores"
        self.top_k, self.R_size, self.diversity_times = top_k, R_size, diversity_times
        self.rng = np.random.default_rng(seed)
        self.reset()

    def reset(self):
        self.count = 0
        self.count_seq = 0
        self._text, self._rec, self._gt = [], [], []

    @property
    def metrics(self):
        names = []
        for tag in ("", "gt_"):
            names.append(tag + "Matching_score")
            names += [f"{tag}R_precision_top_{k}" for k in range(1, self.top_k + 1)]
        return names + ["FID", "Diversity", "gt_Diversity"]

    def update(self, text_embeddings, recmotion_embeddings, gtmotion_embeddings, lengths):
        self.count += int(sum(lengths))
        self.count_seq += len(lengths)
        for store, t in ((self._text, text_embeddings), (self._rec, recmotion_embeddings), (self._gt, gtmotion_embeddings)):
            store.append(torch.flatten(t.detach(), start_dim=1).cpu().double().numpy())

    @staticmethod
    def _matching(text, motion, R, top_k):
        groups = text.shape[0] // R
        score, hits = 0.0, np.zeros(top_k)
        want = np.arange(R)
        for i in range(groups):
            a, b = text[i * R:(i + 1) * R], motion[i * R:(i + 1) * R]
            d2 = -2.0 * a @ b.T + np.square(a).sum(1, keepdims=True) + np.square(b).sum(1)
            with np.errstate(invalid="ignore"):
                d = np.nan_to_num(np.sqrt(d2))
            score += np.trace(d)
            rank = np.argsort(d, axis=1, kind="stable")
            found = np.zeros(R, dtype=bool)
            for k in range(top_k):
                found |= rank[:, k] == want
                hits[k] += found.sum()
        n = groups * R
        return score / n, hits / n

    @staticmethod
    def _fid(real, fake, eps=1e-6):
        import scipy.linalg
        mu1, s1 = real.mean(0), np.cov(real, rowvar=False)
        mu2, s2 = fake.mean(0), np.cov(fake, rowvar=False)
        covmean = scipy.linalg.sqrtm(s1.dot(s2))
        if not np.isfinite(covmean).all():
            off = np.eye(s1.shape[0]) * eps
            covmean = scipy.linalg.sqrtm((s1 + off).dot(s2 + off))
        if np.iscomplexobj(covmean):
            if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
                raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
            covmean = covmean.real
        diff = mu1 - mu2
        return float(diff.dot(diff) + np.trace(s1) + np.trace(s2) - 2 * np.trace(covmean))

    def compute(self, sanity_flag=False, order=None, div_first=None, div_second=None):
        out = {m: 0.0 for m in self.metrics}
        if sanity_flag:
            return out
        n = self.count_seq
        if n <= self.R_size or n <= self.diversity_times:
            raise AssertionError("not enough sequences for the R-precision groups / diversity pairs")
        order = self.rng.permutation(n) if order is None else np.asarray(order)
        text, rec, gt = (np.concatenate(x, axis=0)[order] for x in (self._text, self._rec, self._gt))
        for tag, mot in (("", rec), ("gt_", gt)):
            score, hits = self._matching(text, mot, self.R_size, self.top_k)
            out[tag + "Matching_score"] = float(score)
            for k in range(self.top_k):
                out[f"{tag}R_precision_top_{k + 1}"] = float(hits[k])
        out["FID"] = self._fid(gt, rec)
        i1 = self.rng.choice(n, self.diversity_times, replace=False) if div_first is None else np.asarray(div_first)
        i2 = self.rng.choice(n, self.diversity_times, replace=False) if div_second is None else np.asarray(div_second)
        out["Diversity"] = float(np.linalg.norm(rec[i1] - rec[i2], axis=1).mean())
        out["gt_Diversity"] = float(np.linalg.norm(gt[i1] - gt[i2], axis=1).mean())
        return out
