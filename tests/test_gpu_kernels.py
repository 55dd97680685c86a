"""GPU parity of the individual HIP kernels, called through the C ABI, against the CPU oracle."""
import math

import pytest
import torch
import torch.nn.functional as F

from ladiff_amd import _lib
from oracle import ladiff_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def lib():
    return _lib.lib()


def sync():
    torch.cuda.synchronize()


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (scale * torch.randn(*shape, generator=g)).float()


def gemm(A, W, bias=None, A2=None, res=None, ln=None, act="none"):
    M, K1 = A.shape
    N, K = W.shape
    Y = torch.full((M, N), float("nan"), device=DEV)
    d = lambda t: None if t is None else t.to(DEV).contiguous()
    A_, W_, b_, A2_, r_ = d(A), d(W), d(bias), d(A2), d(res)
    g_, be_ = (d(ln[0]), d(ln[1])) if ln else (None, None)
    rc = lib().ladiff_gemm(_lib.ptr(A_), A_.shape[1], _lib.ptr(A2_), 0 if A2 is None else A2_.shape[1], K1, _lib.ptr(W_), K,
                           _lib.ptr(b_), _lib.ptr(r_), N, _lib.ptr(g_), _lib.ptr(be_), _lib.ptr(Y), N, M, N, K,
                           _lib.ACT[act], _lib.stream_ptr())
    _lib.check(rc)
    sync()
    return Y.cpu()


def ref_gemm(A, W, bias=None, A2=None, res=None, ln=None, act="none"):
    X = A if A2 is None else torch.cat([A, A2], dim=1)
    y = F.linear(X.double(), W.double(), None if bias is None else bias.double())
    y = {"none": lambda v: v, "relu": F.relu, "gelu": F.gelu, "silu": F.silu}[act](y)
    if res is not None:
        y = y + res.double()
    if ln is not None:
        y = F.layer_norm(y, (y.shape[-1],), ln[0].double(), ln[1].double(), 1e-5)
    return y


# (M, N, K) chosen to hit every tile configuration of launch_gemm and ragged edges
@pytest.mark.parametrize("M,N,K,act", [
    (1280, 768, 256, "none"), (1280, 1024, 256, "relu"), (1280, 256, 1024, "none"), (256, 256, 256, "silu"),
    (50, 512, 256, "none"), (7, 256, 768, "gelu"), (4099, 768, 256, "none"), (4100, 1024, 256, "gelu"),
    (5000, 263, 256, "none"), (4097, 251, 256, "none"), (33, 263, 256, "none"), (1, 32, 32, "none"),
])
def test_gemm_bias_act(M, N, K, act):
    A, W, b = rnd(M, K), rnd(N, K, scale=1 / math.sqrt(K)), rnd(N)
    got, want = gemm(A, W, b, act=act), ref_gemm(A, W, b, act=act)
    assert torch.isfinite(got).all()
    assert (got.double() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M,K", [(1280, 256), (1280, 1024), (37, 256), (4100, 256), (5003, 1024), (16, 256)])
def test_gemm_residual_layernorm(M, K):
    A, W, b, res = rnd(M, K), rnd(256, K, scale=1 / math.sqrt(K)), rnd(256), rnd(M, 256, scale=3.0)
    ln = (1 + 0.1 * rnd(256, seed=3), 0.1 * rnd(256, seed=4))
    got, want = gemm(A, W, b, res=res, ln=ln), ref_gemm(A, W, b, res=res, ln=ln)
    assert (got.double() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("M", [1280, 4200, 19])
def test_gemm_concat_k(M):
    A, A2, W, b = rnd(M, 256), rnd(M, 256, seed=9), rnd(256, 512, scale=0.05), rnd(256)
    got, want = gemm(A, W, b, A2=A2), ref_gemm(A, W, b, A2=A2)
    assert (got.double() - want).abs().max().item() < 2e-5


def test_gemm_k_order_is_a_pure_permutation():
    """A = I picks out columns of W^T: catches a row/col swap or a wrong k permutation exactly (asymmetric W)."""
    K = 256
    A = torch.eye(K)
    W = torch.arange(96 * K, dtype=torch.float32).reshape(96, K) % 251
    assert torch.equal(gemm(A, W), W.t().contiguous())


def test_gemm_shape_errors():
    A = torch.zeros(4, 40, device=DEV); W = torch.zeros(8, 40, device=DEV); Y = torch.zeros(4, 8, device=DEV)
    rc = lib().ladiff_gemm(_lib.ptr(A), 40, None, 0, 40, _lib.ptr(W), 40, None, None, 0, None, None, _lib.ptr(Y), 8,
                           4, 8, 40, 0, _lib.stream_ptr())
    assert rc == -2    # K must be a multiple of 32


@pytest.mark.parametrize("M", [1, 5, 1280, 25088])
def test_layernorm(M):
    x, g, b = rnd(M, 256, scale=4.0), 1 + 0.1 * rnd(256), 0.1 * rnd(256, seed=2)
    y = torch.empty(M, 256, device=DEV)
    xd, gd, bd = x.to(DEV), g.to(DEV), b.to(DEV)
    _lib.check(lib().ladiff_layernorm(_lib.ptr(xd), _lib.ptr(gd), _lib.ptr(bd), _lib.ptr(y), M, _lib.stream_ptr()))
    sync()
    want = F.layer_norm(x.double(), (256,), g.double(), b.double(), 1e-5)
    assert (y.cpu().double() - want).abs().max().item() < 1e-5


def test_timestep_sinusoid_kernel():
    t = torch.tensor([981, 961, 481, 21, 1, 0, 999], dtype=torch.int64)
    out = torch.empty(len(t), 768, device=DEV)
    td = t.to(DEV)
    _lib.check(lib().ladiff_timestep_sinusoid(td.data_ptr(), len(t), _lib.ptr(out), _lib.stream_ptr()))
    sync()
    # the angle t*f carries 1 ulp of f (x t <= 999): 1e-4 is the honest bound for any fp32 evaluation
    assert (out.cpu() - orc.timestep_sinusoid(t)).abs().max().item() < 1e-4


def ref_self_attention(qkv, lengths, B, Fr):
    q, k, v = qkv.double().view(B, Fr, 3, 4, 64).permute(2, 0, 3, 1, 4)
    s = (q * 0.125) @ k.transpose(-1, -2)
    pad = ~orc.lengths_to_mask(lengths, Fr)
    s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * Fr, 256)


@pytest.mark.parametrize("lengths", [[196, 196, 60], [60] * 4, [1, 33, 32, 31, 101], [224, 200], [5], [128, 129, 97]])
def test_decoder_self_attention(lengths):
    B, Fr = len(lengths), max(lengths)
    qkv = rnd(B * Fr, 768, scale=2.0)
    out = torch.full((B * Fr, 256), float("nan"), device=DEV)
    qd, ld = qkv.to(DEV), torch.tensor(lengths, dtype=torch.int32, device=DEV)
    _lib.check(lib().ladiff_decoder_self_attention(_lib.ptr(qd), ld.data_ptr(), None, _lib.ptr(out), B, Fr, _lib.stream_ptr()))
    sync()
    got = out.cpu()
    assert torch.isfinite(got).all()
    assert (got.double() - ref_self_attention(qkv, lengths, B, Fr)).abs().max().item() < 2e-5


def test_decoder_self_attention_softmax_extremes():
    """One key dominates by a large margin (exp underflow of the rest) and all-equal scores."""
    B, Fr = 2, 196
    qkv = torch.zeros(B * Fr, 768)
    qkv[:, 512:] = rnd(B * Fr, 256)
    qkv[:Fr, :256] = 8.0          # sample 0: big q ...
    qkv[77, 256:512] = 8.0        # ... against one big key -> softmax is one-hot on key 77
    out = torch.empty(B * Fr, 256, device=DEV)
    qd, ld = qkv.to(DEV), torch.tensor([196, 150], dtype=torch.int32, device=DEV)
    _lib.check(lib().ladiff_decoder_self_attention(_lib.ptr(qd), ld.data_ptr(), None, _lib.ptr(out), B, Fr, _lib.stream_ptr()))
    sync()
    assert (out.cpu().double() - ref_self_attention(qkv, [196, 150], B, Fr)).abs().max().item() < 2e-5


def key_bitmap(valid):
    """bool [B,F] -> int32 tensor [B,8] holding the 256-bit key map (uint32 words, LSB first)."""
    import numpy as np
    B, Fr = valid.shape
    bits = np.zeros((B, 8), dtype=np.uint32)
    for b in range(B):
        for k in range(Fr):
            if valid[b, k]:
                bits[b, k // 32] |= np.uint32(1 << (k % 32))
    return torch.from_numpy(bits.view(np.int32).copy())


def test_self_attention_arbitrary_key_mask():
    """Bitmap key masks (LA-VAE encoder: masked latent tokens in the middle of the sequence)."""
    B, Fr = 3, 206
    g = torch.Generator().manual_seed(4)
    valid = torch.rand(B, Fr, generator=g) < 0.6
    valid[:, 0] = True
    valid[2, 150:] = False
    qkv = rnd(B * Fr, 768, scale=2.0)
    bd = key_bitmap(valid).to(DEV)
    out = torch.full((B * Fr, 256), float("nan"), device=DEV)
    qd = qkv.to(DEV)
    _lib.check(lib().ladiff_decoder_self_attention(_lib.ptr(qd), None, bd.data_ptr(), _lib.ptr(out), B, Fr, _lib.stream_ptr()))
    sync()
    q, k, v = qkv.double().view(B, Fr, 3, 4, 64).permute(2, 0, 3, 1, 4)
    s = ((q * 0.125) @ k.transpose(-1, -2)).masked_fill(~valid[:, None, None, :], float("-inf"))
    want = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * Fr, 256)
    assert (out.cpu().double() - want).abs().max().item() < 2e-5


def test_decoder_self_attention_rejects_long_sequences():
    x = torch.zeros(300, 768, device=DEV); o = torch.zeros(300, 256, device=DEV)
    l = torch.tensor([300], dtype=torch.int32, device=DEV)
    assert lib().ladiff_decoder_self_attention(_lib.ptr(x), l.data_ptr(), None, _lib.ptr(o), 1, 300, _lib.stream_ptr()) == -2


@pytest.mark.parametrize("T,counts", [(5, [5, 2, 3, 1]), (5, [5] * 3), (3, [1, 3]), (8, [8, 4, 1])])
def test_decoder_cross_attention(T, counts):
    B, Fr = len(counts), 50
    q, kv = rnd(B * Fr, 256, scale=2.0), rnd(T * B, 512, scale=2.0)
    out = torch.empty(B * Fr, 256, device=DEV)
    qd, kd, cd = q.to(DEV), kv.to(DEV), torch.tensor(counts, dtype=torch.int32, device=DEV)
    _lib.check(lib().ladiff_decoder_cross_attention(_lib.ptr(qd), _lib.ptr(kd), cd.data_ptr(), _lib.ptr(out), B, Fr, T,
                                                    _lib.stream_ptr()))
    sync()
    qq = q.double().view(B, Fr, 4, 64).transpose(1, 2) * 0.125
    kk = kv.double().view(T, B, 2, 4, 64)
    k, v = kk[:, :, 0].permute(1, 2, 0, 3), kk[:, :, 1].permute(1, 2, 0, 3)
    s = (qq @ k.transpose(-1, -2)).masked_fill(~orc.count_mask(counts, T)[:, None, None, :], float("-inf"))
    want = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * Fr, 256)
    assert (out.cpu().double() - want).abs().max().item() < 2e-5


# ---------------------------------------------------------------- small-M (denoiser) kernels
def to_split(t):
    """fp32 [R,K] -> S-format [R,K] on the GPU (ladiff_split_rows)."""
    x = t.to(DEV).contiguous()
    y = torch.empty_like(x)
    _lib.check(lib().ladiff_split_rows(_lib.ptr(x), _lib.ptr(y), x.shape[0], x.shape[1], _lib.stream_ptr()))
    return y


def from_split(y):
    """S-format [R,K] (GPU) -> fp32 hi + lo on the CPU."""
    half = torch.float16 if lib().ladiff_split_format() == 1 else torch.bfloat16
    b = y.cpu().contiguous().view(half).view(y.shape[0], y.shape[1] // 64, 2, 64).float()
    return (b[:, :, 0] + b[:, :, 1]).reshape(y.shape[0], y.shape[1])


def gemm_resident(A, W, bias=None, A2=None, res=None, act="none", split=False, want_split_out=False):
    M, K1 = A.shape
    N, K = W.shape
    splits = K // 256
    d = lambda t: None if t is None else t.to(DEV).contiguous()
    A_, W_, b_, A2_, r_ = d(A), d(W), d(bias), d(A2), d(res)
    if split:
        A_, W_ = to_split(A_), to_split(W_)
        A2_ = None if A2_ is None else to_split(A2_)
    Y = torch.full((splits, M, N) if splits > 1 else (M, N), float("nan"), device=DEV)
    Ys = torch.zeros(M, N, device=DEV) if want_split_out else None
    _lib.check(lib().ladiff_gemm_resident(_lib.ptr(A_), A_.shape[1], _lib.ptr(A2_), 0 if A2 is None else A2_.shape[1], K1,
                                          _lib.ptr(W_), K, _lib.ptr(b_), _lib.ptr(r_), N, _lib.ptr(Y), N, M, N, K,
                                          _lib.ACT[act], 1 if split else 0, _lib.ptr(Ys), _lib.stream_ptr()))
    sync()
    if want_split_out:
        return Y.cpu(), from_split(Ys)
    return Y.cpu()


def test_split_format_round_trip():
    x = rnd(37, 256, scale=50.0)
    back = from_split(to_split(x))
    bits = 21 if lib().ladiff_split_format() == 1 else 16       # fp16 pairs (both halves rounded toward zero): 22 bits; bf16 pairs: 16
    assert (back - x).abs().max().item() <= 2.0 ** -bits * x.abs().max().item()


def test_split_operands_saturate_instead_of_overflowing():
    """The fp16 halves of a split operand (csrc/common.h: v_cvt_pkrtz_f16_f32 rounds toward zero and never produces an infinity): an
    operand is exact to 22 bits up to 65504, degrades gracefully up to 131008 (both halves saturated) and is CLIPPED beyond - never an
    infinity or a NaN.  The bf16 flavour of the library keeps fp32's range: there the round trip is exact to 16 bits everywhere."""
    x = torch.tensor([[1.0, -3.5, 60000.0, 65504.0, -100000.0, 131008.0, 1.0e6, -3.0e38] + [0.0] * 56] * 2)
    back = from_split(to_split(x))
    assert torch.isfinite(back).all()
    if lib().ladiff_split_format() == 1:
        assert torch.equal(back[0, :4], x[0, :4]) and back[0, 4].item() == -100000.0 and back[0, 5].item() == 131008.0
        assert back[0, 6].item() == 131008.0 and back[0, 7].item() == -131008.0          # clipped at hi + lo = 2 x 65504
    else:
        assert ((back - x).abs() <= 2.0 ** -16 * x.abs()).all()
    # a product whose operands sit just inside fp16's range: finite, and as accurate as any other
    A, W = rnd(64, 256, scale=15000.0), rnd(256, 256, scale=1 / 16)
    got = gemm_resident(A, W, split=True)
    want = ref_gemm(A, W)
    assert torch.isfinite(got).all()
    bits = 20 if lib().ladiff_split_format() == 1 else 14
    assert (got.double() - want).abs().max().item() < 2.0 ** -bits * (A.abs().double() @ W.abs().double().t()).max().item()


@pytest.mark.parametrize("M,N,K,act", [(1280, 1024, 256, "gelu"), (1280, 768, 256, "none"), (1280, 256, 256, "none"),
                                       (77, 1024, 256, "relu"), (1280, 256, 1024, "none"), (90, 256, 512, "none")])
def test_gemm_resident_split(M, N, K, act):
    """3-term split products: ~2^-21 relative error per product with fp16 pairs, ~2^-16 with bf16 pairs (2^-24: the fp32 MFMA path); the bound below holds for both."""
    A, W, b = rnd(M, K, scale=3.0), rnd(N, K, scale=1 / math.sqrt(K)), rnd(N)
    if K == 256:
        res = rnd(M, N, seed=7)
        got, got_s = gemm_resident(A, W, b, res=res, act=act, split=True, want_split_out=True)
        want = ref_gemm(A, W, b, res=res, act=act)
        assert (got_s.double() - got.double()).abs().max().item() <= 2.0 ** -15 * got.abs().max().item()
    else:
        got = gemm_resident(A, W, split=True).double().sum(0)
        want = ref_gemm(A, W)
    bound = 4 * 2.0 ** -16 * (A.abs().double() @ W.abs().double().t()).max().item()
    assert (got.double() - want).abs().max().item() < bound


@pytest.mark.parametrize("M,N,act", [(1280, 768, "none"), (1280, 1024, "relu"), (1280, 256, "none"), (1280, 1024, "gelu"),
                                     (35, 768, "none"), (81, 1024, "silu"), (33, 256, "none"), (640, 1024, "none")])
def test_gemm_resident_k256(M, N, act):
    A, W, b, res = rnd(M, 256), rnd(N, 256, scale=1 / 16), rnd(N), rnd(M, N, seed=7)
    got = gemm_resident(A, W, b, res=res, act=act)
    want = ref_gemm(A, W, b, res=res, act=act)
    assert torch.isfinite(got).all()
    assert (got.double() - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M,K,concat", [(1280, 1024, False), (1280, 512, True), (45, 1024, False), (90, 512, True)])
def test_gemm_resident_split_k_and_combine(M, K, concat):
    if concat:
        A, A2 = rnd(M, 256), rnd(M, 256, seed=9)
    else:
        A, A2 = rnd(M, K), None
    W, b, res = rnd(256, K, scale=1 / math.sqrt(K)), rnd(256), rnd(M, 256, seed=5, scale=2.0)
    planes = gemm_resident(A, W, A2=A2)
    assert planes.shape == (K // 256, M, 256)
    raw = ref_gemm(A, W, A2=A2)
    assert (planes.double().sum(0) - raw).abs().max().item() < 2e-5
    ln = (1 + 0.1 * rnd(256, seed=3), 0.1 * rnd(256, seed=4))
    T, Bs = 5, 4
    nsamp = (M + T - 1) // T
    counts = torch.tensor([5, 2, 3, 1], dtype=torch.int32)
    ctab = rnd(nsamp + 1, 256, seed=11)
    mod = rnd(512, seed=12)
    pd, bd, rd = planes.to(DEV), b.to(DEV), res.to(DEV)
    gd, bed = ln[0].to(DEV), ln[1].to(DEV)
    x = raw + b.double() + res.double()
    xn = F.layer_norm(x, (256,), ln[0].double(), ln[1].double(), 1e-5)
    rows = torch.arange(M)
    valid = (rows % T) < counts[(rows // T) % Bs]
    sel = torch.where(valid, rows // T, torch.tensor(nsamp))
    wants = {0: x, 1: xn, 2: xn + ctab.double()[sel], 3: F.silu(xn * (1 + mod[:256].double()) + mod[256:].double())}
    for mode, tab in ((0, None), (1, None), (2, ctab), (3, mod)):
        out = torch.full((M, 256), float("nan"), device=DEV)
        td = None if tab is None else tab.to(DEV)
        _lib.check(lib().ladiff_combine_rows(_lib.ptr(pd), K // 256, M, _lib.ptr(bd), _lib.ptr(rd), mode, _lib.ptr(gd),
                                             _lib.ptr(bed), _lib.ptr(td), counts.to(DEV).data_ptr(), Bs, T, nsamp,
                                             _lib.ptr(out), _lib.stream_ptr()))
        sync()
        assert (out.cpu().double() - wants[mode]).abs().max().item() < 3e-5, mode


# ---------------------------------------------------------------- large-M f16x3 GEMM (decoder / encoder / CLIP)
@pytest.mark.parametrize("M,N,K,act,concat,res", [(4100, 256, 256, "none", False, True), (5000, 768, 256, "none", False, False),
                                                  (4224, 1024, 256, "gelu", False, False), (4099, 256, 1024, "none", False, True),
                                                  (4160, 256, 512, "none", True, False), (300, 768, 768, "qgelu", False, True),
                                                  (1, 128, 64, "lrelu", False, False)])
def test_gemm_split_large_m(M, N, K, act, concat, res):
    """`ladiff_gemm_split`: 128x128 tiles, 32-k stages, two workgroups per CU; ragged last row tile, concat K, fused residual,
    both output formats."""
    K1 = K // 2 if concat else K
    A, W, b = rnd(M, K, scale=2.0), rnd(N, K, scale=1 / math.sqrt(K)), rnd(N)
    R = rnd(M, N, seed=9) if res else None
    As, Ws = to_split(A[:, :K1]), to_split(W)
    A2s = to_split(A[:, K1:]) if concat else None
    Y = torch.full((M, N), float("nan"), device=DEV); Ys = torch.zeros(M, N, device=DEV)
    r_ = None if R is None else R.to(DEV)
    _lib.check(lib().ladiff_gemm_split(_lib.ptr(As), K1, _lib.ptr(A2s), 0 if A2s is None else K - K1, K1, _lib.ptr(Ws), K,
                                       _lib.ptr(b.to(DEV)), _lib.ptr(r_), N, _lib.ptr(Y), _lib.ptr(Ys), N, M, N, K,
                                       _lib.ACT[act], _lib.stream_ptr()))
    sync()
    want = A.double() @ W.double().t() + b.double()
    want = {"none": lambda v: v, "gelu": lambda v: F.gelu(v), "qgelu": lambda v: v * torch.sigmoid(1.702 * v),
            "lrelu": lambda v: F.leaky_relu(v, 0.2)}[act](want)
    if R is not None:
        want = want + R.double()
    got = Y.cpu()
    bound = 4 * 2.0 ** -16 * (A.abs().double() @ W.abs().double().t()).max().item() + 2e-6
    assert torch.isfinite(got).all() and (got.double() - want).abs().max().item() < bound
    assert (from_split(Ys).double() - got.double()).abs().max().item() <= 2.0 ** -15 * got.abs().max().item()
    assert lib().ladiff_gemm_split(_lib.ptr(As), K1, None, 0, K1, _lib.ptr(Ws), K, None, None, 0, _lib.ptr(Y), None, N, M, 100,
                                   K, 0, _lib.stream_ptr()) == -2          # N must be a multiple of 128


# ---------------------------------------------------------------- self-attention core, f16x3 arithmetic
@pytest.mark.parametrize("lengths,nheads,causal", [([196, 196, 60], 4, 0), ([1, 33, 32, 31, 101], 4, 0), ([224, 200], 4, 0),
                                                   ([5], 4, 0), ([77, 77, 77], 12, 1), ([20, 20], 12, 1), ([128, 129, 97], 4, 0)])
def test_self_attention_split(lengths, nheads, causal):
    """q, k, v and the probabilities as bf16 hi + lo pairs (3 MFMAs per product): ~2^-16 relative per product."""
    B, Fr, W = len(lengths), max(lengths), 64 * nheads
    qkv = rnd(B * Fr, 3 * W, scale=2.0)
    out = torch.full((B * Fr, W), float("nan"), device=DEV)
    qd = qkv.to(DEV)
    ld = None if causal else torch.tensor(lengths, dtype=torch.int32, device=DEV)
    _lib.check(lib().ladiff_self_attention_split(_lib.ptr(qd), None if ld is None else ld.data_ptr(), None, _lib.ptr(out), B, Fr,
                                                  nheads, causal, _lib.stream_ptr()))
    sync()
    q, k, v = qkv.double().view(B, Fr, 3, nheads, 64).permute(2, 0, 3, 1, 4)
    s = (q * 0.125) @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((Fr, Fr), float("-inf"), dtype=torch.float64).triu(1)
    else:
        s = s.masked_fill((~orc.lengths_to_mask(lengths, Fr))[:, None, None, :], float("-inf"))
    want = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * Fr, W)
    got = out.cpu()
    assert torch.isfinite(got).all()
    assert (got.double() - want).abs().max().item() < 2e-4                 # scores up to ~|30|: 2^-16 of that moves a softmax weight by 5e-4 relative


def test_self_attention_split_key_bitmap_matches_fp32_kernel():
    """Arbitrary 256-bit key maps (LA-VAE encoder): same masks as the fp32 kernel, values within the f16x3 budget."""
    B, Fr = 3, 206
    valid = torch.rand(B, Fr, generator=torch.Generator().manual_seed(4)) > 0.3
    valid[:, 0] = True
    qkv = rnd(B * Fr, 768, scale=1.5)
    bits = key_bitmap(valid).to(DEV)
    qd = qkv.to(DEV)
    a, b_ = torch.empty(B * Fr, 256, device=DEV), torch.empty(B * Fr, 256, device=DEV)
    _lib.check(lib().ladiff_decoder_self_attention(_lib.ptr(qd), None, bits.data_ptr(), _lib.ptr(a), B, Fr, _lib.stream_ptr()))
    _lib.check(lib().ladiff_self_attention_split(_lib.ptr(qd), None, bits.data_ptr(), _lib.ptr(b_), B, Fr, 4, 0, _lib.stream_ptr()))
    sync()
    assert (a - b_).abs().max().item() < 2e-4 and (a - b_).abs().max().item() > 0


# ---------------------------------------------------------------- fused decoder feed-forward block (csrc/dec_mlp.hip)
def split_rows(t):
    """fp32 [R,K] on the GPU -> its S-format twin (bf16 hi | lo blocks)."""
    s = torch.empty_like(t)
    _lib.check(lib().ladiff_split_rows(_lib.ptr(t), _lib.ptr(s), t.shape[0], t.shape[1], _lib.stream_ptr()))
    return s


def unsplit_rows(s):
    """S-format [R,K] -> fp32 (hi + lo), on the CPU."""
    R, K = s.shape
    half = torch.float16 if lib().ladiff_split_format() == 1 else torch.bfloat16
    b = s.cpu().contiguous().view(half).view(R, K // 64, 2, 64).float()
    return (b[:, :, 0] + b[:, :, 1]).reshape(R, K)


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("M,second_ln", [(1, False), (17, True), (480, False), (1000, True), (4133, False)])
def test_fused_mlp_layernorm(M, second_ln, variant):
    """y = LN(x + W2 gelu(W1 x + b1) + b2) [then a second LN] in one kernel (f16x3 products, everything else fp32) against
    fp64: row counts that leave waves / lanes of the last 128-row workgroup without rows; fp32 and S-format outputs agree.
    Asymmetric weights and inputs: a wrong row permutation of a weight panel or a wrong k order shows up as O(1) errors."""
    x = rnd(M, 256, scale=2.0, seed=1)
    w1, b1 = rnd(1024, 256, scale=1 / 16, seed=2), rnd(1024, scale=0.5, seed=3)
    w2, b2 = rnd(256, 1024, scale=1 / 32, seed=4), rnd(256, scale=0.5, seed=5)
    g3, be3 = 1 + 0.1 * rnd(256, seed=6), 0.1 * rnd(256, seed=7)
    g4, be4 = 1 + 0.1 * rnd(256, seed=8), 0.1 * rnd(256, seed=9)
    d = lambda t: t.to(DEV).contiguous()
    xd, w1d, w2d = d(x), d(w1), d(w2)
    xs, w1s, w2s = split_rows(xd), split_rows(w1d), split_rows(w2d)
    b1d, b2d, g3d, be3d, g4d, be4d = d(b1), d(b2), d(g3), d(be3), d(g4), d(be4)
    y = torch.full((M, 256), float("nan"), device=DEV)
    ys = torch.zeros(M, 256, device=DEV)
    # variant 0: the launcher's own choice (64-row workgroups at these sizes); 1 / 2 / 3: eight waves x 16 rows, four waves x 16 rows,
    # four waves x 32 rows
    assert lib().ladiff_debug_set_mlp_variant(variant) == 0
    try:
        _fused_mlp_case(M, second_ln, x, w1, b1, w2, b2, g3, be3, g4, be4, xd, xs, w1s, w2s, b1d, b2d, g3d, be3d, g4d, be4d, y, ys)
    finally:
        lib().ladiff_debug_set_mlp_variant(0)


def _fused_mlp_case(M, second_ln, x, w1, b1, w2, b2, g3, be3, g4, be4, xd, xs, w1s, w2s, b1d, b2d, g3d, be3d, g4d, be4d, y, ys):
    _lib.check(lib().ladiff_mlp_ln_fused(_lib.ptr(xs), _lib.ptr(xd), _lib.ptr(w1s), _lib.ptr(b1d), _lib.ptr(w2s), _lib.ptr(b2d),
                                         _lib.ptr(g3d), _lib.ptr(be3d), _lib.ptr(g4d) if second_ln else None,
                                         _lib.ptr(be4d) if second_ln else None, _lib.ptr(y), _lib.ptr(ys), M, _lib.stream_ptr()))
    sync()
    h = F.gelu(F.linear(x.double(), w1.double(), b1.double()))
    want = F.layer_norm(x.double() + F.linear(h, w2.double(), b2.double()), (256,), g3.double(), be3.double(), 1e-5)
    if second_ln:
        want = F.layer_norm(want, (256,), g4.double(), be4.double(), 1e-5)
    err = (y.cpu().double() - want).abs().max().item()
    assert torch.isfinite(y).all() and err < 2e-4, err                    # ~2^-16 per product, O(1) normalised outputs
    assert (unsplit_rows(ys).double() - y.cpu().double()).abs().max().item() < 1e-4     # the S-format twin: 16 significant bits
    # only one of the two outputs requested
    y2 = torch.full((M, 256), float("nan"), device=DEV)
    _lib.check(lib().ladiff_mlp_ln_fused(_lib.ptr(xs), _lib.ptr(xd), _lib.ptr(w1s), _lib.ptr(b1d), _lib.ptr(w2s), _lib.ptr(b2d),
                                         _lib.ptr(g3d), _lib.ptr(be3d), _lib.ptr(g4d) if second_ln else None,
                                         _lib.ptr(be4d) if second_ln else None, _lib.ptr(y2), None, M, _lib.stream_ptr()))
    sync()
    assert torch.equal(y, y2)


def test_fused_mlp_matches_three_launch_path_on_a_decode():
    """LADiffVae.decode in f16x3 mode with the feed-forward block fused (default) and as linear1 / linear2 / LayerNorm launches
    (the round-2 path, same products): the frames agree to rounding."""
    from ladiff_amd import LADiffVae, synthetic as syn
    from test_abi import ABL, VAE_KW
    vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(DEV).eval()
    vae.precision = "f16x3"
    lens = [196, 60, 120, 1, 77, 196, 48, 150, 33]
    z = torch.randn(5, len(lens), 256, generator=torch.Generator().manual_seed(4)).to(DEV)
    for i, l in enumerate(lens):
        z[-(-l // 48):, i] = 0
    try:
        assert lib().ladiff_debug_set_decoder_fusion(0) == 0
        a = vae.decode(z, lens)
        assert lib().ladiff_debug_set_decoder_fusion(2) == 0              # fused at every size (the default fuses from 10,000 rows up)
        b = vae.decode(z, lens)
    finally:
        lib().ladiff_debug_set_decoder_fusion(1)
    assert torch.isfinite(b).all() and (a - b).abs().max().item() < 5e-5 * max(1.0, a.abs().max().item())
    assert lib().ladiff_debug_set_decoder_fusion(3) != 0
    # few rows: the small-M GEMM routing (default) against the large-M kernels on the same decode
    try:
        assert lib().ladiff_debug_set_decoder_fusion(4) == 0
        c = vae.decode(z, lens)
    finally:
        lib().ladiff_debug_set_decoder_fusion(1)
    d = vae.decode(z, lens)
    assert (c - d).abs().max().item() < 5e-5 * max(1.0, c.abs().max().item())


@pytest.mark.parametrize("nfeats", [263, 251])
def test_final_layer_on_padded_split_tiles_matches_fp32_kernel(nfeats):
    """From 4,096 frame rows up the f16x3 decode runs final_layer on whole 128-column tiles of the zero-padded S-format weight and
    scatters the real columns into [B, F, C]; + 8 keeps the round-2 fp32-input kernel.  Same frames to f16x3 rounding, exact zeros on
    padded frames, for a padded batch and a ragged one (row scatter) and both feature counts."""
    from ladiff_amd import LADiffVae, synthetic as syn
    from test_abi import ABL, VAE_KW
    vae = LADiffVae(ABL, **{**VAE_KW, "nfeats": nfeats})
    vae.load_state_dict(syn.vae_weights(nfeats)); vae = vae.to(DEV).eval()
    vae.precision = "f16x3"
    for lens in ([196] * 24, [196, 60, 120, 1, 77, 196, 48, 150, 33] * 5):
        z = torch.randn(5, len(lens), 256, generator=torch.Generator().manual_seed(len(lens))).to(DEV)
        for i, l in enumerate(lens):
            z[-(-l // 48):, i] = 0
        try:
            assert lib().ladiff_debug_set_decoder_fusion(1 + 8) == 0
            a = vae.decode(z, lens)
        finally:
            lib().ladiff_debug_set_decoder_fusion(1)
        b = vae.decode(z, lens)
        assert b.shape == (len(lens), max(lens), nfeats) and torch.isfinite(b).all()
        assert (a - b).abs().max().item() < 5e-5 * max(1.0, a.abs().max().item())
        for i, l in enumerate(lens):
            assert (b[i, l:] == 0).all()


@pytest.mark.parametrize("lens", [[196] * 6, [196, 60, 120, 1, 77, 196, 48, 150, 33, 32, 64, 65], [1], [224, 200]])
def test_attention_with_in_proj_inside_matches_two_launches(lens):
    """f16x3 decode with the self-attention kernel that computes its head's q | k | v itself (default) against in_proj GEMM +
    attention kernel (+ 16): same frames to f16x3 rounding - padded batches (keys >= length masked, all F query rows computed),
    ragged batches, one-frame and 224-frame samples, lengths on both sides of a 32-row tile edge."""
    from ladiff_amd import LADiffVae, synthetic as syn
    from test_abi import ABL, VAE_KW
    vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(DEV).eval()
    vae.precision = "f16x3"
    z = torch.randn(5, len(lens), 256, generator=torch.Generator().manual_seed(len(lens))).to(DEV)
    for i, l in enumerate(lens):
        z[-(-l // 48):, i] = 0
    try:
        assert lib().ladiff_debug_set_decoder_fusion(1 + 16) == 0
        a = vae.decode(z, lens)
        assert lib().ladiff_debug_set_decoder_fusion(1 + 32) == 0          # at every size (the default starts at 4,096 frame rows)
        b = vae.decode(z, lens)
    finally:
        lib().ladiff_debug_set_decoder_fusion(1)
    assert lib().ladiff_debug_set_decoder_fusion(1 + 16 + 32) != 0
    assert b.shape == a.shape and torch.isfinite(b).all()
    assert (a - b).abs().max().item() < 5e-5 * max(1.0, a.abs().max().item()), (a - b).abs().max().item()
    for i, l in enumerate(lens):
        assert (b[i, l:] == 0).all()


def test_graphed_decode_matches_direct_decode():
    """Decodes of few frame rows are replayed from a hipGraph over persistent buffers (LADiffVae.graph_rows): same bits as the direct
    launch sequence, for padded and ragged batches, both arithmetic modes, changing inputs, alternating shapes (re-capture), and on
    the null stream (a private side stream is fenced in)."""
    from ladiff_amd import LADiffVae, synthetic as syn
    from test_abi import ABL, VAE_KW
    vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(DEV).eval()
    cases = [[60] * 8, [196, 60, 120, 1, 77], [48, 48]]
    for precision in ("f16x3", "fp32"):
        vae.precision = precision
        for rep in range(2):
            for lens in cases:
                z = torch.randn(5, len(lens), 256, generator=torch.Generator().manual_seed(10 * rep + len(lens))).to(DEV)
                for i, l in enumerate(lens):
                    z[-(-l // 48):, i] = 0
                vae.graph_rows = 0
                want = vae.decode(z, lens)
                vae.graph_rows = 4096
                got = vae.decode(z, lens)                      # null stream: runs on the fenced side stream
                assert torch.equal(got, want), (precision, lens)
                st = torch.cuda.Stream()
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    got2 = vae.decode(z, lens)
                st.synchronize()
                assert torch.equal(got2, want), (precision, lens)
    assert len(vae._dec_plans) <= 4
