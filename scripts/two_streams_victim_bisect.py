"""(1) Which trivial host call after a decode's launches disturbs it?  (2) Which launch of the victim is the one that goes wrong: the
victim is cut after n launches (twin, LADIFF_DEC_CUT) and its WHOLE workspace is compared with a reference run of the same cut.
usage: decode_victim6.py [runs]"""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import LADiffVae, _lib, synthetic as syn
from test_abi import ABL, VAE_KW
if os.environ.get("LADIFF_LIB"):                      # the diagnostic twin (same ABI)
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
dev = "cuda:0"
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(dev).eval()
vae.precision = "f16x3"
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
g = torch.Generator().manual_seed(1)
B, F, T, C = 64, 196, 5, 263
z = torch.randn(T, B, 256, generator=g).to(dev)
z2 = torch.randn(T, B, 256, generator=g).to(dev)
A, Bs = torch.cuda.Stream(), torch.cuda.Stream()
lens = [F] * B; counts = [5] * B
L = _lib.lib()
wsb = L.ladiff_decoder_workspace_bytes(B, F, T, C)
wt = vae._weight_table(); wsp = wt.split_array()
with torch.cuda.stream(A):
    featsA = torch.zeros(B, F, C, device=dev); wsA = torch.zeros((wsb + 3) // 4, device=dev)
    lA = _lib.device_ints(lens, dev); cA = _lib.device_ints(counts, dev)
with torch.cuda.stream(Bs):
    featsB = torch.zeros(B, F, C, device=dev); wsB = torch.zeros((wsb + 3) // 4, device=dev)
    lB = _lib.device_ints(lens, dev); cB = _lib.device_ints(counts, dev)
torch.cuda.synchronize()
def victim(cut):
    if cut is None: os.environ.pop("LADIFF_DEC_CUT", None)
    else: os.environ["LADIFF_DEC_CUT"] = str(cut)
    _lib.check(L.ladiff_vae_decode(wt.array, wsp, z.data_ptr(), lA.data_ptr(), cA.data_ptr(), B, F, T, C, featsA.data_ptr(), wsA.data_ptr(), wsb, A.cuda_stream))
AGG_CUT = "0"
def agg_decode0():
    os.environ["LADIFF_DEC_CUT"] = AGG_CUT
    _lib.check(L.ladiff_vae_decode(wt.array, wsp, z2.data_ptr(), lB.data_ptr(), cB.data_ptr(), B, F, T, C, featsB.data_ptr(), wsB.data_ptr(), wsb, Bs.cuda_stream))
libc = ctypes.CDLL(None)
def agg_version(): L.ladiff_version()
def agg_getpid(): libc.getpid()
def agg_wsbytes():
    for _ in range(5): L.ladiff_decoder_workspace_bytes(B, F, T, C)
victim(None); torch.cuda.synchronize(); ref = featsA.clone()
print("part 1: the whole decode as victim, launched through the C entry with a fixed workspace")
for name, fn in (("nothing", None), ("ladiff_vae_decode cut at 0", agg_decode0)):
    bad = 0
    for it in range(runs):
        victim(None)
        if fn is not None: fn()
        torch.cuda.synchronize()
        if not torch.equal(featsA, ref): bad += 1
    print(f"  host then calls {name:38s}: decode differs in {bad} of {runs} runs", flush=True)
AGG_CUT = "100"
print("part 2: victim cut after n launches, aggressor = a whole decode; regions of the victim's workspace that differ from a run of the same cut alone")
names = ["P0", "P1", "P2", "P3", "SK0", "SK1", "SK2", "SK3", "Ps0", "Ps1", "Ps2", "Ps3", "SKs0", "SKs1", "SKs2", "SKs3", "qkv", "qkv", "qkv", "att", "hid", "hid", "hid", "hid", "rest"]
MD = B * F * 256
for cut in [int(a) for a in sys.argv[2:]] or (8, 11, 14, 17, 18, 19, 20, 21, 22, 23, 24, 28, 32, 36, 38):
    wsA.zero_(); torch.cuda.synchronize()
    victim(cut); torch.cuda.synchronize(); wref = wsA.clone()
    bad = 0; regions = {}
    for it in range(runs):
        victim(cut); agg_decode0(); torch.cuda.synchronize()
        if not torch.equal(wsA, wref):
            bad += 1
            idx = (wsA != wref).nonzero().flatten()
            for r in set((idx // MD).tolist()): regions[names[min(r, 24)]] = regions.get(names[min(r, 24)], 0) + 1
    print(f"  victim cut after {cut:2d} launches: workspace differs in {bad} of {runs} runs; regions {regions}", flush=True)
