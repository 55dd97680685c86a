"""Same-box A/B of the decode at the benchmark size (128 x 196 frames, f16x3, default fusion): HIP-event ms per decode for each library
named on the command line (paths relative to the repo; `product` = the shipped one), each with dec_mlp's workgroup forms (0 = by size,
1 = <8,1>), three rounds interleaved so that a drift of the box shows.  Every library runs in a child process of its own.
usage: decode_ab.py product ladiff_amd/libladiff_hip_x.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for rnd in range(3):
        for lib in sys.argv[1:]:
            env = dict(os.environ)
            if lib != "product": env["LADIFF_LIB"] = lib
            else: env.pop("LADIFF_LIB", None)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            print(f"round {rnd} {lib:44s} " + (r.stdout.strip() or r.stderr[-300:]), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from ladiff_amd import LADiffVae, _lib, synthetic as syn
from ladiff_amd.schema import ABL, VAE_KW
if os.environ.get("LADIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
dev = "cuda:0"
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(dev).eval()
vae.precision = "f16x3"
L = _lib.lib()
B, F = 128, 196
lens = [F] * B
z = torch.randn(5, B, 256, generator=torch.Generator().manual_seed(1)).to(dev)
for i, m in enumerate(syn.max_iter_elements(lens)):
    z[m:, i] = 0
out = []
for form in (0, 1):
    L.ladiff_debug_set_mlp_variant(form)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(5): vae.decode(z, lens)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(60): vae.decode(z, lens)
        e1.record(s)
        torch.cuda.synchronize()
    out.append(f"mlp form {form}: {e0.elapsed_time(e1) / 60:.4f} ms")
print(" | ".join(out))
