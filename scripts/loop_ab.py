"""Same-box A/B of the pipeline loop kernel between builds of the library: loop kernel ms (HIP events inside the library, calls queued back to
back - calls separated by a host synchronisation read 3 - 5 % faster, NOTEBOOK round 4) at 64 / 128 / 128 mixed / 256 prompts, 50 steps, and a
checksum of the latents (same bits expected unless a build re-associates).  Every library in a child process, three interleaved rounds.
usage: loop_ab.py product ladiff_amd/libladiff_hip_x.so ...     (LOOP_AB_CASES="64,u 128,u" to choose)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for rnd in range(int(os.environ.get("LOOP_AB_ROUNDS", "3"))):
        for lib in sys.argv[1:]:
            env = dict(os.environ)
            if lib != "product": env["LADIFF_LIB"] = lib
            else: env.pop("LADIFF_LIB", None)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            print(f"round {rnd} {lib:40s} " + (r.stdout.strip().replace(chr(10), " | ") or r.stderr[-400:]), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from ladiff_amd import _lib, synthetic as syn
if os.environ.get("LADIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
import bench
dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128)
pipe.precision = os.environ.get("LOOP_AB_PRECISION", "f16x3")
stream = torch.cuda.Stream(device=dev)
cases = [c.split(",") for c in os.environ.get("LOOP_AB_CASES", "64,u 128,u 128,m 256,u").split()]
for B, kind in cases:
    B = int(B)
    lens = [196] * B if kind == "u" else ([196, 60, 120] * 200)[:B]
    text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
    ms = []
    with torch.cuda.stream(stream), torch.no_grad():
        for _ in range(3): z = pipe._diffusion_reverse(text, lens, init_noise=noise)
        for rep in range(3):
            for _ in range(6): z = pipe._diffusion_reverse(text, lens, init_noise=noise)
            torch.cuda.synchronize()
            ms.append(pipe.loop_ms())
    st = pipe.loop_status()
    print(f"{B}{kind} {sorted(ms)[1]:.3f} ms (chk {z.double().sum().item():+.6e}{'' if st[0] == 0 else ' STATUS ' + str(st)})")
