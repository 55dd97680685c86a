import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from ladiff_amd import synthetic as syn
dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128); pipe.precision = "bf16x3"
vae = pipe.vae; vae.length_aware = False
main = torch.cuda.Stream(device=dev)
sides = [torch.cuda.Stream(device=dev) for _ in range(3)]
nb, F = 43, 120
z = torch.randn(5, nb, 256, device=dev)
lens = [F] * nb
def once(streams):
    for st in streams:
        st.wait_stream(main)
        with torch.cuda.stream(st):
            vae.decode(z, lens)
    for st in streams:
        main.wait_stream(st)
with torch.no_grad(), torch.cuda.stream(main):
    for k in (1, 2, 3):
        for _ in range(3): once(sides[:k])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(main)
        for _ in range(5): once(sides[:k])
        e1.record(main)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"{k} concurrent decodes of B={nb} F={F}: {e0.elapsed_time(e1) / 5:.3f} ms per round; host enqueue {1e3 * (t1 - t0) / 5:.3f} ms per round")
