"""Graph captures beside torch.distributed's watchdog thread (profiles/r5/26_*).  One process, world size 1, RCCL: a collective on the
stream, then at once a call that captures (the arithmetic mode alternates, so every call re-captures; the launch-per-stage loop, whose
step graphs take several ms to capture), N times.  With captures on the CALLER's stream (round 5 before the fix) this died within 150
pairs: the watchdog polls the collective's end event, which belongs to the current stream, and a hipEventQuery of an event of a
capturing stream is refused and invalidates the capture.  The library now captures on a stream of the handle's own (csrc/api.hip,
Sampler::cap).  usage: capture_vs_watchdog.py [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29701", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
import bench
from ladiff_amd import synthetic as syn
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
pipe = bench.build_pipe(dev, 16); pipe.num_inference_timesteps = 10
pipe.loop = "launches"                                       # step graphs of ~100 launches each: a capture of several ms, a wide window
B = 16; lens = [196] * B
text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
t = torch.ones(1 << 20, device=dev)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream), torch.no_grad():
    for i in range(n):
        dist.all_reduce(t)                                   # pending in the watchdog's list until it has polled it complete
        pipe.precision = "fp32" if i & 1 else "f16x3"       # a new capture key: the library captures its graphs again
        pipe._diffusion_reverse(text, lens, init_noise=noise)
    torch.cuda.synchronize()
pipe.check(); dist.destroy_process_group()
print(f"{n} collective + capture pairs done", flush=True)
