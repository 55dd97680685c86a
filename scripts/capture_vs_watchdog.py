"""Does a hipEventQuery of torch.distributed's watchdog thread that falls into the library's graph capture take the process down
(profiles/r5/26_*)?  One process, world size 1, RCCL: a collective, then at once a call that captures (the arithmetic mode alternates, so
every call re-captures; the launch-per-stage loop, whose step graphs take several ms to capture), N times - with and without the loop owner's pause (LADIFF.capture_guard).  Each mode runs in a child process;
the parent reports how it ended.  usage: capture_vs_watchdog.py [iterations]        (child: capture_vs_watchdog.py child <guard> <n>)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    guard, n = sys.argv[2] == "1", int(sys.argv[3])
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + (1 if guard else 0)), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch, torch.distributed as dist
    import bench
    from ladiff_amd import synthetic as syn
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    pipe = bench.build_pipe(dev, 16); pipe.num_inference_timesteps = 10; pipe.capture_guard = guard
    pipe.loop = "launches"                                       # step graphs of ~100 launches each: a capture of several ms, a wide window
    B = 16; lens = [196] * B
    text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
    t = torch.ones(1 << 20, device=dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream), torch.no_grad():
        for i in range(n):
            dist.all_reduce(t)                                   # pending in the watchdog's list until it has polled it complete
            pipe.precision = "fp32" if i & 1 else "bf16x3"       # a new capture key: the library captures its graphs again
            pipe._diffusion_reverse(text, lens, init_noise=noise)
            if i % 20 == 19: print(f"  {i + 1} captures", flush=True)
        torch.cuda.synchronize()
    pipe.check(); dist.destroy_process_group()
    print("child done", flush=True)
    sys.exit(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for guard in (0, 1):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(guard), str(n)], capture_output=True, text=True, timeout=1500)
    last = [l for l in r.stdout.splitlines() if l.strip()][-1:] or ["-"]
    err = [l for l in r.stderr.splitlines() if "LadiffHipError" in l or "watchdog thread terminated" in l][:2]
    print(f"capture_guard {guard}: {n} collective + capture pairs -> exit code {r.returncode}, last line '{last[0].strip()}'" + ("".join("\n    " + e[:220] for e in err)), flush=True)
