"""Stress (GPU box): ONE LADIFF object driven through random sequences - arithmetic mode, loop form, guidance, batch shape, step count
and scheduler eta change from call to call (plans, samplers, graphs and weight tables turn over), sample() = reverse loop + decode.
Every call is checked against the SAME call on a fresh object with the launch-per-stage loop in fp32 (tolerance), repeated once (same
bits), its status read.  python scripts/stress_sample.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import synthetic as syn

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 21)
dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128)
ref_pipe = bench.build_pipe(dev, 128)
ref_pipe.precision = "fp32"; ref_pipe.loop = "launches"
bad = 0
for case in range(cases):
    precision = rng.choice(["f16x3", "f16x3", "fp32"])
    loop = rng.choice(["pipeline", "pipeline", "pipeline16", "pipeline32", "launches"])
    guided = rng.random() < 0.8
    if not guided and loop == "pipeline32":
        loop = "pipeline"
    B = rng.choice([1, 2, 5, 8, 16, 33, 64, 128, 170, 200, 256, 330])
    kind = rng.choice(["full", "mixed", "short"])
    lens = [196 if kind == "full" else rng.randint(1, 196) if kind == "mixed" else rng.randint(1, 60) for _ in range(B)]
    steps = rng.choice([2, 3, 5])
    for p in (pipe, ref_pipe):
        p.num_inference_timesteps = steps
        p.guidance_scale = 7.5 if guided else 1.0
        p.do_classifier_free_guidance = guided
    pipe.precision = precision
    pipe.loop = loop
    text, noise = syn.text_embeddings(B, seed=3000 + case).to(dev), syn.init_noise(lens, seed=4000 + case).to(dev)
    if not guided:
        text = text[B:].contiguous()
    print(f"case {case}: {precision} {loop} guided={guided} B={B} {kind} steps={steps}", file=sys.stderr, flush=True)
    with torch.no_grad():
        z1, f1 = pipe.sample(text, lens, init_noise=noise)
        st = pipe.loop_status()
        z2, f2 = pipe.sample(text, lens, init_noise=noise)
        zr, fr = ref_pipe.sample(text, lens, init_noise=noise)
    torch.cuda.synchronize()
    msgs = []
    if st != (0, 0): msgs.append(f"status {st}")
    if not (torch.equal(z1, z2) and torch.equal(f1, f2)): msgs.append("repeat differs")
    tol = 2e-3 if precision == "f16x3" else 2e-4
    dz = (z1 - zr).abs().max().item() / max(1.0, zr.abs().max().item())
    df = (f1 - fr).abs().max().item() / max(1.0, fr.abs().max().item())
    if not (dz < tol and df < tol): msgs.append(f"vs fp32 launches: latents {dz:.2e} frames {df:.2e}")
    if not torch.isfinite(f1).all(): msgs.append("non-finite")
    if msgs:
        bad += 1
        print(f"case {case}: {precision} {loop} guided={guided} B={B} {kind} steps={steps}: " + "; ".join(msgs), flush=True)
print(f"{cases} sample() cases done, {bad} bad", flush=True)
sys.exit(1 if bad else 0)
