"""Stress (GPU box): random batch shapes through the persistent pipeline loop - every call twice (same bits), against the
launch-per-stage loop (tolerance), 16-row against 32-row plan (same bits), status word clean.  Complements
scripts/stress_pipeline.py (one shape, many repetitions).  python scripts/stress_shapes.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import synthetic as syn

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
dev = torch.device("cuda", 0)
bad = 0
for precision, tol in (("f16x3", 2e-4), ("fp32", 2e-5)):
    pipe = bench.build_pipe(dev, 128)
    pipe.precision = precision
    for case in range(cases):
        # beyond 320 prompts the call runs as several launches (balanced chunks <= 256, LADIFF.max_prompts_per_launch): the plan cache
        # (four shapes, each with its own sampler) turns over constantly in this loop
        B = rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 128, 144, 200, 233, 300, 321, 450, 520])
        guided = rng.random() < 0.8
        kind = rng.choice(["full", "mixed", "short", "bimodal"])
        lens = [196 if kind == "full" else rng.randint(1, 196) if kind == "mixed" else rng.randint(1, 60) if kind == "short"
                else rng.choice([20, 196]) for _ in range(B)]
        steps = rng.choice([2, 3, 7, 12])
        pipe.num_inference_timesteps = steps
        text, noise = syn.text_embeddings(B, seed=1000 + case).to(dev), syn.init_noise(lens, seed=2000 + case).to(dev)
        pipe.guidance_scale = 7.5 if guided else 1.0                 # no guidance: the conditional rows only, one-branch blocks
        pipe.do_classifier_free_guidance = guided
        if not guided:
            text = text[B:].contiguous()
        out = {}
        if os.environ.get("STRESS_VERBOSE"):
            print(f"{precision} case {case}: B={B} {kind} steps={steps} guided={guided} lens[:6]={lens[:6]}", file=sys.stderr, flush=True)
        with torch.no_grad():
            for loop in ("launches", "pipeline16", "pipeline16", "pipeline32" if guided else "pipeline"):
                pipe.loop = loop
                if os.environ.get("STRESS_VERBOSE"):
                    print(f"    {loop}", file=sys.stderr, flush=True)
                z = pipe._diffusion_reverse(text, lens, init_noise=noise)
                torch.cuda.synchronize()
                st = pipe.loop_status()
                if st != (0, 0):
                    bad += 1; print(f"{precision} case {case} B={B} {kind} steps={steps} {loop}: status {st}", flush=True)
                out.setdefault(loop, []).append(z)
        scale = max(1.0, out["launches"][0].abs().max().item())
        d_l = (out["pipeline16"][0] - out["launches"][0]).abs().max().item()
        same_rep = torch.equal(out["pipeline16"][0], out["pipeline16"][1])
        same_plan = torch.equal(out["pipeline16"][0], out["pipeline32" if guided else "pipeline"][0])
        if d_l > tol * scale or not same_rep or not same_plan:
            bad += 1
            print(f"{precision} case {case} B={B} {kind} steps={steps} guided={guided}: |pipeline - launches| = {d_l:.3e} (scale {scale:.1f}), "
                  f"repeat identical {same_rep}, plans identical {same_plan}", flush=True)
    print(f"{precision}: {cases} shapes done, {bad} bad so far", flush=True)
sys.exit(1 if bad else 0)
