"""The tagged hand-off's reader-side rule, checked by the loop kernel itself (test infrastructure: tests/test_gpu_pipeline.py runs this in
a child process).  libladiff_hip_diag.so = the product with csrc/systolic.hip built -DLADIFF_TAG_BITS=4 -DLADIFF_SELFCHECK: tags count 16
generations instead of 2; every look at handed-off rows whose LOW tag bit is right while the wide tag is not - a look the product's one-bit
tag would have accepted from another generation - is counted per stage type, and every stage reloads the rows it accepted and compares.
Both counters must stay 0 over the block geometries below (many blocks of mixed lengths: the attention stages dozens of blocks ahead
of the MLP stages; padding rows in every tile; one prompt; no guidance).
usage: handoff_diag_check.py [quick]      (environment: PACING, THRESH, XCD_LOCAL, STAGE_PLAN = the library's measurement switches)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ladiff_amd import _lib, build
_lib.LIB_PATH = build.diag_lib()
from ladiff_amd import LADIFF, DDIMScheduler, LADiffDenoiser, LADiffVae, synthetic as syn
from ladiff_amd.schema import ABL, DEN_KW, VAE_KW
dev = "cuda:0"
den = LADiffDenoiser(ABL, **DEN_KW); den.load_state_dict(syn.denoiser_weights())
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263))
den, vae = den.to(dev).eval(), vae.to(dev).eval()
NAMES = {0: "LIN", 1: "RED2", 2: "STYL", 3: "FFN", 4: "QKV", 5: "OUT", 6: "SKIP", 7: "TAIL"}
# the library's measurement switches, from the environment (PACING=eighths,mask  THRESH=look_ahead_from,small_upto  XCD_LOCAL  STAGE_PLAN)
L = _lib.lib()
for env, fn in (("PACING", "ladiff_debug_set_pacing"), ("THRESH", "ladiff_debug_set_loop_thresholds")):
    if env in os.environ: _lib.check(getattr(L, fn)(*[int(v) for v in os.environ[env].split(",")]))
for env, fn in (("XCD_LOCAL", "ladiff_debug_set_xcd_local"), ("STAGE_PLAN", "ladiff_debug_set_stage_plan")):
    if env in os.environ: _lib.check(getattr(L, fn)(int(os.environ[env])))
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
cases = [("mixed 256", syn.mixed_lengths(256), 7.5, 50, 6), ("mixed 1024", syn.mixed_lengths(1024), 7.5, 50, 3 if quick else 6),
         ("uniform 128", [196] * 128, 7.5, 50, 3), ("15-row tiles, 43 prompts", [196] * 43, 7.5, 20, 3),
         ("ragged 130", [max(1, min(196, 48 * ((i % 5) + 1) - 5 * (i % 3))) for i in range(130)], 7.5, 20, 3),
         ("one prompt", [120], 7.5, 50, 3), ("no guidance, 40 prompts", syn.mixed_lengths(40), 1.0, 20, 3)]
total = 0
for name, lens, guidance, steps, calls in cases:
    B = len(lens)
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                        set_alpha_to_one=False, steps_offset=1)
    pipe = LADIFF(denoiser=den, vae=vae, scheduler=sch, guidance_scale=guidance, num_inference_timesteps=steps, precision="f16x3", loop="pipeline16")
    text = syn.text_embeddings(B, seed=300 + B).to(dev)
    if guidance <= 1.0:
        text = text[B:].contiguous()
    noise = syn.init_noise(lens, seed=301 + B).to(dev)
    looks = changed = 0
    first = None
    with torch.no_grad():
        for _ in range(calls):
            z = pipe._diffusion_reverse(text, lens, init_noise=noise)
            torch.cuda.synchronize()
            assert pipe.loop_status()[0] == 0 and pipe.last_loop()[0], (name, pipe.loop_status(), pipe.last_loop())
            first = z.clone() if first is None else first
            assert torch.equal(z, first), f"{name}: a call differs from the first"
            for plan in pipe._plans.values():
                base = plan["status_dev"].storage_offset()
                w = plan["ws"][base + 16: base + 64].view(torch.int32).tolist()
                changed += w[8]                       # status[24]: rows that changed under a stage
                looks += w[16]                        # status[32]: looks a parity would have accepted from another generation
                if w[16]:
                    print(f"  {name}: " + "; ".join(f"{NAMES[i]} x{w[24 + i]}" for i in range(8) if w[24 + i]), flush=True)
    print(f"{name}: {calls} calls, {looks} looks from another generation, {changed} rows changed under a stage", flush=True)
    total += looks + changed
print("hand-off diagnostics:", "clean" if total == 0 else f"{total} events")
sys.exit(0 if total == 0 else 1)
