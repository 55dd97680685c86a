"""Determinism hunt at config c5's stated batch (1,024 prompts of mixed lengths, KIT 251 features, split mode): the same sample() N times,
every result against the first one bit for bit, latents and frames separately, and the decode alone on the first latents.
Knobs (environment): LADIFF_LIB (variant library), MLP_VARIANT, DEC_FUSION, HANDOFF, LOOP (pipeline / launches), PRECISION,
PACING=eighths,mask  THRESH=look_ahead_from,small_upto  XCD_LOCAL  STAGE_WAVES  STAGE_PLAN (the library's debug switches).
usage: race_hunt.py [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ladiff_amd import _lib
if os.environ.get("LADIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
from ladiff_amd import LADIFF, DDIMScheduler, LADiffDenoiser, LADiffVae, synthetic as syn
from ladiff_amd.schema import ABL, DEN_KW, VAE_KW
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
B, C = int(os.environ.get("BATCH", "1024")), 251
den = LADiffDenoiser(ABL, **DEN_KW); den.load_state_dict(syn.denoiser_weights())
vae = LADiffVae(ABL, **{**VAE_KW, "nfeats": C}); vae.load_state_dict(syn.vae_weights(C))
sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                    set_alpha_to_one=False, steps_offset=1)
pipe = LADIFF(denoiser=den.to(dev).eval(), vae=vae.to(dev).eval(), scheduler=sch, guidance_scale=7.5, num_inference_timesteps=50,
              precision=os.environ.get("PRECISION", "f16x3"), loop=os.environ.get("LOOP", "pipeline"))
L = _lib.lib()
if "MLP_VARIANT" in os.environ: _lib.check(L.ladiff_debug_set_mlp_variant(int(os.environ["MLP_VARIANT"])))
if "DEC_FUSION" in os.environ: _lib.check(L.ladiff_debug_set_decoder_fusion(int(os.environ["DEC_FUSION"])))
if "HANDOFF" in os.environ: _lib.check(L.ladiff_debug_set_handoff(int(os.environ["HANDOFF"])))
for env, fn in (("PACING", "ladiff_debug_set_pacing"), ("THRESH", "ladiff_debug_set_loop_thresholds")):
    if env in os.environ: _lib.check(getattr(L, fn)(*[int(v) for v in os.environ[env].split(",")]))
for env, fn in (("XCD_LOCAL", "ladiff_debug_set_xcd_local"), ("STAGE_WAVES", "ladiff_debug_set_stage_waves"), ("STAGE_PLAN", "ladiff_debug_set_stage_plan")):
    if env in os.environ: _lib.check(getattr(L, fn)(int(os.environ[env])))
lens = syn.mixed_lengths(B)
text, noise = syn.text_embeddings(B, seed=81).to(dev), syn.init_noise(lens, seed=82).to(dev)
print(f"split format {L.ladiff_split_format()}, B = {B}, knobs: " + " ".join(f"{k}={os.environ[k]}" for k in ("LADIFF_LIB", "MLP_VARIANT", "DEC_FUSION", "HANDOFF", "LOOP", "PRECISION", "PACING", "THRESH", "XCD_LOCAL", "STAGE_WAVES", "STAGE_PLAN") if k in os.environ), flush=True)
z0 = f0 = None
bad_z = bad_f = bad_d = 0
with torch.no_grad():
    for it in range(n):
        z, feats = pipe.sample(text, lens, init_noise=noise)
        torch.cuda.synchronize(); pipe.check()
        if os.environ.get("SELFCHECK"):              # a -DLADIFF_SELFCHECK library: the loop kernel's own record of rows that changed under a stage
            for plan in pipe._plans.values():
                st = plan["status_dev"]
                base = st.storage_offset()
                rec = plan["ws"][base + 16: base + 32].view(torch.int32).tolist()
                aba = plan["ws"][base + 32: base + 64].view(torch.int32).tolist()
                if aba[0]:
                    names = {0: "LIN", 1: "RED2", 2: "STYL", 3: "FFN", 4: "QKV", 5: "OUT", 6: "SKIP", 7: "TAIL"}
                    parts = []
                    for ri in range(8):
                        if aba[8 + ri]:
                            w, v = aba[16 + 2 * ri] & 0xffffffff, aba[17 + 2 * ri] & 0xffffffff
                            parts.append(f"{names[ri]} x{aba[8 + ri]} (first: layer {(w >> 8) & 0xff} slice {(w >> 16) & 0xff} where {w >> 24} step {v >> 24} block {(v >> 8) & 0xffff} wave {v & 0xff} tag xor {w & 0xff:#x})")
                    print(f"  run {it}: A-B-A looks {aba[0]}: " + "; ".join(parts) + (f"; RED2 residual words {aba[28] & 0xffffffff:#010x} {aba[29] & 0xffffffff:#010x} {aba[30] & 0xffffffff:#010x} row {aba[31]}" if aba[28] else ""), flush=True)
                if rec[0] or rec[8]:
                    w = rec[1]
                    print(f"  run {it}: self-check: {rec[8]} lane records; first: role bit {w & 0xff} layer {(w >> 8) & 0xff} slice {(w >> 16) & 0xff} where {(w >> 24) & 0xff} "
                          f"step {rec[2]} block {rec[3]} wave {rec[4] & 0xff} lane {rec[4] >> 8} unit mask {rec[5] & 0xffffffff:#x} used {rec[6] & 0xffffffff:#010x} reloaded {rec[7] & 0xffffffff:#010x}", flush=True)
        if z0 is None:
            z0, f0 = z.clone(), feats.clone(); continue
        dz = (z - z0).abs().amax(dim=(0, 2)); df = (feats - f0).abs().amax(dim=(1, 2))
        if dz.max().item() != 0: bad_z += 1; print(f"  run {it}: latents differ on prompts {torch.nonzero(dz).flatten().tolist()[:12]} max {dz.max().item():.3e}", flush=True)
        if df.max().item() != 0: bad_f += 1; print(f"  run {it}: frames differ on prompts {torch.nonzero(df).flatten().tolist()[:12]} max {df.max().item():.3e}", flush=True)
    for it in range(n):                      # the decode alone, same latents
        feats = pipe.vae.decode(z0, lens)
        torch.cuda.synchronize()
        df = (feats - f0).abs().amax(dim=(1, 2))
        if df.max().item() != 0: bad_d += 1; print(f"  decode {it}: frames differ from the first sample()'s on prompts {torch.nonzero(df).flatten().tolist()[:12]} max {df.max().item():.3e}", flush=True)
print(f"{n - 1} repeats: latents differ in {bad_z}, frames in {bad_f}; {n} decodes of the first latents: {bad_d} differ")
