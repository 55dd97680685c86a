"""`LADIFF`-compatible owner of the sampling loop (SURVEY.md §8b "Loop owner").

The reference's loop is a *method* of its LightningModule (`LADIFF._diffusion_reverse`,
`src/ladiff/models/modeltype/ladiff.py:333-571`), not a plugin, so the drop-in for it is a class with the same
method names and call behaviour: `forward(batch)` (ladiff.py:250-308), `_diffusion_reverse(text_emb, lengths)`,
`gen_from_latent(batch)` (:310-318) and the attributes callers touch (`sample_mean`, `fact`, `times`, `cfg`,
`feats2joints`, `guidance_scale`, `do_classifier_free_guidance`).  The whole reverse loop - hoisted time/text
tables, N x (denoiser, guidance, scheduler step) captured as one hipGraph step replayed N times, final masking -
is ONE call into libladiff_hip.so; the decoder is a second one.  Nothing here falls back to PyTorch math.

Out of scope (not built): training/eval steps, metrics, losses, the CLIP text encoder (pass any callable
`texts -> [len(texts),1,768]`), `feats2joints` (pass the datamodule's).
"""
import importlib
import math
import time
import ctypes
from ctypes import c_void_p, byref

import numpy as np
import torch
from torch import nn

from . import _lib
from .feats2joints import Feats2Joints
from .schedulers import DDIMScheduler, DDPMScheduler, timestep_sinusoid

_TARGET_ALIASES = {
    # reference dotted paths -> this package (so an unmodified reference YAML also resolves)
    "ladiff.models.architectures.ladiff_denoiser.LADiffDenoiser": "ladiff_amd.modules.LADiffDenoiser",
    "ladiff.models.architectures.ladiff_vae.LADiffVae": "ladiff_amd.modules.LADiffVae",
    "ladiff.models.architectures.mld_clip.MldTextEncoder": "ladiff_amd.text_encoder.MldTextEncoder",
    "ladiff.models.architectures.t2m_textenc.TextEncoderBiGRUCo": "ladiff_amd.evaluators.TextEncoderBiGRUCo",
    "ladiff.models.architectures.t2m_motionenc.MovementConvEncoder": "ladiff_amd.evaluators.MovementConvEncoder",
    "ladiff.models.architectures.t2m_motionenc.MotionEncoderBiGRUCo": "ladiff_amd.evaluators.MotionEncoderBiGRUCo",
    "diffusers.DDIMScheduler": "ladiff_amd.schedulers.DDIMScheduler",
    "diffusers.DDPMScheduler": "ladiff_amd.schedulers.DDPMScheduler",
}


def _cfg_get(node, key, default=None):
    if node is None:
        return default
    if isinstance(node, dict):
        return node.get(key, default)
    try:
        return node[key] if key in node else default
    except TypeError:
        return getattr(node, key, default)


def instantiate_from_config(config):
    """`{target: "pkg.mod.Class", params: {...}}` -> object (the reference's plugin API, src/ladiff/config.py:16-33)."""
    target = _cfg_get(config, "target")
    if target is None:
        raise KeyError("Expected key `target` to instantiate.")
    target = _TARGET_ALIASES.get(target, target)
    module, cls = target.rsplit(".", 1)
    params = _cfg_get(config, "params", {}) or {}
    return getattr(importlib.import_module(module), cls)(**dict(params))


def remove_padding(tensors, lengths):
    return [t[:l] for t, l in zip(tensors, lengths)]


class LADIFF(nn.Module):
    def __init__(self, cfg=None, datamodule=None, *, denoiser=None, vae=None, scheduler=None, text_encoder=None,
                 guidance_scale=None, num_inference_timesteps=None, eta=None, max_it=None, frame_per_latent=None,
                 test_efficiency=None, use_graph=True, precision=None, loop="pipeline", fallback=False,
                 max_prompts_per_launch=320, **kwargs):
        super().__init__()
        self.cfg = cfg
        self.datamodule = datamodule
        model = _cfg_get(cfg, "model")
        abl = _cfg_get(_cfg_get(cfg, "TRAIN"), "ABLATION")
        sch_cfg = _cfg_get(model, "scheduler")

        def pick(explicit, node, key, default):
            return explicit if explicit is not None else _cfg_get(node, key, default)

        self.guidance_scale = float(pick(guidance_scale, model, "guidance_scale", 7.5))
        self.num_inference_timesteps = int(pick(num_inference_timesteps, sch_cfg, "num_inference_timesteps", 20))
        self.eta = float(pick(eta, sch_cfg, "eta", 0.0))
        self.max_it = int(pick(max_it, abl, "MAX_IT", 5))
        self.frame_per_latent = int(pick(frame_per_latent, abl, "FRAME_PER_LATENT", 48))
        self.test_efficiency = bool(pick(test_efficiency, abl, "TEST_EFFICIENCY", False))
        if _cfg_get(cfg, "ARDIFF", False) or _cfg_get(abl, "JOINT_DISTRO_FIX", False):
            raise NotImplementedError("ARDIFF / JOINT_DISTRO_FIX branches of _diffusion_reverse are not built")
        self.denoiser = denoiser if denoiser is not None else instantiate_from_config(_cfg_get(model, "denoiser"))
        self.vae = vae if vae is not None else instantiate_from_config(_cfg_get(model, "motion_vae"))
        self.scheduler = scheduler if scheduler is not None else instantiate_from_config(sch_cfg)
        te_cfg = _cfg_get(model, "text_encoder")
        self.text_encoder = text_encoder if text_encoder is not None else (
            instantiate_from_config(te_cfg) if te_cfg is not None else None)
        self.latent_dim = [self.max_it, 256]
        self.do_classifier_free_guidance = self.guidance_scale > 1.0
        self.feats2joints = getattr(datamodule, "feats2joints", None)      # the reference's CPU function (HumanML3D.py:44-48)
        # same arithmetic on the GPU when the datamodule exposes mean / std / njoints; joints then cross PCIe, not features
        self.feats2joints_device = Feats2Joints.from_datamodule(datamodule) if datamodule is not None else None
        self.sample_mean = False
        self.fact = None
        self.times = []
        self.use_graph = use_graph
        # how the N steps run (include/ladiff_hip.h, ladiff_sampler_set_loop): "pipeline" = one persistent weight-stationary
        # kernel for the whole loop when the call qualifies (guidance on, one text token); its block geometry is planned per call
        # ("pipeline16" / "pipeline32" force the length-aware 16-row / the padded 32-row blocks), "launches" = one launch per
        # stage in hipGraphs
        if loop not in ("pipeline", "pipeline16", "pipeline32", "launches"):
            raise ValueError(f"loop {loop!r} not supported")
        self.loop = loop
        # matrix-product arithmetic of the denoiser loop: "fp32" (fp32-input MFMA) or "f16x3" (3-term split: operands as hi + lo pairs of fp16)
        self.precision = precision if precision is not None else getattr(self.denoiser, "precision", "fp32")
        # What happens when the persistent pipeline kernel abandons a call (a stage timed out on its producer: the GPU was shared with
        # another process, or a long kernel on another stream kept a CU busy).  The 8-byte status is copied to pinned host memory
        # behind every call; it is read when the host next synchronises anyway (sample, forward, t2m_eval, loop_ms, check()), and a
        # new call looks at its predecessor's words if they have arrived and waits only for the call before that (the host can queue
        # one call ahead); z is NaN in the meantime.  fallback=False: raise LadiffHipError.  fallback=True: the call waits for
        # its own status and, when it aborted, runs again launch-per-stage in this process (same library, same arithmetic; counted
        # in `fallback_count`).
        self.fallback = bool(fallback)
        self.fallback_count = 0
        self._pending = []            # (event, pinned status words, plan key, call number) per launch not yet looked at
        self._call = 0                # number of the current `_diffusion_reverse` call
        self._window_timing = False   # per-window events wanted (window_ms(enable=True)): applied to every sampler, also later ones
        self.noise_first_prompt = 0   # global index of this object's prompt 0 (a rank of a sharded batch sets its offset): keys the device noise
        self.last_noise_seed = None
        self._fault = (-1, 0)         # fault injection of the abort-path tests: applied to every sampler of THIS object (set_pipeline_fault)
        self._stream = None
        self._plans = {}              # plan key -> persistent buffers + sampler (a few shapes stay cached: chunks, alternating batches)
        self._last = []               # the plans the last call ran on, in order
        # Batches larger than this are run as several launches of the loop (balanced chunks of <= 256 prompts): the pipeline's
        # per-layer buffers of more than ~170 blocks fall out of the 256 MiB memory-side cache (512 prompts in one launch cost 11.5 ms
        # per 128 against 9.8 ms at 256).  None = never split.
        self.max_prompts_per_launch = max_prompts_per_launch

    # ------------------------------------------------------------------ plumbing
    @property
    def precision(self):
        return self._precision

    @precision.setter
    def precision(self, value):
        _lib.is_split(value)                               # raises on an unknown name ("fp32" | "f16x3"; "bf16x3" / "split": the same path)
        self._precision = value
        self.denoiser.precision = value
        self.vae.precision = value
        if hasattr(self.text_encoder, "precision"):
            self.text_encoder.precision = value

    @property
    def device(self):
        return next(self.denoiser.parameters()).device

    @property
    def _plan(self):
        return self._last[-1] if self._last else None

    @property
    def _sampler(self):
        return self._last[-1]["sampler"] if self._last else None

    def __del__(self):
        try:
            for ev, host, _, _ in self._pending:
                ev.synchronize()
                if int(host[0]) != 0:
                    import warnings
                    warnings.warn(f"LADIFF: the last pipeline loop aborted (workgroup {int(host[1])} timed out) and its result was never checked")
        except Exception:
            pass
        try:
            for plan in self._plans.values():
                if plan.get("sampler") is not None:
                    _lib.lib().ladiff_sampler_destroy(plan["sampler"])
        except Exception:
            pass

    def _get_plan(self, B, T, n_steps, eta, dev, n_text=1):
        """Persistent device buffers + scheduler tables + sampler for one (B, T, schedule): hipGraph kernel nodes bake
        pointers in, and the per-step scalar tables are built once, not per call."""
        sch = self.scheduler
        key = (B, T, n_steps, float(eta), str(dev), id(sch), n_text)
        plan = self._plans.get(key)
        if plan is not None:
            self._plans[key] = self._plans.pop(key)          # most recently used last
            return plan
        L = _lib.lib()
        sch.set_timesteps(n_steps)
        n_steps = len(sch.timesteps)
        need_noise = sch.needs_noise(eta)
        wsb = L.ladiff_reverse_workspace_bytes(B, T, n_steps, n_text)
        plan = {
            "key": key, "n": n_steps, "need_noise": need_noise,
            "timesteps": sch.timesteps.clone(),
            "coef": sch.coef_table(eta).to(dev),
            "sinus": timestep_sinusoid(sch.timesteps, 768).to(dev),
            "text": torch.empty(2 * B, n_text, 768, dtype=torch.float32, device=dev), "n_text": n_text,
            "noise": torch.empty(B, T, 256, dtype=torch.float32, device=dev),
            "counts": torch.empty(B, dtype=torch.int32, device=dev),
            # per-step noise (DDPM / eta > 0): the CALLER's tensor is used in place when it is a contiguous fp32 tensor on this device
            # (1000 steps x 128 prompts are 655 MB: not held twice, not copied per call); otherwise a plan-owned copy, made on first need
            "step_noise": None,
            "z": torch.empty(T, B, 256, dtype=torch.float32, device=dev),
            "ws": _lib.workspace(wsb, dev), "ws_bytes": wsb,
            "tables_key": None,      # weights the time tables inside `ws` were built from
            "sampler": None,
        }
        if self.use_graph:
            h = c_void_p()
            _lib.check(L.ladiff_sampler_create(byref(h)))
            plan["sampler"] = h
            if self._fault != (-1, 0):
                _lib.check(L.ladiff_sampler_set_fault(h, *self._fault))
            if self._window_timing:
                _lib.check(L.ladiff_sampler_set_window_timing(h, 1))
        # the pipeline kernel's {code, info} words inside the workspace, and where the host reads them
        off = L.ladiff_reverse_status_offset_bytes(B, T, n_steps, n_text)
        if off == 0 or off % 4:
            raise _lib.LadiffHipError("ladiff_reverse_status_offset_bytes rejected the plan's shape")
        plan["status_dev"] = plan["ws"][off // 4: off // 4 + 2].view(torch.int32)
        # one pinned {code, info} slot + event per launch of a call (a chunked batch can run the same plan several times)
        plan["status_host"] = torch.zeros(8, 2, dtype=torch.int32).pin_memory()
        plan["status_event"] = [torch.cuda.Event() for _ in range(8)]
        plan["launches"] = 0
        while len(self._plans) >= 4:                         # a few shapes stay cached; the oldest goes (its graphs with it)
            old = self._plans.pop(next(iter(self._plans)))
            if old.get("sampler") is not None:
                _lib.check(L.ladiff_sampler_destroy(old["sampler"]))
        self._plans[key] = plan
        return plan

    def set_pipeline_fault(self, workgroup=-1, timeout_ms=0):
        """Test aid (tests/test_gpu_pipeline.py): the pipeline launches of THIS object's samplers lose workgroup `workgroup` right after the
        start-up handshake and bound every wait to `timeout_ms` (-1, 0: off).  Other LADIFF objects of the process are not affected."""
        self._fault = (int(workgroup), int(timeout_ms))
        for plan in self._plans.values():
            if plan.get("sampler") is not None:
                _lib.check(_lib.lib().ladiff_sampler_set_fault(plan["sampler"], *self._fault))

    def check(self, wait=True, before=None):
        """Look at the status of the `_diffusion_reverse` calls not looked at yet: raises LadiffHipError when a pipeline loop of one
        was abandoned (the returned z is NaN then).  wait=False only looks if the copies have already arrived; `before` restricts
        the look to calls numbered below it.  Returns True when everything asked for was looked at."""
        mine = [e for e in self._pending if before is None or e[3] < before]
        if not mine:
            return True
        if not wait and not all(e[0].query() for e in mine):
            return False
        self._pending = [e for e in self._pending if not (before is None or e[3] < before)]
        for ev, host, _, _ in mine:
            ev.synchronize()
            code, info = int(host[0]), int(host[1])
            if code != 0:
                raise _lib.LadiffHipError(
                    f"the persistent pipeline loop was abandoned (status {code}: workgroup {info} timed out waiting for its producer - "
                    "is the GPU shared, or was a long kernel running on another stream?); the latents of that call are NaN.  "
                    "Run again, construct LADIFF(fallback=True) to re-run such a call launch-per-stage automatically, or use loop='launches'")
        return True

    def _counts(self, lengths):
        return [int(math.ceil(l / self.frame_per_latent)) for l in lengths]

    # ------------------------------------------------------------------ the hot loop
    def _chunks(self, B):
        """[lo, hi) prompt ranges of the launches a batch of B prompts runs as: balanced chunks of at most min(cap, 256) prompts (beyond
        ~170 blocks the per-layer buffers of one launch fall out of the memory-side cache)."""
        cap = self.max_prompts_per_launch
        if cap is None or B <= cap or self.loop == "launches":
            return [(0, B)]
        n = -(-B // max(1, min(int(cap), 256)))
        base, extra = divmod(B, n)
        spans, lo = [], 0
        for i in range(n):
            hi = lo + base + (1 if i < extra else 0)
            spans.append((lo, hi))
            lo = hi
        return spans

    def _diffusion_reverse(self, encoder_hidden_states, lengths=None, init_noise=None, step_noise=None, noise_seed=None):
        """text_emb [2B,1,768] (unconditional half first), lengths list[int] -> z [max_it, B, 256]  (ladiff.py:333-571).
        Stochastic schedules (DDPM, eta > 0): `step_noise` [n,B,T,256] if given; otherwise the noise is drawn on the device where it
        is consumed (csrc/noise_gen.h) from `noise_seed` (default: a seed taken from torch's CPU generator, so torch.manual_seed makes
        a run reproducible) - the reference draws it inside scheduler.step, ladiff.py:492.  `last_noise_seed` holds the seed used."""
        dev = encoder_hidden_states.device
        if not encoder_hidden_states.is_cuda:
            raise _lib.LadiffHipError("_diffusion_reverse needs GPU tensors; there is no CPU fallback")
        # earlier calls' status: the last-but-one call's words are waited for (long there), the previous call's only if they have
        # arrived - the host never blocks on a loop that is still running, and no status is dropped unread
        self._call += 1
        self.check(before=self._call - 1)
        self.check(wait=False)
        n_text = int(encoder_hidden_states.shape[1])          # 1: CLIP pooled token; > 1: clip_hidden / bert (mld_clip.py:80-86)
        cfg = bool(self.do_classifier_free_guidance)        # ladiff.py:339-340, :472-490
        dup = 2 if cfg else 1
        B = encoder_hidden_states.shape[0] // dup
        lengths = [int(l) for l in lengths]
        if encoder_hidden_states.shape[0] != dup * len(lengths):
            raise ValueError(f"{encoder_hidden_states.shape[0]} text rows for {len(lengths)} lengths (guidance: {cfg})")
        counts = self._counts(lengths)
        T = counts[0] if self.test_efficiency else self.max_it     # ladiff.py:381
        self._last = []
        spans = self._chunks(B)
        if step_noise is None and noise_seed is None:
            noise_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.last_noise_seed = None if step_noise is not None else int(noise_seed)
        first = int(self.noise_first_prompt)
        if len(spans) == 1:
            return self._reverse_one(encoder_hidden_states, lengths, counts, T, init_noise, step_noise, noise_seed, first)
        # prompts are independent (attention is per sample, guidance pairs the two branches of one prompt): a large batch is
        # several launches of the loop on contiguous prompt ranges, the noise drawn for the whole batch and sliced
        if init_noise is None:
            init_noise = torch.randn(B, T, 256, device=dev, dtype=torch.float32)       # ladiff.py:380-385
        text = encoder_hidden_states.reshape(dup, B, n_text, 768)
        zs = []
        for lo, hi in spans:
            zs.append(self._reverse_one(text[:, lo:hi].reshape(dup * (hi - lo), n_text, 768), lengths[lo:hi], counts[lo:hi], T,
                                        init_noise[lo:hi], None if step_noise is None else step_noise[:, lo:hi], noise_seed, first + lo))
        return torch.cat(zs, dim=1)

    def _reverse_one(self, encoder_hidden_states, lengths, counts, T, init_noise, step_noise, noise_seed=None, first_prompt=0):
        """One launch sequence of the loop on B prompts: prologue graph, N steps (pipeline kernel or step graphs), final masking."""
        L = _lib.lib()
        dev = encoder_hidden_states.device
        n_text = int(encoder_hidden_states.shape[1])
        cfg = bool(self.do_classifier_free_guidance)
        dup = 2 if cfg else 1
        B = len(lengths)
        sch = self.scheduler
        plan = self._get_plan(B, T, self.num_inference_timesteps, self.eta, dev, n_text)
        self._last.append(plan)
        n, need_noise = plan["n"], plan["need_noise"]
        sampler = plan["sampler"]
        if self._stream is None or self._stream.device != dev:
            self._stream = torch.cuda.Stream(device=dev)
        if init_noise is None:
            init_noise = torch.randn(B, T, 256, device=dev, dtype=torch.float32)       # ladiff.py:380-385
        generated = need_noise and step_noise is None      # drawn inside the loop, keyed by (seed, step, global prompt, latent, column)
        if generated and sampler is None:
            # no sampler handle to carry the seed (use_graph=False): the same values as a tensor (n x B x T x 256 floats)
            step_noise = self.noise_tensor(noise_seed, n, B, T, first_prompt=first_prompt, device=dev)
            generated = False
        if sampler is not None:
            _lib.check(L.ladiff_sampler_set_noise_generator(sampler, int(noise_seed or 0), int(first_prompt), 1 if generated else 0))
        noise_t = None
        if need_noise and not generated:
            if tuple(step_noise.shape) != (n, B, T, 256):
                raise ValueError(f"step_noise {tuple(step_noise.shape)} for a schedule of {n} steps x {B} prompts x {T} latents")
            big = step_noise.numel() * 4 >= (64 << 20)             # a new pointer re-captures the prologue graph (~1 ms): worth it for big tensors only
            if big and step_noise.is_cuda and step_noise.device == dev and step_noise.dtype == torch.float32 and step_noise.is_contiguous():
                noise_t = step_noise                                   # in place: the loop only reads it, on this call's stream
            else:
                if plan["step_noise"] is None:
                    plan["step_noise"] = torch.empty(n, B, T, 256, dtype=torch.float32, device=dev)
                noise_t = plan["step_noise"]
        wt = self.denoiser._weight_table()
        cur = torch.cuda.current_stream(dev)
        # hipStreamBeginCapture is illegal on the null stream: run on the caller's stream when it is a real
        # one, otherwise on a private side stream fenced against it on both sides.
        run = cur if cur.cuda_stream != 0 else self._stream
        if run is not cur:
            run.wait_stream(cur)
        loop_codes = {"pipeline": 1, "pipeline16": 2, "pipeline32": 3, "launches": 0}

        def enqueue(loop):
            if sampler is not None:
                _lib.check(L.ladiff_sampler_set_loop(sampler, loop_codes[loop]))
            _lib.check(L.ladiff_diffusion_reverse(
                sampler, wt.array,
                wt.split_array() if _lib.is_split(self.precision) else None, wt.generation, _lib.ptr(plan["text"]),
                _lib.ptr(plan["noise"]),
                # TEST_EFFICIENCY: no masks inside the denoiser and no zeroing of the initial noise (ladiff.py:381-390,
                # ladiff_denoiser.py:254) - but the final zeroing of ladiff.py:559-566 has no such switch
                None if self.test_efficiency else plan["counts"].data_ptr(), plan["counts"].data_ptr(),
                None if self.test_efficiency else (ctypes.c_int32 * B)(*counts),       # host copy: length-aware block packing
                _lib.ptr(plan["sinus"]), _lib.ptr(plan["coef"]), _lib.ptr(noise_t) if noise_t is not None else None,
                self.guidance_scale, float(sch.init_noise_sigma), 1 if cfg else 0, B, T, n_text, n, _lib.ptr(plan["z"]),
                _lib.ptr(plan["ws"]), plan["ws_bytes"], 1 if plan["tables_key"] == wt.key else 0, run.cuda_stream))
            plan["tables_key"] = wt.key
            # the loop's status words follow it on the stream into pinned memory: 8 bytes, no host synchronisation here
            slot = plan["launches"] % 8
            plan["launches"] += 1
            if len(self._pending) >= 8:
                self.check()                                          # more than eight launches in one call: look at the oldest first
            host, ev = plan["status_host"][slot], plan["status_event"][slot]
            host.copy_(plan["status_dev"], non_blocking=True)
            ev.record(run)
            self._pending.append((ev, host, plan["key"], self._call))

        with torch.cuda.stream(run):
            plan["text"][:dup * B].copy_(encoder_hidden_states.reshape(dup * B, n_text, 768))
            plan["noise"].copy_(init_noise)
            plan["counts"].copy_(_lib.device_ints(counts, dev))       # device-to-device: the graph bakes plan["counts"] in
            if noise_t is not None and noise_t is not step_noise:
                noise_t.copy_(step_noise)
            enqueue(self.loop)
            if self.fallback and self.loop != "launches":
                try:
                    self.check()                                      # waits for THIS call's loop
                except _lib.LadiffHipError:
                    self.fallback_count += 1
                    enqueue("launches")                               # same call, one launch per stage (never the CPU, never a re-exec)
                    self.check()
        if run is not cur:
            cur.wait_stream(run)
        return plan["z"].clone()

    @staticmethod
    def noise_tensor(seed, n_steps, B, T, first_prompt=0, first_step=0, device="cuda:0"):
        """The device generator's values as a tensor [n_steps,B,T,256] (what a call with noise_seed=seed consumes at schedule
        positions first_step .. for global prompts first_prompt ..): for tests and for callers that want to keep the noise."""
        out = torch.empty(n_steps, B, T, 256, dtype=torch.float32, device=device)
        with torch.cuda.device(out.device):
            _lib.check(_lib.lib().ladiff_noise_fill(int(seed), int(first_prompt), int(first_step), int(n_steps), int(B), int(T),
                                                    _lib.ptr(out), torch.cuda.current_stream(out.device).cuda_stream))
        return out

    def loop_ms(self):
        """Device milliseconds of the N-step loop(s) of the last `_diffusion_reverse` call (HIP events on its stream; summed over the
        launches of a chunked batch)."""
        from ctypes import c_float
        total = 0.0
        for plan in self._last:           # a plan that ran k times in the call holds the events of its last run: counted k times
            if plan["sampler"] is None:
                raise _lib.LadiffHipError("loop_ms needs use_graph=True (the events live in the sampler)")
            ms = c_float(0.0)
            _lib.check(_lib.lib().ladiff_sampler_loop_ms(plan["sampler"], byref(ms)))
            total += ms.value
        self.check()                  # the loop has ended: its status is there (an abandoned loop must not be reported as a timing)
        return total

    def window_ms(self, enable=None):
        """(sum of the per-window loop times in ms, windows) of the last call; `enable=True/False` switches the per-window events on
        or off for the following calls (measurement aid: a 1000-step schedule runs as 20 windows of 50 steps)."""
        from ctypes import c_float, c_int
        L = _lib.lib()
        if enable is not None:
            self._window_timing = bool(enable)
            for plan in self._plans.values():
                if plan["sampler"] is not None:
                    _lib.check(L.ladiff_sampler_set_window_timing(plan["sampler"], 1 if enable else 0))
            return None
        ms, n = c_float(0.0), c_int(0)
        _lib.check(L.ladiff_sampler_window_ms(self._sampler, byref(ms), byref(n)))
        return ms.value, n.value

    def last_loop(self):
        """(ran as the persistent pipeline kernel?, rows per block, blocks) of the last `_diffusion_reverse` call."""
        from ctypes import c_int
        if self._sampler is None:
            return False, 0, 0
        pl, rows, nb = c_int(0), c_int(0), c_int(0)
        _lib.check(_lib.lib().ladiff_sampler_last_loop(self._sampler, byref(pl), byref(rows), byref(nb)))
        return bool(pl.value), rows.value, nb.value

    def loop_status(self):
        """(code, info) of the persistent pipeline kernel of the last `_diffusion_reverse` call; blocks until the stream has
        drained.  code 0 = completed; 2 = a stage timed out on its producer (results are then invalid)."""
        from ctypes import c_int
        if self._plan is None:
            return 0, 0
        p = self._plan
        code, info = c_int(0), c_int(0)
        B, T = p["key"][0], p["key"][1]
        _lib.check(_lib.lib().ladiff_reverse_status(_lib.ptr(p["ws"]), B, T, p["n"], p["n_text"], byref(code), byref(info)))
        return code.value, info.value

    # ------------------------------------------------------------------ callers' surface
    def sample(self, text_emb, lengths, init_noise=None, step_noise=None, check=True, noise_seed=None):
        """text embeddings -> (z [max_it,B,256], feats [B,max(len),nfeats]): ladiff.py:266 + :283.  Returns checked frames: the
        decode is queued, then the host waits for the loop's status words (LadiffHipError if the loop was abandoned).
        check=False skips the wait (the host may then queue a call ahead) - the caller owes a `check()` before using the frames."""
        z = self._diffusion_reverse(text_emb, lengths, init_noise=init_noise, step_noise=step_noise, noise_seed=noise_seed)
        feats = self.vae.decode(z, lengths)
        if check:
            self.check()
        return z, feats

    def forward(self, batch, latentwise_gen=None, plot_att_map=None):
        texts, lengths = batch["text"], batch["length"]
        if self.text_encoder is None or (self.feats2joints is None and self.feats2joints_device is None):
            raise RuntimeError("LADIFF.forward needs a text_encoder callable and datamodule.feats2joints; "
                               "use .sample(text_emb, lengths) for embeddings -> features")
        start = time.time()
        text_emb = self.text_encoder(self._guided_texts(texts))
        z = self._diffusion_reverse(text_emb, lengths)
        with torch.no_grad():
            if latentwise_gen:                                      # ladiff.py:274-283
                lengths = list(lengths) * self.max_it
                z = z.repeat(1, self.max_it, 1)
                for idx in range(self.max_it):
                    if latentwise_gen == "fw":
                        z[idx + 1:, idx, :] = 0
                    elif latentwise_gen == "bw":
                        z[:self.max_it - (idx + 1), idx, :] = 0
            feats_rst = self.vae.decode(z, lengths, plot_att_map=plot_att_map, latentwise_gen=latentwise_gen)
        if self.feats2joints_device is not None:
            joints = self.feats2joints_device(feats_rst.detach())
            torch.cuda.synchronize()
            self.check()
            self.times.append(time.time() - start)
            return remove_padding(joints.cpu(), lengths)
        torch.cuda.synchronize()
        self.check()
        self.times.append(time.time() - start)
        joints = self.feats2joints(feats_rst.detach().cpu())
        return remove_padding(joints, lengths)

    def _guided_texts(self, texts):
        """`[""] * B + texts` when classifier-free guidance is on, the texts alone otherwise (ladiff.py:258-264, :1039-1047, :1135-1142)."""
        texts = list(texts)
        return [""] * len(texts) + texts if self.do_classifier_free_guidance else texts

    def test_diffusion_forward(self, batch, finetune_decoder=False):
        """`LADIFF.test_diffusion_forward` (ladiff.py:1035-1109, `condition == 'text'`, a VAE present): text -> latents -> features
        -> joints, and - when the batch carries the ground-truth motion - both motions through the LA-VAE encoder.  Returns the
        reference's `rs_set`: m_rst [B,F,C], lat_t [B,max_it,256], joints_rst, (m_ref, lat_m, lat_rm, joints_ref)."""
        if self.text_encoder is None or (self.feats2joints is None and self.feats2joints_device is None):
            raise RuntimeError("test_diffusion_forward needs a text_encoder callable and datamodule.feats2joints")
        lengths = [int(l) for l in batch["length"]]
        cond_emb = self.text_encoder(self._guided_texts(batch["text"]))                  # :1038-1048
        f2j = self.feats2joints_device or (lambda f: self.feats2joints(f.detach().cpu()))
        with torch.no_grad():
            z = self._diffusion_reverse(cond_emb, lengths)                               # :1060-1061
            feats_rst = self.vae.decode(z, lengths)                                      # :1064-1067
        rs_set = {"m_rst": feats_rst, "lat_t": z.permute(1, 0, 2), "joints_rst": f2j(feats_rst)}      # :1083-1090
        if "motion" in batch and not finetune_decoder:                                   # :1092-1108
            feats_ref = batch["motion"].detach().to(feats_rst.device)
            with torch.no_grad():
                motion_z, _, _ = self.vae.encode(feats_ref, lengths)
                recons_z, _, _ = self.vae.encode(feats_rst, lengths)
            rs_set["m_ref"] = feats_ref
            rs_set["lat_m"] = motion_z.permute(1, 0, 2)
            rs_set["lat_rm"] = recons_z.permute(1, 0, 2)
            rs_set["joints_ref"] = f2j(feats_ref)
        return rs_set

    def set_t2m_evaluators(self, text_encoder, movement_encoder, motion_encoder, unit_len=4):
        """The three frozen evaluator networks of `_get_t2m_evaluator` (ladiff.py:179-223); `unit_len` =
        cfg.DATASET.HUMANML3D.UNIT_LEN (ladiff.py:1259-1261)."""
        self.t2m_textencoder, self.t2m_moveencoder, self.t2m_motionencoder = text_encoder, movement_encoder, motion_encoder
        self.t2m_unit_len = int(unit_len)

    def t2m_eval(self, batch):
        """Text -> motion -> evaluator embeddings for the TM2T metrics (`ladiff.py:1111-1282`, diffusion stage): returns the
        reference's `rs_set` (m_ref, m_rst, lat_t, lat_m, lat_rm, joints_ref, joints_rst), sequences sorted by length."""
        if getattr(self, "t2m_motionencoder", None) is None:
            raise RuntimeError("call set_t2m_evaluators(text_encoder, movement_encoder, motion_encoder) first")
        if self.text_encoder is None or not hasattr(self.datamodule, "renorm4t2m"):
            raise RuntimeError("t2m_eval needs a text_encoder and datamodule.renorm4t2m / feats2joints")
        texts, lengths = list(batch["text"]), [int(l) for l in batch["length"]]
        dev = self.device
        motions = batch["motion"].detach().clone().to(dev)
        start = time.time()
        text_emb = self.text_encoder(self._guided_texts(texts))                       # ladiff.py:1135-1144
        z = self._diffusion_reverse(text_emb, lengths)
        with torch.no_grad():
            feats_rst = self.vae.decode(z, lengths)        # [B, max(lengths), nfeats], frames >= length are zero (:1196-1207)
        torch.cuda.synchronize()
        self.check()
        self.times.append(time.time() - start)
        f2j = self.feats2joints_device or (lambda f: self.feats2joints(f.detach().cpu()))
        joints_rst, joints_ref = f2j(feats_rst), f2j(motions)
        feats_rst = self.datamodule.renorm4t2m(feats_rst)                               # :1244-1246
        motions = self.datamodule.renorm4t2m(motions)
        align = np.argsort(lengths)[::-1].copy()                                         # :1249-1258, longest first
        idx = torch.as_tensor(align, device=dev)
        motions, feats_rst = motions[idx], feats_rst[idx]
        m_lens = torch.tensor(lengths, device=dev)[idx] // self.t2m_unit_len
        recons_emb = self.t2m_motionencoder(self.t2m_moveencoder(feats_rst[..., :-4]), m_lens)
        motion_emb = self.t2m_motionencoder(self.t2m_moveencoder(motions[..., :-4]), m_lens)
        text_lat = self.t2m_textencoder(batch["word_embs"].to(dev), batch["pos_ohot"].to(dev), batch["text_len"])[idx]
        return {"m_ref": motions, "m_rst": feats_rst, "lat_t": text_lat, "lat_m": motion_emb, "lat_rm": recons_emb,
                "joints_ref": joints_ref, "joints_rst": joints_rst}

    def recon_from_motion(self, batch):
        """encode -> decode -> joints of the reconstruction and of the input (ladiff.py:320-331)."""
        feats_ref, length = batch["motion"], batch["length"]
        z, dist, _ = self.vae.encode(feats_ref, length)
        feats_rst = self.vae.decode(z, length)
        f2j = self.feats2joints_device or (lambda f: self.feats2joints(f.detach().cpu()))
        return remove_padding(f2j(feats_rst.detach()).cpu(), length), remove_padding(f2j(feats_ref.detach()).cpu(), length)

    def gen_from_latent(self, batch):
        feats_rst = self.vae.decode(batch["latent"], batch["length"])
        if self.feats2joints_device is not None:
            return remove_padding(self.feats2joints_device(feats_rst.detach()).cpu(), batch["length"])
        joints = self.feats2joints(feats_rst.detach().cpu())
        return remove_padding(joints, batch["length"])
