"""DDIM / DDPM schedulers with the duck-typed surface `LADIFF` uses.

The reference instantiates `diffusers.DDIMScheduler` / `diffusers.DDPMScheduler` from YAML
(`src/configs/modules/scheduler.yaml:1-14`, `modules_novae/scheduler.yaml:16-29`) and touches only:
`.init_noise_sigma` (ladiff.py:407), `.set_timesteps(n)` (:410), `.timesteps` (:411),
`.step(eps, t, x, eta=...)` -> `.prev_sample` (:491-492; `eta` is probed with inspect.signature, :415-417),
`.config.num_train_timesteps` (:770) and `.add_noise` (:776, training).

diffusers is not vendored, not pinned and not installed here, so these classes RESTATE its published
formulas for the options the reference sets (scaled_linear betas, clip_sample false, set_alpha_to_one
false, steps_offset 1, fixed_small variance).  The host part only builds per-step scalar tables; the
update itself (guidance + step on [B,T,256]) runs in the HIP kernel `ladiff_cfg_scheduler_step`.
"""
import math
from types import SimpleNamespace

import torch

from . import _lib


class SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class _Scheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, prediction_type="epsilon", **kwargs):
        # the default is diffusers' (True), so that a YAML which omits the key fails loudly instead of silently
        # sampling without the clipping diffusers would apply; the reference sets `clip_sample: false` (scheduler.yaml)
        if clip_sample:
            raise NotImplementedError("clip_sample=True is not built (the reference sets clip_sample: false; "
                                      "pass clip_sample=False explicitly)")
        if prediction_type != "epsilon":
            raise NotImplementedError("only prediction_type='epsilon' is built (base.yaml:27 PREDICT_EPSILON: True)")
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule!r} is not built")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.init_noise_sigma = 1.0
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule, clip_sample=clip_sample,
                                      prediction_type=prediction_type, **kwargs)
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)
        self._coef_dev = {}
        self._step_dev = {}

    # ---- table of per-step scalars consumed by the HIP kernel, one row of 8 floats per step
    def _row(self, t, eta):
        raise NotImplementedError

    def coef_table(self, eta=0.0):
        rows = [self._row(int(t), float(eta)) for t in self.timesteps]
        tab = torch.zeros(len(rows), _lib.COEF_STRIDE, dtype=torch.float32)
        for i, r in enumerate(rows):
            tab[i, :len(r)] = torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in r])
        return tab

    def needs_noise(self, eta=0.0):
        return bool((self.coef_table(eta)[:, 5] != 0).any())

    # ---- single step on the GPU (drop-in for `scheduler.step(...).prev_sample`, ladiff.py:491-492)
    def _step(self, model_output, timestep, sample, eta, variance_noise):
        t = int(timestep)
        idx = (self.timesteps == t).nonzero()
        if idx.numel() == 0:
            raise ValueError(f"timestep {t} is not in the current schedule (call set_timesteps first)")
        key = (sample.device, float(eta), int(self.num_inference_steps or -1))
        if key not in self._coef_dev:
            self._coef_dev = {key: self.coef_table(eta).to(sample.device)}
            self._step_dev = {key: torch.arange(len(self.timesteps), dtype=torch.int32, device=sample.device)}
        coef, steps = self._coef_dev[key], self._step_dev[key]
        i = int(idx[0])
        if coef[i, 5].item() != 0 and variance_noise is None:
            variance_noise = torch.randn_like(sample)
        out = sample.detach().to(torch.float32).contiguous().clone()
        eps = model_output.detach().to(torch.float32).contiguous()
        if out.numel() % 256 or eps.shape != out.shape:
            raise ValueError(f"scheduler.step works on [..., 256] latents; got sample {tuple(sample.shape)}, "
                             f"model_output {tuple(model_output.shape)}")
        n = out.numel() // 256
        noise_ptr = None
        if variance_noise is not None:
            # the kernel indexes noise by step: hand it a view whose row `i` is this step's noise
            vz = variance_noise.detach().to(torch.float32).contiguous()
            noise_ptr = vz.data_ptr() - i * vz.numel() * 4
        _lib.check(_lib.lib().ladiff_cfg_scheduler_step(_lib.ptr(eps), _lib.ptr(out), _lib.ptr(coef),
                                                        steps[i:].data_ptr(), noise_ptr, 1.0, 0, n, 1,
                                                        _lib.stream_ptr()))
        return SchedulerOutput(out.to(sample.dtype))

    def add_noise(self, original_samples, noise, timesteps):
        """x_t = sqrt(a_t) x_0 + sqrt(1 - a_t) eps  (training side, ladiff.py:776; plain torch, not on the hot path)."""
        a = self.alphas_cumprod.to(original_samples.device)[timesteps.to(original_samples.device)]
        a = a.to(original_samples.dtype)
        while a.dim() < original_samples.dim():
            a = a.unsqueeze(-1)
        return a ** 0.5 * original_samples + (1 - a) ** 0.5 * noise

    def __len__(self):
        return self.config.num_train_timesteps


class DDIMScheduler(_Scheduler):
    def __init__(self, set_alpha_to_one=True, steps_offset=0, **kwargs):
        super().__init__(set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, **kwargs)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def set_timesteps(self, num_inference_steps, device=None):
        n = int(num_inference_steps)
        self.num_inference_steps = n
        ratio = self.config.num_train_timesteps // n
        self.timesteps = torch.arange(n - 1, -1, -1, dtype=torch.int64) * ratio + self.config.steps_offset
        self._coef_dev, self._step_dev = {}, {}

    def _row(self, t, eta):
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        var = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
        sigma = eta * var ** 0.5
        return [a_t ** 0.5, (1 - a_t) ** 0.5, a_p ** 0.5, 0.0, (1 - a_p - sigma ** 2) ** 0.5, sigma]

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("call set_timesteps before step")
        out = self._step(model_output, timestep, sample, eta, variance_noise)
        return out if return_dict else (out.prev_sample,)


class DDPMScheduler(_Scheduler):
    """`prev_timestep`: "t-1" follows diffusers <= 0.14, the reference's era (alpha_prod_t_prev = alphas_cumprod[t-1],
    beta_t = betas[t]); "schedule" follows later releases (prev_t = t - num_train // num_inference,
    beta_t = 1 - a_t / a_prev).  They coincide for num_inference_steps == num_train_timesteps, which is how the
    reference runs DDPM (modules_novae/scheduler.yaml:16-29); the dependency is not version-pinned
    (src/requirements.txt:23), hence the switch."""

    def __init__(self, variance_type="fixed_small", prev_timestep="t-1", **kwargs):
        if variance_type != "fixed_small":
            raise NotImplementedError("only variance_type='fixed_small' is built (scheduler.yaml)")
        if prev_timestep not in ("t-1", "schedule"):
            raise ValueError(f"prev_timestep {prev_timestep!r}")
        self.prev_timestep = prev_timestep
        super().__init__(variance_type=variance_type, **kwargs)

    def set_timesteps(self, num_inference_steps, device=None):
        n = min(self.config.num_train_timesteps, int(num_inference_steps))
        self.num_inference_steps = n
        self.timesteps = torch.arange(0, self.config.num_train_timesteps, self.config.num_train_timesteps // n,
                                      dtype=torch.int64).flip(0)
        self._coef_dev, self._step_dev = {}, {}

    def _row(self, t, eta):
        a_t = self.alphas_cumprod[t]
        if self.prev_timestep == "schedule":
            prev = t - self.config.num_train_timesteps // self.num_inference_steps
            a_p = self.alphas_cumprod[prev] if prev >= 0 else torch.tensor(1.0)
            b_t = 1 - a_t / a_p
        else:
            a_p = self.alphas_cumprod[t - 1] if t > 0 else torch.tensor(1.0)
            b_t = self.betas[t]
        k_x0 = a_p ** 0.5 * b_t / (1 - a_t)
        k_x = (1 - b_t) ** 0.5 * (1 - a_p) / (1 - a_t)
        k_n = torch.clamp((1 - a_p) / (1 - a_t) * b_t, min=1e-20) ** 0.5 if t > 0 else 0.0
        return [a_t ** 0.5, (1 - a_t) ** 0.5, k_x0, k_x, 0.0, k_n]

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict=True):
        if self.num_inference_steps is None:
            self.set_timesteps(self.config.num_train_timesteps)
        out = self._step(model_output, timestep, sample, 0.0, variance_noise)
        return out if return_dict else (out.prev_sample,)


def timestep_sinusoid(timesteps, dim=768):
    """Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) on the host (tools/embeddings.py:245-285), fp32 op for op.

    A t-only table like the scheduler coefficients: [n, dim] = [cos(t f) | sin(t f)], f_i = exp(-ln(1e4) i / (dim/2)).
    """
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32)
    exponent = exponent / (half - 0)
    arg = torch.as_tensor(timesteps).reshape(-1, 1).float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1).contiguous()
