"""DPM-Solver++ (2M, midpoint) multistep scheduler, API-compatible with the subset of
diffusers==0.19.3 ``DPMSolverMultistepScheduler`` the reference uses
(exp-1-debias-gender/1-main-debias.py:738-741, :1038-1056, :1104-1131): ``set_timesteps``,
``timesteps``, ``scale_model_input``, ``step(...).prev_sample``, ``alphas_cumprod``, ``alphas``.

Per-step scalar coefficients are computed on the host in float32 exactly as the reference's
scheduler does; the tensor update itself runs in the fused HIP kernel ``fd_cfg_dpm_step``
(CFG combine + x0 conversion + multistep update in one pass over the latents).
``chain_coefs`` gives d x_final / d eps_i for the truncated backward.
"""
import numpy as np
import torch

from . import ops


class _StepOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class DPMSolverMultistepScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, solver_order=2, lower_order_final=True):
        self.num_train_timesteps = num_train_timesteps
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.alpha_t = torch.sqrt(self.alphas_cumprod)
        self.sigma_t = torch.sqrt(1 - self.alphas_cumprod)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.solver_order = solver_order
        self.lower_order_final = lower_order_final
        self.set_timesteps(num_train_timesteps)

    # ------------------------------------------------------------------ diffusers-compatible surface
    def set_timesteps(self, num_inference_steps, device=None):
        ts = np.linspace(0, self.num_train_timesteps - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        _, uniq = np.unique(ts, return_index=True)
        ts = ts[np.sort(uniq)]
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = len(ts)
        self._x0_prev = None
        self._step_index = 0

    def scale_model_input(self, sample, *a, **k):
        return sample

    def coefficients(self, i):
        """(alpha_t, sigma_t, c_x, c_d0, c_d1) for step i:  x0 = (x - sigma_t*eps)/alpha_t ;
        x' = c_x*x - c_d0*x0 - c_d1*(x0 - x0_prev)   (c_d1 = 0 on first-order steps)."""
        ts = self.timesteps
        n = len(ts)
        t = int(ts[i])
        prev_t = 0 if i == n - 1 else int(ts[i + 1])
        lower_final = (i == n - 1) and self.lower_order_final and n < 15
        lt, ls = self.lambda_t[prev_t], self.lambda_t[t]
        at = self.alpha_t[prev_t]
        st, ss = self.sigma_t[prev_t], self.sigma_t[t]
        h = lt - ls
        c_x = st / ss
        c_d0 = at * (torch.exp(-h) - 1.0)
        if self.solver_order == 1 or i == 0 or lower_final:
            c_d1 = torch.zeros(())
        else:
            s1 = int(ts[i - 1])
            h0 = ls - self.lambda_t[s1]
            r0 = h0 / h
            c_d1 = 0.5 * c_d0 * (1.0 / r0)
        return float(self.alpha_t[t]), float(self.sigma_t[t]), float(c_x), float(c_d0), float(c_d1)

    def chain_coefs(self):
        """c_i = d x_final / d eps_i through the (linear, scalar-coefficient) multistep recurrence;
        the reference's autograd computes exactly this because the U-Net input is detached (:1115)
        while ``scheduler.step`` keeps the latents in the graph (:1131)."""
        n = len(self.timesteps)
        dlat = np.zeros(n)
        dx0_prev = np.zeros(n)
        for k in range(n):
            a, s, cx, c0, c1 = self.coefficients(k)
            e = np.zeros(n)
            e[k] = 1.0
            dx0 = (dlat - s * e) / a
            dlat = cx * dlat - c0 * dx0 - c1 * (dx0 - dx0_prev)
            dx0_prev = dx0
        return dlat

    def grad_coefs(self):
        """Per-step gradient re-weighting of generate_image_w_gradient (:1105-1109)."""
        c = []
        for t in self.timesteps:
            acp = self.alphas_cumprod[t]
            c.append(acp.sqrt().item() * (1 - acp).sqrt().item() / (1 - self.alphas[t].item()))
        c = np.array(c)
        return c / (float(np.prod(c)) ** (1 / len(c)))

    # ------------------------------------------------------------------ fused device update
    def cfg_step(self, i, eps_2n, guidance_scale, latents, state):
        """In-place latents update for step i from the raw CFG pair eps_2n [2N,4,H,W] fp32.
        ``state`` is a dict owning the multistep history buffers of this rollout."""
        a, s, cx, c0, c1 = self.coefficients(i)
        x0_prev = state.get("x0")
        x0_out = state.get("x0_spare")
        if x0_out is None:
            x0_out = torch.empty_like(latents)
        ops.cfg_dpm_step(eps_2n, float(guidance_scale), latents, x0_prev if c1 != 0.0 else None, x0_out, a, s, cx, c0, c1)
        state["x0"], state["x0_spare"] = x0_out, x0_prev
        return latents

    def step(self, model_output, timestep, sample):
        """diffusers-style ``step`` on device tensors (model_output already CFG-combined)."""
        idx = (self.timesteps == int(timestep)).nonzero()
        i = len(self.timesteps) - 1 if len(idx) == 0 else int(idx[0])
        if i == 0:
            self._state = {}
        lat = sample.detach().to(torch.float32).clone()
        e = model_output.detach().to(torch.float32).contiguous()
        self.cfg_step(i, torch.cat([e, e]), 0.0, lat, self._state)
        return _StepOutput(lat)
