"""Restatement of diffusers==0.19.3 ``DPMSolverMultistepScheduler`` in the
configuration the reference obtains from the SD-v1.5 ``scheduler_config.json``
(exp-1-debias-gender/1-main-debias.py:738-741): beta_start 0.00085, beta_end 0.012,
scaled_linear, 1000 train steps, epsilon prediction; solver_order 2, dpmsolver++,
midpoint, lower_order_final, timestep_spacing "linspace".
TEST ORACLE -- parity unpinned (diffusers is not vendored/installed; the reference
holds no golden vectors).  Self-validated by tests/test_scheduler.py: exactness on a
linear-Gaussian toy ODE and the scalar-chain identity of SURVEY.md fact 3.

Call sites on the path: ``set_timesteps`` :1038/:1104, ``scale_model_input``
:1044/:1116 (identity), ``step(...).prev_sample`` :1056/:1131, ``alphas_cumprod`` /
``alphas`` :1107.
"""
import numpy as np
import torch


class StepOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class DPMSolverMultistepScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, solver_order=2,
                 lower_order_final=True):
        self.num_train_timesteps = num_train_timesteps
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.alpha_t = torch.sqrt(self.alphas_cumprod)
        self.sigma_t = torch.sqrt(1 - self.alphas_cumprod)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.solver_order = solver_order
        self.lower_order_final = lower_order_final
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.model_outputs = [None] * solver_order
        self.lower_order_nums = 0
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps, device=None):
        ts = np.linspace(0, self.num_train_timesteps - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        _, uniq = np.unique(ts, return_index=True)
        ts = ts[np.sort(uniq)]
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = len(ts)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0

    def scale_model_input(self, sample, *a, **k):
        return sample

    def convert_model_output(self, model_output, timestep, sample):
        a, s = self.alpha_t[timestep], self.sigma_t[timestep]
        return (sample - s * model_output) / a

    def _first_order(self, m0, timestep, prev_timestep, sample):
        lt, ls = self.lambda_t[prev_timestep], self.lambda_t[timestep]
        at = self.alpha_t[prev_timestep]
        st, ss = self.sigma_t[prev_timestep], self.sigma_t[timestep]
        h = lt - ls
        return (st / ss) * sample - (at * (torch.exp(-h) - 1.0)) * m0

    def _second_order(self, outs, tlist, prev_timestep, sample):
        t, s0, s1 = prev_timestep, tlist[-1], tlist[-2]
        m0, m1 = outs[-1], outs[-2]
        lt, ls0, ls1 = self.lambda_t[t], self.lambda_t[s0], self.lambda_t[s1]
        at = self.alpha_t[t]
        st, ss0 = self.sigma_t[t], self.sigma_t[s0]
        h, h0 = lt - ls0, ls0 - ls1
        r0 = h0 / h
        D0, D1 = m0, (1.0 / r0) * (m0 - m1)
        return (st / ss0) * sample - (at * (torch.exp(-h) - 1.0)) * D0 - 0.5 * (at * (torch.exp(-h) - 1.0)) * D1

    def step(self, model_output, timestep, sample):
        timestep = int(timestep)
        idx = (self.timesteps == timestep).nonzero()
        step_index = len(self.timesteps) - 1 if len(idx) == 0 else int(idx[0])
        prev_timestep = 0 if step_index == len(self.timesteps) - 1 else int(self.timesteps[step_index + 1])
        few = len(self.timesteps) < 15
        lower_final = (step_index == len(self.timesteps) - 1) and self.lower_order_final and few
        lower_second = (step_index == len(self.timesteps) - 2) and self.lower_order_final and few
        m = self.convert_model_output(model_output, timestep, sample)
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = m
        if self.solver_order == 1 or self.lower_order_nums < 1 or lower_final:
            prev = self._first_order(m, timestep, prev_timestep, sample)
        else:  # solver_order == 2 (the only higher order on this path)
            tl = [int(self.timesteps[step_index - 1]), timestep]
            prev = self._second_order(self.model_outputs, tl, prev_timestep, sample)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        return StepOutput(prev)
