"""Diffusion schedule tables and sampler step sequences (host side, numpy/float64 -> fp32).

Mirrors the pieces of the reference runner that sit on the hot path's boundary:
``get_beta_schedule`` (reference ``runners/diffusion.py:32-62``), the fp32 cumulative product of
``Diffusion.__init__`` (``:90-128``) and the ``seq`` construction of ``sample_image`` (``:475-500``).
"""
import numpy as np
import torch


def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps):
    n = num_diffusion_timesteps
    if beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, n, dtype=np.float64)
    elif beta_schedule == "quad":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float64) ** 2
    elif beta_schedule == "const":
        betas = beta_end * np.ones(n, dtype=np.float64)
    elif beta_schedule == "jsd":
        betas = 1.0 / np.linspace(n, 1, n, dtype=np.float64)
    elif beta_schedule == "sigmoid":
        x = np.linspace(-6, 6, n)
        betas = 1.0 / (np.exp(-x) + 1.0) * (beta_end - beta_start) + beta_start
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (n,)
    return betas


def alphas_cumprod(betas):
    """fp32 cumprod of [1, 1-beta] without the leading 1: the ``self.alphas`` table the reference
    runner hands to ``generalized_steps`` (``runners/diffusion.py:109-115,497-499``).  The product is
    taken in fp32, sequentially, exactly as ``torch.cumprod`` on a float tensor does."""
    a = torch.from_numpy(np.concatenate([[1.0], 1.0 - np.asarray(betas, dtype=np.float64)])).to(torch.float32)
    return a.cumprod(dim=0)[1:].contiguous()


def make_schedule(diffusion_cfg):
    betas = get_beta_schedule(
        diffusion_cfg.beta_schedule,
        beta_start=diffusion_cfg.beta_start,
        beta_end=diffusion_cfg.beta_end,
        num_diffusion_timesteps=diffusion_cfg.num_diffusion_timesteps,
    )
    return torch.from_numpy(betas).to(torch.float32), alphas_cumprod(betas)


def make_seq(num_timesteps, timesteps, skip_type="uniform"):
    """Timestep subsequence of ``sample_image`` (reference ``runners/diffusion.py:482-494``)."""
    if skip_type == "uniform":
        skip = num_timesteps // timesteps
        return list(range(0, num_timesteps, skip))
    if skip_type == "quad":
        seq = np.linspace(0, np.sqrt(num_timesteps * 0.8), timesteps) ** 2
        return [int(s) for s in list(seq)]
    raise NotImplementedError(skip_type)


def ddim_coefficients(seq, alpha, eta=0.0):
    """Per-iteration scalars of ``generalized_steps`` (reference ``functions/denoising.py:12-40``).

    ``alpha`` is the fp32 alphas-cumprod table; like the reference, the scalars are formed in
    Python double precision from the fp32 table values.  Returns a float64 array [n_iter, 6] in
    execution order (reversed ``seq``): columns (t, sqrt(1-at), sqrt(at), sqrt(at_next), c2, c1).
    """
    a = [1.0] + torch.as_tensor(alpha).to("cpu", torch.float32).numpy().tolist()
    seq = list(seq)
    seq_next = [-1] + seq[:-1]
    rows = []
    for i, j in zip(reversed(seq), reversed(seq_next)):
        at = a[int(i) + 1]
        at_next = a[int(j) + 1]
        c1 = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)) ** 0.5
        c2 = ((1 - at_next) - c1 ** 2) ** 0.5
        rows.append((float(int(i)), (1 - at) ** 0.5, at ** 0.5, at_next ** 0.5, c2, c1))
    return np.asarray(rows, dtype=np.float64).reshape(-1, 6)


def ddpm_coefficients(seq, betas):
    """Per-iteration scalars of ``ddpm_steps`` (reference ``functions/denoising.py:4-7,64-88``), formed with the
    reference's fp32 *tensor* arithmetic (it works on [n,1,1,1] fp32 tensors, not Python doubles).  Returns a float32
    array [n_iter, 7] in execution order: (t, (1/at).sqrt(), (1/at-1).sqrt(), atm1.sqrt()*beta_t,
    (1-beta_t).sqrt()*(1-atm1), 1-at, mask*exp(0.5*log(beta_t)))."""
    b = torch.as_tensor(betas).to("cpu", torch.float32)
    acp = (1 - torch.cat([torch.zeros(1), b], dim=0)).cumprod(dim=0)  # compute_alpha's table, index t+1
    seq = list(seq)
    seq_next = [-1] + seq[:-1]
    rows = []
    for i, j in zip(reversed(seq), reversed(seq_next)):
        at, atm1 = acp[int(i) + 1], acp[int(j) + 1]
        beta_t = 1 - at / atm1
        mask = 1.0 - float(int(i) == 0)
        rows.append(torch.stack([torch.tensor(float(int(i))), (1.0 / at).sqrt(), (1.0 / at - 1).sqrt(), atm1.sqrt() * beta_t,
                                 (1 - beta_t).sqrt() * (1 - atm1), 1.0 - at, mask * torch.exp(0.5 * beta_t.log())]))
    return torch.stack(rows).to(torch.float32).numpy()
