"""Oracle (TEST INFRASTRUCTURE ONLY) for the DP-SGD mode of the discriminator update.

The reference has no per-example clipping (its private=True is grad-hook noise + a weight clamp,
dp_cgan.py:638-661), so this mode has no reference counterpart: parity for it is
pinned to this per-example autograd loop, NOT to the reference ("unpinned by the reference").
What anchors the loop itself to the reference: with an infinite clip bound the clipped mean below
is exactly the gradient of  mean(y_fake) - mean(y_real) + pen  that the reference back-propagates
(dp_cgan.py:629-655), which tests/test_oracle_golden.py checks against the captured trace.

An example is one pac group i; its loss is
    l_i = D(fake_i) - D(real_i) + lambda * (||d D(xhat_i) / d xhat_i||_2 - 1)^2
and the update uses  (1/n) * [ sum_i g_i * min(1, C/||g_i||) + sigma*C*N(0, I) ].
"""
import torch

from . import nets


def per_example_grads(dp, pac, fake_cat, real_cat, masks_fake, masks_real, masks_gp, alpha):
    """-> (keys, [n][len(keys)] gradient tensors of l_i w.r.t. the discriminator parameters)."""
    n = fake_cat.shape[0] // pac
    keys = list(dp.keys())
    out = []
    for i in range(n):
        rows = slice(i * pac, (i + 1) * pac)
        dpl = {k: v.detach().clone().requires_grad_(True) for k, v in dp.items()}
        yf = nets.disc_forward(dpl, fake_cat[rows], pac, [m[i:i + 1] for m in masks_fake])
        yr = nets.disc_forward(dpl, real_cat[rows], pac, [m[i:i + 1] for m in masks_real])
        a = alpha[i]
        xhat = (a * real_cat[rows] + (1 - a) * fake_cat[rows]).detach().requires_grad_(True)
        y = nets.disc_forward(dpl, xhat, pac, [m[i:i + 1] for m in masks_gp])
        (gx,) = torch.autograd.grad(y, xhat, torch.ones_like(y), create_graph=True)
        pen = (gx.reshape(1, -1).norm(2, dim=1) - 1) ** 2 * nets.GP_LAMBDA
        loss = yf.sum() - yr.sum() + pen.sum()
        gs = torch.autograd.grad(loss, [dpl[k] for k in keys], allow_unused=True)
        out.append([torch.zeros_like(dpl[k]) if g is None else g.detach() for k, g in zip(keys, gs)])
    return keys, out


def d_step_grads_dpsgd(gp, dp, span_info, pac, z, cond, perm, real, exp_noise, masks_fake, masks_real, masks_gp,
                       alpha, clip, n_total=None):
    """Clipped mean gradient of one discriminator update (no noise, no optimiser).
    Returns dict(norms [n], factors [n], grads {key: tensor}, loss_d, pen)."""
    fake = nets.gen_forward(gp, torch.cat([z, cond], 1))
    fakeact = nets.apply_activate(fake, span_info, exp_noise)
    fake_cat = torch.cat([fakeact, cond], 1).detach()
    real_cat = torch.cat([real, cond[perm]], 1)
    keys, per = per_example_grads(dp, pac, fake_cat, real_cat, masks_fake, masks_real, masks_gp, alpha)
    n = len(per)
    n_total = n_total or n
    norms = torch.stack([torch.sqrt(sum((g.double() ** 2).sum() for g in gs)) for gs in per])
    factors = torch.clamp(clip / norms, max=1.0)
    grads = {}
    for j, k in enumerate(keys):
        acc = torch.zeros_like(per[0][j], dtype=torch.float64)
        for i in range(n):
            acc += factors[i] * per[i][j].double()
        grads[k] = (acc / n_total).float()
    return dict(norms=norms.float(), factors=factors.float(), grads=grads, fake=fake.detach(), fakeact=fakeact.detach())


def add_noise(grads, noise, sigma, clip, n_total):
    """grads + sigma*clip/n * noise (noise: {key: N(0,1) tensor}, injected)."""
    return {k: g + (sigma * clip / n_total) * noise[k] for k, g in grads.items()}
