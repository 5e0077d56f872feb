"""Oracle (TEST INFRASTRUCTURE ONLY): the D-step / G-step arithmetic of the reference, on torch-CPU
fp32 autograd, with every random draw injected by the caller.

Reference being restated (all under /root/reference/src/dp_cgans/synthesizers/):
  Discriminator / calc_gradient_penalty   dp_cgan.py:31-68
  Residual / Generator                    dp_cgan.py:71-100
  _apply_activate (+ F.gumbel_softmax)    dp_cgan.py:280-327
  _cond_loss_pair                         dp_cgan.py:330-389
  one discriminator update                dp_cgan.py:577-661
  one generator update                    dp_cgan.py:673-722
  Adam(betas .5/.9, eps 1e-8, L2 decay)   dp_cgan.py:510-518 (torch.optim.Adam semantics)

``span_info`` everywhere is an int array [n_spans, 3] = (dim, is_softmax, spans_in_column), the
flattened ``DataTransformer.output_info_list`` (data_transformer.py:53-71).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

TAU = 0.2          # dp_cgan.py:321
LEAK = 0.2         # dp_cgan.py:40
GP_LAMBDA = 10.0   # dp_cgan.py:46


# ----------------------------------------------------------------------------------------------
# span tables
# ----------------------------------------------------------------------------------------------
def spans(span_info):
    """[(data_offset, dim, is_softmax)] in generator-output order."""
    out, st = [], 0
    for dim, sm, _ in np.asarray(span_info).tolist():
        out.append((st, dim, bool(sm)))
        st += dim
    return out


def discrete_columns(span_info):
    """[(data_offset, cond_offset, dim)] for every column that is ONE softmax span
    (is_discrete_column, data_sampler.py:12-14 / dp_cgan.py:344)."""
    out, st, cst = [], 0, 0
    for dim, sm, ncol in np.asarray(span_info).tolist():
        if sm and ncol == 1:
            out.append((st, cst, dim))
            cst += dim
        st += dim
    return out


# ----------------------------------------------------------------------------------------------
# activations and conditional loss
# ----------------------------------------------------------------------------------------------
def apply_activate(fake, span_info, exp_noise):
    """tanh on value spans; softmax((x - log E)/tau) on the others, E ~ Exp(1) injected
    (``exp_noise`` is [B, data_dim]; tanh columns are ignored)."""
    parts = []
    for st, dim, sm in spans(span_info):
        x = fake[:, st:st + dim]
        if sm:
            g = -torch.log(exp_noise[:, st:st + dim])
            parts.append(torch.softmax((x + g) / TAU, dim=-1))
        else:
            parts.append(torch.tanh(x))
    return torch.cat(parts, dim=1)


def cond_loss_pair_loop(fake, cond, mask, span_info):
    """The reference's per-pair formulation: a [B * n_pairs, n_disc] buffer, two columns of it
    filled per pair with mask-weighted BCE-with-logits sums, then sum / rows."""
    cols = discrete_columns(span_info)
    nd, B = len(cols), fake.shape[0]
    n_pairs = nd * (nd - 1) // 2
    buf = torch.zeros(B * n_pairs, nd)
    maskf = mask.to(fake.dtype)
    p = 0
    for i in range(nd):
        di, ci, ki = cols[i]
        for j in range(i + 1, nd):
            dj, cj, kj = cols[j]
            li = F.binary_cross_entropy_with_logits(fake[:, di:di + ki], cond[:, ci:ci + ki], reduction='none')
            lj = F.binary_cross_entropy_with_logits(fake[:, dj:dj + kj], cond[:, cj:cj + kj], reduction='none')
            buf[p * B:(p + 1) * B, i] = li.sum(1) * maskf[:, i]
            buf[p * B:(p + 1) * B, j] = lj.sum(1) * maskf[:, j]
            p += 1
    return buf.sum() / buf.shape[0]


def cond_loss_closed(fake, cond, mask, span_info):
    """Closed form of the loop above: every column d appears in (nd-1) pairs, so
    loss = (nd-1) * sum_b sum_d m[b,d] * BCE_d[b] / (B * n_pairs) = 2/(B*nd) * sum(...)."""
    cols = discrete_columns(span_info)
    nd, B = len(cols), fake.shape[0]
    tot = fake.new_zeros(())
    for d, (ds, cs, k) in enumerate(cols):
        x, c = fake[:, ds:ds + k], cond[:, cs:cs + k]
        bce = torch.clamp(x, min=0) - x * c + torch.log1p(torch.exp(-x.abs()))
        tot = tot + (bce.sum(1) * mask[:, d].to(fake.dtype)).sum()
    return tot * (2.0 / (B * nd))


# ----------------------------------------------------------------------------------------------
# networks (functional, parameters in dicts keyed like the reference state_dict)
# ----------------------------------------------------------------------------------------------
def n_residual(gp):
    return sum(1 for k in gp if k.endswith('.fc.weight'))


def gen_forward(gp, x, bn_state=None):
    """Generator forward, BatchNorm in training mode (the reference never calls .eval(), Q2).
    ``bn_state`` (dict of running_mean/var/num_batches_tracked) is updated in place if given."""
    L = n_residual(gp)
    h = x
    for l in range(L):
        o = F.linear(h, gp['seq.%d.fc.weight' % l], gp['seq.%d.fc.bias' % l])
        rm = rv = None
        if bn_state is not None:
            rm, rv = bn_state['seq.%d.bn.running_mean' % l], bn_state['seq.%d.bn.running_var' % l]
            bn_state['seq.%d.bn.num_batches_tracked' % l] += 1
        o = F.batch_norm(o, rm, rv, gp['seq.%d.bn.weight' % l], gp['seq.%d.bn.bias' % l],
                         training=True, momentum=0.1, eps=1e-5)
        h = torch.cat([torch.relu(o), h], dim=1)
    return F.linear(h, gp['seq.%d.weight' % L], gp['seq.%d.bias' % L])


def disc_layers(dp):
    return sorted(int(k.split('.')[1]) for k in dp if k.endswith('.weight'))


def disc_forward(dp, x, pac, masks):
    """Discriminator forward on pac-grouped rows; ``masks`` = one {0,1} keep-mask per hidden layer
    (Dropout(0.5) in training mode: kept activations are scaled by 2)."""
    idx = disc_layers(dp)
    h = x.reshape(-1, pac * x.shape[1])
    for li, mk in zip(idx[:-1], masks):
        h = F.leaky_relu(F.linear(h, dp['seq.%d.weight' % li], dp['seq.%d.bias' % li]), LEAK)
        h = h * mk.to(h.dtype) * 2.0
    return F.linear(h, dp['seq.%d.weight' % idx[-1]], dp['seq.%d.bias' % idx[-1]])


def gradient_penalty(dp, real_cat, fake_cat, pac, alpha, masks):
    """10 * mean((||d D(xhat)/d xhat||_2 - 1)^2), xhat = a*real + (1-a)*fake with one a per pac
    group; the norm runs over the whole pac row.  Differentiable (double backward)."""
    n = real_cat.shape[0] // pac
    a = alpha.reshape(n, 1, 1).repeat(1, pac, real_cat.shape[1]).reshape(-1, real_cat.shape[1])
    xhat = a * real_cat + (1 - a) * fake_cat
    y = disc_forward(dp, xhat, pac, masks)
    (g,) = torch.autograd.grad(y, xhat, torch.ones_like(y), create_graph=True, retain_graph=True)
    nrm = g.reshape(-1, pac * real_cat.shape[1]).norm(2, dim=1)
    return ((nrm - 1) ** 2).mean() * GP_LAMBDA


def gradient_penalty_analytic(dp, pac, masks):
    """Closed form of the penalty and of its weight gradients.  With the activation
    slopes s_i = mask_i*2*{1 or 0.2} frozen, D is linear in xhat, so the input gradient is a
    product of the weights and slopes and does not depend on xhat at all; the slopes do (through
    the sign of the pre-activations), which the caller supplies via ``masks`` = full slope arrays.
    Returns (pen, {weight key: grad}).  Used to check the HIP kernels' formulation."""
    idx = disc_layers(dp)
    Ws = [dp['seq.%d.weight' % i] for i in idx]
    n = masks[0].shape[0]
    # g-chain (downwards): g_M = w_last, u_i = g_i * s_i, g_{i-1} = u_i W_i
    g = Ws[-1].expand(n, -1)
    us = []
    for W, s in zip(reversed(Ws[:-1]), reversed(masks)):
        u = g * s
        us.append(u)
        g = u @ W
    us = us[::-1]
    nrm = g.norm(2, dim=1)
    pen = ((nrm - 1) ** 2).mean() * GP_LAMBDA
    # v-chain (upwards): v_0 = dpen/dg_0
    v = (GP_LAMBDA * 2.0 / n) * ((nrm - 1) / nrm).unsqueeze(1) * g
    grads = {}
    for li, W, u, s in zip(idx[:-1], Ws[:-1], us, masks):
        grads['seq.%d.weight' % li] = u.t() @ v
        v = (v @ W.t()) * s
    grads['seq.%d.weight' % idx[-1]] = v.sum(0, keepdim=True)
    return pen, grads


# ----------------------------------------------------------------------------------------------
# optimiser
# ----------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, t, lr=2e-4, wd=1e-6, b1=0.5, b2=0.9, eps=1e-8):
    """torch.optim.Adam single-tensor update with coupled L2 decay; t is the 1-based
    step count.  Returns (p, m, v) new tensors."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# ----------------------------------------------------------------------------------------------
# one discriminator update and one generator update
# ----------------------------------------------------------------------------------------------
def d_step_grads(gp, dp, span_info, pac, z, cond, perm, real, exp_noise, masks_fake, masks_real,
                 masks_gp, alpha, bn_state=None):
    """Forward + both backward passes of one discriminator update, no optimiser.
    Returns dict(fake, fakeact, y_fake, y_real, loss_d, pen, grads{key: tensor})."""
    dpl = {k: v.detach().clone().requires_grad_(True) for k, v in dp.items()}
    gpl = {k: v.detach().clone().requires_grad_(True) for k, v in gp.items()}
    fake = gen_forward(gpl, torch.cat([z, cond], 1), bn_state)
    fakeact = apply_activate(fake, span_info, exp_noise)
    if cond.shape[1] > 0:
        fake_cat = torch.cat([fakeact, cond], 1)
        real_cat = torch.cat([real, cond[perm]], 1)
    else:               # no discrete columns: the RAW generator output goes to D (dp_cgan.py:621-626, Q5)
        fake_cat, real_cat = fake, real
    y_fake = disc_forward(dpl, fake_cat, pac, masks_fake)
    y_real = disc_forward(dpl, real_cat, pac, masks_real)
    loss_d = -(y_real.mean() - y_fake.mean())
    pen = gradient_penalty(dpl, real_cat, fake_cat, pac, alpha, masks_gp)
    keys = list(dpl.keys())
    g_pen = torch.autograd.grad(pen, [dpl[k] for k in keys], retain_graph=True, allow_unused=True)
    g_loss = torch.autograd.grad(loss_d, [dpl[k] for k in keys], allow_unused=True)
    grads = {}
    for k, a, b in zip(keys, g_pen, g_loss):
        t = torch.zeros_like(dpl[k])
        if a is not None:
            t = t + a
        if b is not None:
            t = t + b
        grads[k] = t.detach()
    return dict(fake=fake.detach(), fakeact=fakeact.detach(), y_fake=y_fake.detach(),
                y_real=y_real.detach(), loss_d=loss_d.detach(), pen=pen.detach(), grads=grads)


def g_step_grads(gp, dp, span_info, pac, z, cond, mask, exp_noise, masks_d, bn_state=None,
                 loop_loss=True):
    """Forward + backward of one generator update, no optimiser."""
    gpl = {k: v.detach().clone().requires_grad_(True) for k, v in gp.items()}
    fake = gen_forward(gpl, torch.cat([z, cond], 1), bn_state)
    fakeact = apply_activate(fake, span_info, exp_noise)
    y_fake = disc_forward(dp, torch.cat([fakeact, cond], 1), pac, masks_d)
    if cond.shape[1] > 0:
        ce = (cond_loss_pair_loop if loop_loss else cond_loss_closed)(fake, cond, mask, span_info)
    else:               # condvec_pair is None: cross_entropy_pair = 0 (dp_cgan.py:708-709)
        ce = torch.zeros(())
    loss_g = -y_fake.mean() + ce
    keys = list(gpl.keys())
    gs = torch.autograd.grad(loss_g, [gpl[k] for k in keys])
    return dict(fake=fake.detach(), fakeact=fakeact.detach(), y_fake=y_fake.detach(),
                ce=ce.detach(), loss_g=loss_g.detach(),
                grads={k: g.detach() for k, g in zip(keys, gs)})


def apply_update(params, grads, state, lr=2e-4, wd=1e-6, clamp=None):
    """Adam on every tensor of ``params`` (state = {'t': int, 'm': {}, 'v': {}}), then the
    reference's weight clamp when ``clamp`` is given (dp_cgan.py:657-661)."""
    state['t'] += 1
    out = {}
    for k, p in params.items():
        m = state['m'].get(k, torch.zeros_like(p))
        v = state['v'].get(k, torch.zeros_like(p))
        p2, m2, v2 = adam_step(p, grads[k], m, v, state['t'], lr, wd)
        if clamp is not None:
            p2 = p2.clamp(-clamp, clamp)
        out[k], state['m'][k], state['v'][k] = p2, m2, v2
    return out
