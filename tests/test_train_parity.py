"""one Network.train step (blackbird_amd/training.py, PyTorch-ROCm float32, NCHW conv2d) against an
INDEPENDENT statement of the reference's training graph (/root/reference/src/NetworkFactory.py:37-245) written here in
float64 as shifted-window einsum contractions in NHWC with a hand-written optimiser in numpy:

  * the three loss terms (value MSE :187-188, the BxB policy cross term :190-194, the unweighted L2 mean over non-bias
    trainables :196-201), with the graph's shared Beta noise and batch-wide normalisation (:176-182) inside;
  * d loss / d variable for EVERY trainable variable;
  * the variables after one and two optimiser steps for adam / momentum / sgd (TF1 update rules, :234-242).

Tolerance 1e-5 (BASELINE.json north_star) relative to the largest entry of each tensor.  TensorFlow itself is not
installable here, so this pins the implementation to the published formulas, not to TF's binaries ("parity unpinned by
the reference").  The CPU variant runs in the `not gpu` suite; the `gpu` variant runs the same step on the
MI355X and also checks that the engine's logits after the weight reload are those of the exported weights."""
import numpy as np
import pytest
import torch

from blackbird_amd import weights as W
from blackbird_amd.training import Trainer

TOL = 1e-5
H, Wd, C, F, R, D, A, B = 6, 7, 3, 16, 2, 16, 7, 6
ALPHA, EPS = 0.2, 0.3


def _batch(seed):
    rng = np.random.RandomState(seed)
    cells = rng.randint(0, 3, size=(B, H, Wd))
    boards = np.zeros((B, H, Wd, C), dtype=np.int8)
    boards[..., 0] = cells == 1
    boards[..., 1] = cells == 2
    boards[..., 2] = rng.choice([-1, 1], size=(B, 1, 1))
    ev = rng.choice([-1.0, 0.0, 1.0], size=B).astype(np.float32)
    pl = rng.dirichlet(np.ones(A), size=B)
    pl[-1] = 0.0  # a terminal example: pi = zeros (Blackbird.py:256-258)
    noise = rng.beta(ALPHA, 1 - ALPHA, size=A)
    return boards, ev, pl, noise


# ---- the independent statement (float64, NHWC, no conv2d) ---------------------------------------------------------
def _ref_loss(w, boards, ev, pl, noise):
    x = torch.tensor(boards.astype(np.float64))
    ev = torch.tensor(ev.astype(np.float64))
    pl = torch.tensor(pl.astype(np.float64))
    nz = torch.tensor(noise.astype(np.float64))

    def conv(t, scope):  # tf.layers.conv2d, SAME, stride 1, bias
        k, b = w[scope + "/kernel"], w[scope + "/bias"]
        kh = k.shape[0]
        p = kh // 2
        tp = torch.nn.functional.pad(t, (0, 0, p, p, p, p))
        out = 0
        for dy in range(kh):
            for dx in range(kh):
                out = out + torch.einsum("bhwc,cf->bhwf", tp[:, dy:dy + H, dx:dx + Wd, :], k[dy, dx])
        return out + b

    def bn(t, scope):  # batch_normalization, training=False, epsilon 1e-3
        g, be, mu, var = (w[f"{scope}/{f}"] for f in W.BN_FIELDS)
        return g * (t - mu) / torch.sqrt(var + 1e-3) + be

    t = torch.relu(bn(conv(x, "resTower/conv_block/conv"), "resTower/conv_block/batch_norm"))
    for i in range(R):
        h = torch.relu(bn(conv(t, f"resTower/block_{i}/conv_1"), f"resTower/block_{i}/batch_norm_1"))
        h = bn(conv(h, f"resTower/block_{i}/conv_2"), f"resTower/block_{i}/batch_norm_2")
        t = torch.relu(h + t)
    v = torch.relu(bn(conv(t, "value/convolution"), "value/batch_norm"))                       # [B,H,W,1]
    v = torch.einsum("bhwo,od->bhwd", v, w["value/dense_1/kernel"]) + w["value/dense_1/bias"]  # dense on the last axis
    v = torch.relu(v.sum(dim=(1, 2)))                                                           # reduce_sum over H, W
    value = torch.tanh((torch.einsum("bd,do->bo", v, w["value/dense_2/kernel"]) + w["value/dense_2/bias"]).sum(dim=1))
    p = torch.relu(bn(conv(t, "policy/convolution"), "policy/batch_norm"))                     # [B,H,W,2]
    logits = (torch.einsum("bhwo,oa->bhwa", p, w["policy/policy/kernel"]) + w["policy/policy/bias"]).sum(dim=(1, 2))
    z = logits - logits.max(dim=1, keepdim=True).values
    base = torch.exp(z) / torch.exp(z).sum(dim=1, keepdim=True)
    policy = (1 - EPS) * base + EPS * nz[None, :]
    policy = policy / policy.sum()                        # over ALL elements, batch axis included (:182)
    l_eval = ((value - ev) ** 2).mean()
    l_pol = -(torch.log(policy) @ pl.t()).mean()          # the full B x B cross matrix (:190-194)
    l2 = [0.5 * (t_ ** 2).sum() for k_, t_ in w.items() if t_.requires_grad and "bias" not in k_]
    l_par = torch.stack(l2).mean()
    return l_eval + l_pol + l_par, (l_eval, l_pol, l_par)


def _ref_weights(w0):
    w = {}
    for k, v in w0.items():
        t = torch.tensor(np.asarray(v, dtype=np.float64))
        if not (k.endswith("moving_mean") or k.endswith("moving_variance")):
            t.requires_grad_(True)
        w[k] = t
    return w


def _ref_grads(w0, batch):
    w = _ref_weights(w0)
    total, parts = _ref_loss(w, *batch)
    names = [k for k, t in w.items() if t.requires_grad]
    gs = torch.autograd.grad(total, [w[k] for k in names])
    return float(total.detach()), [float(p.detach()) for p in parts], {k: g.numpy() for k, g in zip(names, gs)}


class _RefOptimizer:
    """tf.compat.v1.train.{Adam,Momentum,GradientDescent}Optimizer update rules in numpy float64."""

    def __init__(self, kind, momentum=0.9):
        self.kind, self.mom, self.t, self.m, self.v = kind, momentum, 0, {}, {}

    def apply(self, w, grads, lr):
        out = dict(w)
        self.t += 1
        for k, g in grads.items():
            x = np.asarray(w[k], dtype=np.float64)
            if self.kind == "adam":
                m = 0.9 * self.m.get(k, 0.0) + 0.1 * g
                v = 0.999 * self.v.get(k, 0.0) + 0.001 * g * g
                self.m[k], self.v[k] = m, v
                lr_t = lr * np.sqrt(1 - 0.999 ** self.t) / (1 - 0.9 ** self.t)
                out[k] = x - lr_t * m / (np.sqrt(v) + 1e-8)
            elif self.kind == "momentum":
                acc = self.mom * self.m.get(k, 0.0) + g
                self.m[k] = acc
                out[k] = x - lr * acc
            else:
                out[k] = x - lr * g
        return out


def _close(got, want, what):
    scale = max(1.0, float(np.max(np.abs(want))))
    err = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - want)))
    assert err <= TOL * scale, (what, err, scale)


def _check_step(device, kind):
    w0 = W.init_weights(C, F, R, D, A, seed=3, perturb=True)
    tr = Trainer(w0, alpha=ALPHA, epsilon=EPS, optimizer=kind, momentum=0.9, device=device)
    ref_opt = _RefOptimizer(kind, 0.9)
    lr = 1e-2
    for step in range(2):  # the second step exercises the optimiser's slots (m, v, beta powers / the accumulator)
        batch = _batch(40 + step)
        here = {k: np.asarray(v, dtype=np.float64) for k, v in tr.export().items()}
        total, parts, grads = tr.gradients(*batch)
        r_total, r_parts, r_grads = _ref_grads(here, batch)
        _close(total, r_total, (kind, step, "loss"))
        for got, want, name in zip(parts, r_parts, ("lossEvaluation", "lossPolicy", "lossParam")):
            _close(got, want, (kind, step, name))
        assert set(grads) == set(r_grads)
        for k in r_grads:
            _close(grads[k].detach().cpu().numpy(), r_grads[k], (kind, step, "grad", k))
        # The update rule is checked on the gradients the step actually used: Adam maps g to g / (|g| + 3.2e-7) on its
        # first step, which turns the 1e-8 rounding of a float32 gradient near zero into percents of lr -- that
        # conditioning belongs to the optimiser (TF's float32 kernels have it too), not to either implementation.
        g64 = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in grads.items()}
        tr.step(batch[0], batch[1], batch[2], lr, noise=batch[3])
        want_w = ref_opt.apply(here, g64, lr)
        new = tr.export()
        for k in want_w:
            _close(new[k], want_w[k], (kind, step, "weights", k))
            if k in g64 and np.abs(g64[k]).max() > 1e-6:
                assert not np.array_equal(new[k], here[k].astype(np.float32)), (kind, step, "did not move", k)
    return tr


@pytest.mark.parametrize("kind", ["adam", "momentum", "sgd"])
def test_train_step_matches_independent_statement_cpu(kind):
    _check_step("cpu", kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["adam", "momentum", "sgd"])
def test_train_step_matches_independent_statement_gpu(kind, orc):
    tr = _check_step("cuda", kind)
    # the engine after the reload computes the network of the exported weights (and the oracle agrees within 1e-5)
    from blackbird_amd import _lib
    game = _lib.GAME_CONNECT4
    flat = W.flatten(tr.export())
    boards = _batch(7)[0]
    a = _lib.Engine(game, n_slots=2, sims_per_move=2, evaluator=_lib.EVAL_NET)
    a.load_weights(W.flatten(W.init_weights(C, F, R, D, A, seed=3, perturb=True)))
    before = a.net_eval(planes=boards)
    a.load_weights(flat)                      # what Model._weights_changed does after train()
    after = a.net_eval(planes=boards)
    b = _lib.Engine(game, n_slots=2, sims_per_move=2, evaluator=_lib.EVAL_NET)
    b.load_weights(flat)
    fresh = b.net_eval(planes=boards)
    assert not np.array_equal(before[1], after[1])
    for x, y in zip(after, fresh):
        assert np.array_equal(x, y)
    ov, ol, op = orc.net_forward(orc.NetWeights(H, Wd, C, F, R, D, A, flat), boards)
    assert np.max(np.abs(after[0] - ov)) <= TOL and np.max(np.abs(after[1] - ol) / np.maximum(1.0, np.abs(ol))) <= TOL
    # and the trainer's own forward (the graph that was differentiated) is that network too
    with torch.no_grad():
        tv, tl = tr.forward(torch.tensor(boards.astype(np.float32), device=tr.device))
    assert np.max(np.abs(tv.cpu().numpy() - after[0])) <= TOL
    assert np.max(np.abs(tl.cpu().numpy() - after[1]) / np.maximum(1.0, np.abs(after[1]))) <= TOL
    a.close()
    b.close()


def test_teacher_is_refused_and_reload_resets_the_trainer(tmp_path, monkeypatch):
    """train(teacher=...) must not silently drop the term; loadModel must not leave a stale optimiser."""
    monkeypatch.chdir(tmp_path)
    from blackbird_amd.Network import Network
    from blackbird_amd.NetworkFactory import NetworkFactory
    cfg = {"blocks": 1, "filters": 16, "eval": {"dense": 16}, "hasTeacher": False,
           "policy": {"dirichlet": {"alpha": 0.2, "epsilon": 0.3}}, "training": {"optimizer": "adam"}}
    net = Network("t_1", NetworkFactory(cfg, 7, inputShape=(6, 7, 3)))
    with pytest.raises(NotImplementedError):
        net.train(np.zeros((2, 6, 7, 3), np.int8), np.zeros(2), np.zeros((2, 7)), teacher=object())
    net._trainer = object()  # stands for a trainer built from older weights
    assert net.loadModel("t_1") and net._trainer is None
