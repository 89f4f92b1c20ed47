"""Weight container for the residual tower + heads.

Variables carry the names the reference's TensorFlow graph gives them
(/root/reference/src/NetworkFactory.py:22-183, scopes `resTower/conv_block`, `resTower/block_{i}`,
`value`, `policy`), so a checkpoint exported from the original as name->array can be loaded as is
(SURVEY.md 2.3 / 8f-f4).  `flatten()` regroups them into the contiguous float32 blocks the C ABI
(include/blackbird_hip.h: bb_net_weights) takes.
"""
import numpy as np

BN_FIELDS = ("gamma", "beta", "moving_mean", "moving_variance")


def variable_names(blocks):
    names = ["resTower/conv_block/conv/kernel", "resTower/conv_block/conv/bias"]
    names += [f"resTower/conv_block/batch_norm/{f}" for f in BN_FIELDS]
    for i in range(blocks):
        for j in (1, 2):
            names += [f"resTower/block_{i}/conv_{j}/kernel", f"resTower/block_{i}/conv_{j}/bias"]
            names += [f"resTower/block_{i}/batch_norm_{j}/{f}" for f in BN_FIELDS]
    for head in ("value", "policy"):
        names += [f"{head}/convolution/kernel", f"{head}/convolution/bias"]
        names += [f"{head}/batch_norm/{f}" for f in BN_FIELDS]
    names += ["value/dense_1/kernel", "value/dense_1/bias", "value/dense_2/kernel", "value/dense_2/bias",
              "policy/policy/kernel", "policy/policy/bias"]
    return names


def _glorot(rng, shape):
    # TF default kernel_initializer = glorot_uniform: limit = sqrt(6 / (fan_in + fan_out)),
    # fan_in = prod(shape[:-2]) * shape[-2], fan_out = prod(shape[:-2]) * shape[-1]
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def init_weights(in_planes, filters, blocks, dense, actions, seed=0, perturb=False):
    """Random-init weights as `global_variables_initializer` would leave them (Network.py:23):
    glorot-uniform kernels, zero biases, gamma=1, beta=0, moving_mean=0, moving_variance=1.
    perturb=True also randomises biases and batch-norm statistics (used by parity tests so that
    every term of the graph is exercised)."""
    rng = np.random.RandomState(seed)
    C, F, D, A = in_planes, filters, dense, actions
    w = {}

    def bn(prefix, n):
        if perturb:
            w[f"{prefix}/gamma"] = rng.uniform(0.5, 1.5, n).astype(np.float32)
            w[f"{prefix}/beta"] = rng.uniform(-0.3, 0.3, n).astype(np.float32)
            w[f"{prefix}/moving_mean"] = rng.uniform(-0.3, 0.3, n).astype(np.float32)
            w[f"{prefix}/moving_variance"] = rng.uniform(0.5, 2.0, n).astype(np.float32)
        else:
            w[f"{prefix}/gamma"] = np.ones(n, np.float32)
            w[f"{prefix}/beta"] = np.zeros(n, np.float32)
            w[f"{prefix}/moving_mean"] = np.zeros(n, np.float32)
            w[f"{prefix}/moving_variance"] = np.ones(n, np.float32)

    def bias(name, n):
        w[name] = (rng.uniform(-0.2, 0.2, n) if perturb else np.zeros(n)).astype(np.float32)

    w["resTower/conv_block/conv/kernel"] = _glorot(rng, (3, 3, C, F))
    bias("resTower/conv_block/conv/bias", F)
    bn("resTower/conv_block/batch_norm", F)
    for i in range(blocks):
        for j in (1, 2):
            w[f"resTower/block_{i}/conv_{j}/kernel"] = _glorot(rng, (3, 3, F, F))
            bias(f"resTower/block_{i}/conv_{j}/bias", F)
            bn(f"resTower/block_{i}/batch_norm_{j}", F)
    w["value/convolution/kernel"] = _glorot(rng, (1, 1, F, 1))
    bias("value/convolution/bias", 1)
    bn("value/batch_norm", 1)
    w["value/dense_1/kernel"] = _glorot(rng, (1, D))
    bias("value/dense_1/bias", D)
    w["value/dense_2/kernel"] = _glorot(rng, (D, 1))
    bias("value/dense_2/bias", 1)
    w["policy/convolution/kernel"] = _glorot(rng, (1, 1, F, 2))
    bias("policy/convolution/bias", 2)
    bn("policy/batch_norm", 2)
    w["policy/policy/kernel"] = _glorot(rng, (2, A))
    bias("policy/policy/bias", A)
    return w


def infer_shape(w):
    k0 = w["resTower/conv_block/conv/kernel"]
    C, F = k0.shape[2], k0.shape[3]
    R = 0
    while f"resTower/block_{R}/conv_1/kernel" in w:
        R += 1
    D = w["value/dense_1/kernel"].shape[1]
    A = w["policy/policy/kernel"].shape[1]
    return C, F, R, D, A


def flatten(w):
    """TF-named dict -> dict of contiguous float32 blocks named like bb_net_weights' fields."""
    C, F, R, D, A = infer_shape(w)

    def bn(prefix):
        return np.stack([np.asarray(w[f"{prefix}/{f}"], np.float32).ravel() for f in BN_FIELDS])

    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    out = {
        "conv0_k": f32(w["resTower/conv_block/conv/kernel"]),
        "conv0_b": f32(w["resTower/conv_block/conv/bias"]),
        "conv0_bn": f32(bn("resTower/conv_block/batch_norm")),
        "blk_k": f32(np.stack([np.stack([w[f"resTower/block_{i}/conv_{j}/kernel"] for j in (1, 2)])
                               for i in range(R)]) if R else np.zeros((0, 2, 3, 3, F, F))),
        "blk_b": f32(np.stack([np.stack([w[f"resTower/block_{i}/conv_{j}/bias"] for j in (1, 2)])
                               for i in range(R)]) if R else np.zeros((0, 2, F))),
        "blk_bn": f32(np.stack([np.stack([bn(f"resTower/block_{i}/batch_norm_{j}") for j in (1, 2)])
                                for i in range(R)]) if R else np.zeros((0, 2, 4, F))),
        "v_conv_k": f32(w["value/convolution/kernel"]).reshape(F),
        "v_conv_b": f32(w["value/convolution/bias"]).reshape(1),
        "v_bn": f32(bn("value/batch_norm")),
        "v_d1_k": f32(w["value/dense_1/kernel"]).reshape(D),
        "v_d1_b": f32(w["value/dense_1/bias"]).reshape(D),
        "v_d2_k": f32(w["value/dense_2/kernel"]).reshape(D),
        "v_d2_b": f32(w["value/dense_2/bias"]).reshape(1),
        "p_conv_k": f32(w["policy/convolution/kernel"]).reshape(F, 2),
        "p_conv_b": f32(w["policy/convolution/bias"]).reshape(2),
        "p_bn": f32(bn("policy/batch_norm")),
        "p_d_k": f32(w["policy/policy/kernel"]).reshape(2, A),
        "p_d_b": f32(w["policy/policy/bias"]).reshape(A),
    }
    return out


def save_npz(path, w):
    np.savez(path, **{k.replace("/", "."): v for k, v in w.items()})


def load_npz(path):
    with np.load(path, allow_pickle=False) as z:
        return {k.replace(".", "/"): z[k] for k in z.files}
