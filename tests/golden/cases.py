"""Shared definitions for the golden fixtures (data only -- no reference code).

`CASES` names the shapes SURVEY.md 8(c) asks for.  Small shapes pin the
reference's torch-seeded initialisation (G1); for the large benchmark shapes the
weights are drawn here from numpy's PCG64 (portable across numpy versions) and
were loaded into the reference with `load_state_dict` when the fixture was
made, so that the .npz files only need to hold inputs and expected outputs.
"""
import numpy as np

# name: (L, d, c, hidden, activation, weights)   weights: "torch" (stored in fixture) | "numpy" (regenerated)
CASES = {
    "c1_L4":     (4, 2, 1, (10,), "tanh", "torch"),     # BASELINE.json configs[0] ("4 coupling layers")
    "c1_L8":     (8, 2, 1, (10,), "tanh", "torch"),     # README.md:51-59 (defaults)
    "tm":        (8, 5, 3, (10,), "tanh", "torch"),     # tests/test_models.py:12-18
    "tm_nocond": (8, 5, 0, (10,), "tanh", "torch"),     # tests/test_models.py:23-28
    "reg1d":     (4, 1, 1, (10,), "tanh", "torch"),     # docs/examples/regression.ipynb cell 9
    "relu_mh":   (3, 6, 2, (7, 9), "relu", "torch"),    # G9
    "relu_sh":   (6, 5, 3, (24,), "relu", "torch"),     # ReLU nets with one hidden layer (realnvp.py:32-37): MFMA path
    "tanh_mh":   (4, 4, 2, (8, 8), "tanh", "torch"),    # G9
    "d8":        (4, 8, 4, (32,), "tanh", "torch"),
    "c2":        (8, 16, 4, (128,), "tanh", "numpy"),   # BASELINE.json configs[1]
    "c3":        (12, 32, 8, (256,), "tanh", "numpy"),  # configs[2]
    "c4":        (8, 64, 16, (128,), "tanh", "numpy"),  # configs[3]
    "c2_nocond": (8, 16, 0, (128,), "tanh", "numpy"),
}


GRAD_STRIDE = 53   # large cases store grad[::GRAD_STRIDE] plus its l2 norm and sum


def linear_shapes(d, c, hidden):
    """[(out, in), ...] of one s/t net -- gen_network, realnvp.py:19-43."""
    dims = [d + c] + list(hidden) + [d]
    return [(dims[k + 1], dims[k]) for k in range(len(dims) - 1)]


def param_count(L, d, c, hidden):
    return 2 * L * sum(o * i + o for o, i in linear_shapes(d, c, hidden))


def numpy_params(name, scale=1.0):
    """Flat parameter vector in nf.parameters() order, U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
    L, d, c, hidden, _, _ = CASES[name]
    rng = np.random.default_rng(abs(hash_name(name)))
    parts = []
    for _l in range(L):
        for _net in range(2):
            for (o, i) in linear_shapes(d, c, hidden):
                b = scale / np.sqrt(i)
                parts.append(rng.uniform(-b, b, size=o * i))
                parts.append(rng.uniform(-b, b, size=o))
    return np.concatenate(parts).astype(np.float32)


def hash_name(name):
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def inputs(name, n, seed=0):
    """Seeded inputs X [n,d], C [n,c] (float32, standard normal) and z for the inverse."""
    L, d, c, hidden, _, _ = CASES[name]
    rng = np.random.default_rng(1000 + seed + hash_name(name) % 1000)
    X = rng.normal(size=(n, d)).astype(np.float32)
    C = rng.normal(size=(n, c)).astype(np.float32) if c > 0 else None
    Z = rng.normal(size=(n, d)).astype(np.float32)
    return X, C, Z
