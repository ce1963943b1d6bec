"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the V-GAN training hot path.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``v-gan_amd/``) must never import anything under ``oracle/``.

It restates, in plain numpy (dtype-generic: run it in float32 to mimic the
reference's fp32 path, in float64 for a high-precision answer), the algorithm of
one ``VGAN_no_kl.fit`` step of jcribeiro98/V-GAN:

    z -> Generator_big (4 bias-Linear, no activation) -> upper_softmax -> U
      -> Y = U * X -> MMDLossConstrained(X, Y, U) -> backward -> Adadelta

Every function cites the reference ``file:line`` it follows (paths relative to
the reference checkout).  Parity status: **pinned** -- ``tests/test_oracle_golden.py``
checks every function here against fixtures in ``tests/golden/*.npz`` that were
produced by importing and running the reference itself (``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import numpy as np

# src/models/Mmd_loss_constrained.py:12-13  -> mul_factor ** (arange(n_kernels) - n_kernels // 2)
N_KERNELS = 5
MUL_FACTOR = 2.0
BANDWIDTH_MULTIPLIERS = MUL_FACTOR ** (np.arange(N_KERNELS) - N_KERNELS // 2)  # [.25,.5,1,2,4]

# torch.optim.Adadelta defaults used at src/vgan.py:567-568 / 207-210
ADADELTA_RHO = 0.9
ADADELTA_EPS = 1e-6


# --------------------------------------------------------------------------------------
# Generator_big : src/models/Generator.py:58-70
# --------------------------------------------------------------------------------------
def generator_layer_sizes(latent: int, d: int):
    """[(out,in)] of the four Linear layers, src/models/Generator.py:61-66."""
    return [(2 * latent, latent), (4 * latent, 2 * latent), (8 * latent, 4 * latent), (d, 8 * latent)]


def latent_size(d: int) -> int:
    """src/vgan.py:559 (and :196): max(int(d/16), 1)."""
    return max(int(d / 16), 1)


def generator_forward(params, z):
    """params = [W1,b1,...,W4,b4] with W [out,in] (PyTorch layout).  Returns (logits, acts)
    where acts[k] is the input of layer k (acts[0] = z).  src/models/Generator.py:61-66, 69-70."""
    acts = [z]
    h = z
    for k in range(4):
        W, b = params[2 * k], params[2 * k + 1]
        h = h @ W.T + b
        acts.append(h)
    return h, acts[:-1]


def softmax_rows(x):
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=1, keepdims=True)


def upper_softmax_forward(logits):
    """src/models/Generator.py:18-22.  s = softmax(x,1); out = (s < 1/d)*s + (s >= 1/d).
    The python scalar 1/d is compared in the tensor's dtype.  Returns (U, s)."""
    s = softmax_rows(logits)
    tau = logits.dtype.type(1.0 / logits.shape[1])
    U = np.where(s < tau, s, logits.dtype.type(1.0))
    return U, s


def upper_softmax_backward(gU, s):
    """Autograd of Generator.py:19-21: only the (s<tau)*s branch carries gradient, then the
    softmax Jacobian g_logit = s * (g_s - sum_j g_s s)."""
    tau = s.dtype.type(1.0 / s.shape[1])
    gs = np.where(s < tau, gU, s.dtype.type(0.0))
    dot = (gs * s).sum(axis=1, keepdims=True)
    return s * (gs - dot)


def subspace_mask(U):
    """src/vgan.py:646 -- torch.greater_equal(u, 1/d)."""
    return U >= U.dtype.type(1.0 / U.shape[1])


# --------------------------------------------------------------------------------------
# RBF + MMDLossConstrained : src/models/Mmd_loss_constrained.py
# --------------------------------------------------------------------------------------
def squared_distances(Z):
    """torch.cdist(Z, Z) ** 2, src/models/Mmd_loss_constrained.py:25.  ATen's p=2 path for
    more than 25 rows is the matmul form |a|^2+|b|^2-2ab, clamp_min(0), sqrt; then **2."""
    s = (Z * Z).sum(axis=1)
    g = Z @ Z.T
    L = np.maximum(s[:, None] + s[None, :] - 2 * g, Z.dtype.type(0.0))
    D = np.sqrt(L)
    return D * D


def rbf_bandwidth(L):
    """src/models/Mmd_loss_constrained.py:16-22 -- sum(L) / (N^2 - N), first call only."""
    N = L.shape[0]
    return L.sum() / L.dtype.type(N * N - N)


def rbf_scales(bw, dtype):
    """bw * bandwidth_multipliers, src/models/Mmd_loss_constrained.py:26.  The multipliers are a
    float32 tensor (2.0 ** int64 arange), and a 0-dim bandwidth does not promote it, so the five
    scales are rounded to float32 even when the data is float64 -- reproduced here."""
    return (np.float32(bw) * BANDWIDTH_MULTIPLIERS.astype(np.float32)).astype(dtype)


def rbf_kernel(L, bw):
    """src/models/Mmd_loss_constrained.py:26 -- sum_k exp(-L / (bw * m_k))."""
    K = np.zeros_like(L)
    for sc in rbf_scales(bw, L.dtype):
        K += np.exp(-L / sc)
    return K


def mmd_forward(X, Y, U, weight, bw=None):
    """MMDLossConstrained.forward, src/models/Mmd_loss_constrained.py:42-50.
    Returns dict(loss, mmd2, penalty, bw, xx, xy, yy).  ``bw=None`` reproduces the first-call
    calibration (Mmd_loss_constrained.py:16-20); otherwise the frozen bandwidth is used."""
    n = X.shape[0]
    Z = np.vstack([X, Y])
    L = squared_distances(Z)
    if bw is None:
        bw = rbf_bandwidth(L)
    bw = L.dtype.type(bw)
    K = rbf_kernel(L, bw)
    xx = K[:n, :n].mean()
    xy = K[:n, n:].mean()
    yy = K[n:, n:].mean()
    mmd2 = xx - 2 * xy + yy
    penalty = weight * np.mean(1.0 - U.max(axis=0))  # mean(ones(d) - topk(U,1,0).values)
    return dict(loss=mmd2 + penalty, mmd2=mmd2, penalty=penalty, bw=bw, xx=xx, xy=xy, yy=yy)


def mmd_backward(X, Y, U, weight, bw, with_dx=False):
    """Closed-form gradient of MMDLossConstrained w.r.t. Y and (the explicit) U argument
    (verified against the reference's autograd by the golden tests).
    Returns (dY, dU_penalty): dU_penalty is only the penalty term's gradient; the caller adds
    dY * X for the path through Y = U * X (src/vgan.py:616).  X and Y may have different row counts (the block means of
    Mmd_loss_constrained.py:46-49 are over n_x^2, n_x n_y and n_y^2 entries); with_dx: returns (dX, dY, dU_penalty)."""
    n, m = X.shape[0], Y.shape[0]
    d = U.shape[1]
    dt = X.dtype
    Z = np.vstack([X, Y])
    s = (Z * Z).sum(axis=1)
    L = np.maximum(s[:, None] + s[None, :] - 2 * (Z @ Z.T), dt.type(0.0))
    dK = np.zeros_like(L)
    for sc in rbf_scales(bw, dt):
        dK -= np.exp(-L / sc) / sc
    c = np.zeros_like(L)
    c[:n, :n] = 1.0 / (n * n)
    c[n:, n:] = 1.0 / (m * m)
    c[:n, n:] = -2.0 / (n * m)
    G = c * dK
    Gs = G + G.T
    dZ = 2 * (Gs.sum(axis=1, keepdims=True) * Z - Gs @ Z)
    dY = dZ[n:]
    dU = np.zeros_like(U)
    arg = U.argmax(axis=0)  # first maximal row, like a stable top-1
    dU[arg, np.arange(d)] = -weight / d
    if with_dx:
        return dZ[:n].astype(dt), dY.astype(dt), dU.astype(dt)
    return dY.astype(dt), dU.astype(dt)


# --------------------------------------------------------------------------------------
# Generator backward + Adadelta
# --------------------------------------------------------------------------------------
def generator_backward(params, acts, dlogits):
    """Gradients of the 4 Linear layers: dW = dout^T @ in, db = colsum(dout), din = dout @ W."""
    grads = [None] * 8
    g = dlogits
    for k in (3, 2, 1, 0):
        W = params[2 * k]
        grads[2 * k] = g.T @ acts[k]
        grads[2 * k + 1] = g.sum(axis=0)
        if k:
            g = g @ W
    return grads


def adadelta_step(p, g, sq, acc, lr, weight_decay, rho=ADADELTA_RHO, eps=ADADELTA_EPS):
    """One torch.optim.Adadelta update (src/vgan.py:567-568, :619): returns (p, sq, acc)."""
    dt = p.dtype.type
    g = g + dt(weight_decay) * p
    sq = dt(rho) * sq + dt(1 - rho) * g * g
    std = np.sqrt(sq + dt(eps))
    delta = np.sqrt(acc + dt(eps)) / std * g
    acc = dt(rho) * acc + dt(1 - rho) * delta * delta
    p = p - dt(lr) * delta
    return p, sq, acc


# --------------------------------------------------------------------------------------
# One full step / a trajectory of VGAN_no_kl.fit : src/vgan.py:597-621
# --------------------------------------------------------------------------------------
def step_forward_backward(params, X, z, weight, bw=None):
    """noise -> G -> mask -> project -> MMD -> backward (src/vgan.py:613-618).
    Returns dict(loss, bw, U, s, Y, dY, dU, dlogits, grads)."""
    logits, acts = generator_forward(params, z)
    U, s = upper_softmax_forward(logits)
    Y = U * X
    f = mmd_forward(X, Y, U, weight, bw)
    dY, dUp = mmd_backward(X, Y, U, weight, f["bw"])
    dU = dY * X + dUp
    dlogits = upper_softmax_backward(dU, s)
    grads = generator_backward(params, acts, dlogits)
    return dict(loss=f["loss"], mmd2=f["mmd2"], bw=f["bw"], xx=f["xx"], xy=f["xy"], yy=f["yy"],
                U=U, s=s, Y=Y, dY=dY, dU=dU, dlogits=dlogits, grads=grads, logits=logits)


class NoKLTrainer:
    """Restates the VGAN_no_kl.fit loop (src/vgan.py:546-637) on caller-provided batches and
    noise (the golden fixtures record both, so no RNG stream has to be reproduced)."""

    def __init__(self, params, lr=0.007, weight_decay=0.04, weight=10.0, bw=None):
        self.params = [p.copy() for p in params]
        self.sq = [np.zeros_like(p) for p in params]
        self.acc = [np.zeros_like(p) for p in params]
        self.lr, self.weight_decay, self.weight = lr, weight_decay, weight
        self.bw = bw

    def step(self, X, z):
        out = step_forward_backward(self.params, X, z, self.weight, self.bw)
        self.bw = out["bw"]  # frozen after the first call (Mmd_loss_constrained.py:16-22)
        for i in range(8):
            self.params[i], self.sq[i], self.acc[i] = adadelta_step(
                self.params[i], out["grads"][i], self.sq[i], self.acc[i], self.lr, self.weight_decay)
        return out

    def masks(self, z):
        """generate_subspaces (src/vgan.py:639-647) on provided noise."""
        logits, _ = generator_forward(self.params, z)
        U, _ = upper_softmax_forward(logits)
        return subspace_mask(U)


# --------------------------------------------------------------------------------------
# Documented synthetic inputs -- shared by tests and bench.py
# --------------------------------------------------------------------------------------
def synthetic_dataset(config: str, rows: int | None = None, seed: int = 0):
    """float32 datasets of SURVEY 8(d).  c1: 2-Gaussian mixture d=20; c2: musk stand-in d=166;
    c3: MNIST-pixel stand-in d=784 (values in [0,1], ~80% zeros, rank-32 + noise);
    c4 / c5: synthetic tabular d=2048 / 4096, N(0,1) with 64 planted correlated feature blocks."""
    rng = np.random.default_rng(seed)
    if config == "c1":
        d, n = 20, 128
        rows = rows or 16 * n
        sign = np.where(rng.random(rows) < 0.5, -2.0, 2.0)[:, None]
        X = rng.normal(size=(rows, d)) + sign
    elif config == "c2":
        d, n = 166, 512
        rows = rows or 3062
        blocks = rng.normal(size=(rows, 16)) @ rng.normal(size=(16, d))
        X = blocks + 0.5 * rng.normal(size=(rows, d))
        X = (X - X.mean(0)) / X.std(0)
    elif config == "c3":
        d, n = 784, 1024
        rows = rows or 16 * n
        low = rng.random(size=(rows, 32)) @ rng.random(size=(32, d)) / 16.0
        X = np.clip(low + 0.05 * rng.normal(size=(rows, d)), 0.0, 1.0)
        X = X * (rng.random(size=(rows, d)) < 0.2)
    elif config in ("c4", "c5"):
        # synthetic tabular: N(0,1) features with 64 planted correlated blocks (each block shares one latent factor)
        d, n = (2048, 4096) if config == "c4" else (4096, 8192)
        rows = rows or 4 * n
        rng32 = np.random.default_rng(seed + (4 if config == "c4" else 5))
        X = rng32.standard_normal(size=(rows, d), dtype=np.float32)
        factors = rng32.standard_normal(size=(rows, 64), dtype=np.float32)
        width = d // 128  # 64 blocks covering half of the features
        for b in range(64):
            X[:, 2 * b * width:(2 * b + 1) * width] = 0.6 * X[:, 2 * b * width:(2 * b + 1) * width] + 0.8 * factors[:, b:b + 1]
    else:
        raise ValueError(config)
    return np.ascontiguousarray(X, dtype=np.float32)


def synthetic_generator_params(d: int, seed: int = 0, dtype=np.float32):
    """PyTorch-default-like Linear init (U(-1/sqrt(in), 1/sqrt(in))) from a numpy stream."""
    rng = np.random.default_rng(seed + 1000)
    L = latent_size(d)
    params = []
    for out, inp in generator_layer_sizes(L, d):
        k = 1.0 / np.sqrt(inp)
        params.append(rng.uniform(-k, k, size=(out, inp)).astype(dtype))
        params.append(rng.uniform(-k, k, size=(out,)).astype(dtype))
    return params


# ---- myopicity two-sample test (check_if_myopic, src/vgan.py:384-431) ------------------------------------------------
# The reference delegates to torch-two-sample (josipd/torch-two-sample, unpinned, not vendored, not installed here):
# MMDStatistic(n1, n2)(x, y, alphas=[a], ret_matrix=True) and .pval(matrix, n_permutations=1000).  The functions below
# restate that package's published algorithm; nothing in /root/reference pins them: PARITY UNPINNED.
def two_sample_kernel_matrix(X, Y, alpha):
    """exp(-alpha * |z_i - z_j|^2) over the pooled sample [X ; Y] (MMDStatistic.__call__ with one alpha)."""
    Z = np.vstack([X, Y]).astype(np.float64)
    L = squared_distances(Z)
    K = np.exp(-float(alpha) * L)
    np.fill_diagonal(K, 1.0)
    return K


def two_sample_statistic(K, group1):
    """Permutation statistic of torch-two-sample's permutation_test_mat for one assignment (group1: bool [m]):
    sum_{i<=j} (K_ij + K_ji) * a(i, j), a = a00 / a11 inside a group, a01 across, with a00 = 1/(n1(n1-1)),
    a11 = 1/(n2(n2-1)), a01 = -1/(n1 n2).  In closed form: a00 (u'Ku + tr_u) + a11 (v'Kv + tr_v) + 2 a01 u'Kv."""
    u = np.asarray(group1, dtype=np.float64)
    v = 1.0 - u
    n1, n2 = u.sum(), v.sum()
    a00, a11, a01 = 1.0 / (n1 * (n1 - 1)), 1.0 / (n2 * (n2 - 1)), -1.0 / (n1 * n2)
    Ku, Kv = K @ u, K @ v
    dg = np.diag(K)
    return a00 * (u @ Ku + dg @ u) + a11 * (v @ Kv + dg @ v) + 2.0 * a01 * (u @ Kv)


def two_sample_pvalue(K, n1, assignments):
    """p = #{permuted statistic >= observed} / #permutations; `assignments` [P, m] bool are the shuffled group-1
    indicators (the package shuffles with numpy's global generator; callers pass theirs)."""
    m = K.shape[0]
    observed = np.zeros(m, dtype=bool)
    observed[:n1] = True
    s0 = two_sample_statistic(K, observed)
    larger = sum(1 for a in assignments if s0 <= two_sample_statistic(K, a))
    return larger / float(len(assignments)), s0
