"""NumPy fp32 restatement of the Keras pieces the reference's hot path uses.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Everything here is float32 and
mirrors the semantics listed in SURVEY.md section 8a ("Keras semantics the
restatement must reproduce"):

* network  Flatten -> Dense(h, LeakyReLU(0.1)) -> Dense(h, LeakyReLU(0.1)) ->
  Dense(out[, softmax])                      (reference main.py:59-82)
* Dense kernels are [in, out], y = x @ W + b
* MeanSquaredError / SparseCategoricalCrossentropy with sample_weight,
  reduction SUM_OVER_BATCH_SIZE (sum of weighted per-sample losses / batch)
* plain SGD, Adam in the `lr*sqrt(1-b2^t)/(1-b1^t)`, `m/(sqrt(v)+eps)` form
  with Keras defaults b1=.9 b2=.999 eps=1e-7
* LeakyReLU gradient: 1 where z>0 else alpha

Parameter lists are always in Keras order [W1, b1, W2, b2, W3, b3].
"""
import numpy as np

F32 = np.float32
LEAK = F32(0.1)


def glorot_uniform(rng, fan_in, fan_out):
    """Keras default Dense kernel init: U(+-sqrt(6/(fan_in+fan_out)))."""
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(F32)


def init_mlp(rng, in_dim, hidden, out_dim):
    """Glorot kernels, zero biases (reference main.py:59-82, Keras defaults)."""
    return [
        glorot_uniform(rng, in_dim, hidden), np.zeros(hidden, F32),
        glorot_uniform(rng, hidden, hidden), np.zeros(hidden, F32),
        glorot_uniform(rng, hidden, out_dim), np.zeros(out_dim, F32),
    ]


def copy_params(params):
    return [np.array(p, dtype=F32, copy=True) for p in params]


def lrelu(z):
    return np.where(z > 0, z, LEAK * z).astype(F32, copy=False)


def lrelu_grad(z):
    return np.where(z > 0, F32(1.0), LEAK).astype(F32, copy=False)


def forward(params, x, want_cache=False):
    """Linear-head forward.  x: [B, in] fp32.  Returns logits/values [B, out]
    (and the cache needed by ``backward``).  The softmax of an actor is applied
    by the caller (``softmax``) so that the CE loss can use the logits."""
    W1, b1, W2, b2, W3, b3 = params
    x = np.asarray(x, dtype=F32)
    z1 = x @ W1 + b1
    a1 = lrelu(z1)
    z2 = a1 @ W2 + b2
    a2 = lrelu(z2)
    out = a2 @ W3 + b3
    if want_cache:
        return out, (x, z1, a1, z2, a2)
    return out


def features(params, x):
    """Output of the second hidden layer (post-activation), i.e. the
    reference's ``critic_features`` / ``TR_features`` sub-model
    (agents/resilient_CAC_agents.py:39-40)."""
    W1, b1, W2, b2 = params[:4]
    x = np.asarray(x, dtype=F32)
    return lrelu(lrelu(x @ W1 + b1) @ W2 + b2)


def softmax(logits):
    m = logits.max(axis=-1, keepdims=True)
    e = np.exp(logits - m, dtype=F32)
    return (e / e.sum(axis=-1, keepdims=True)).astype(F32)


def backward(params, cache, dout, hidden_trainable=True):
    """Gradients of sum(dout * out) wrt all six arrays.  With
    ``hidden_trainable=False`` (the *_update_team case,
    agents/resilient_CAC_agents.py:69,82) hidden grads are returned as None."""
    W1, b1, W2, b2, W3, b3 = params
    x, z1, a1, z2, a2 = cache
    dout = np.asarray(dout, dtype=F32)
    gW3 = a2.T @ dout
    gb3 = dout.sum(axis=0)
    if not hidden_trainable:
        return [None, None, None, None, gW3.astype(F32), gb3.astype(F32)]
    da2 = dout @ W3.T
    dz2 = da2 * lrelu_grad(z2)
    gW2 = a1.T @ dz2
    gb2 = dz2.sum(axis=0)
    da1 = dz2 @ W2.T
    dz1 = da1 * lrelu_grad(z1)
    gW1 = x.T @ dz1
    gb1 = dz1.sum(axis=0)
    return [g.astype(F32) for g in (gW1, gb1, gW2, gb2, gW3, gb3)]


def mse_loss_and_dout(pred, y, sample_weight=None):
    """Keras MeanSquaredError, SUM_OVER_BATCH_SIZE.  pred,y: [B,1].
    Returns (scalar loss, dLoss/dpred [B,1])."""
    pred = np.asarray(pred, dtype=F32)
    y = np.asarray(y, dtype=F32).reshape(pred.shape)
    B = pred.shape[0]
    diff = pred - y
    per = np.mean(diff * diff, axis=-1)          # mean over the output dim (size 1)
    if sample_weight is not None:
        w = np.asarray(sample_weight, dtype=F32).reshape(B)
        per = per * w
    else:
        w = None
    loss = F32(per.sum(dtype=F32) / F32(B))
    dout = (F32(2.0) * diff) / F32(pred.shape[-1]) / F32(B)
    if w is not None:
        dout = dout * w[:, None]
    return loss, dout.astype(F32)


def sparse_ce_loss_and_dlogits(logits, labels, sample_weight=None):
    """Keras SparseCategoricalCrossentropy on a softmax output, computed from
    the logits (log-softmax), SUM_OVER_BATCH_SIZE with sample weights
    (agents/resilient_CAC_agents.py:38,99)."""
    logits = np.asarray(logits, dtype=F32)
    B = logits.shape[0]
    lab = np.asarray(labels).reshape(B).astype(np.int64)
    m = logits.max(axis=-1, keepdims=True)
    sh = logits - m
    lse = np.log(np.exp(sh, dtype=F32).sum(axis=-1, keepdims=True), dtype=F32)
    logp = sh - lse
    per = -logp[np.arange(B), lab]
    p = np.exp(logp, dtype=F32)
    if sample_weight is not None:
        w = np.asarray(sample_weight, dtype=F32).reshape(B)
    else:
        w = np.ones(B, F32)
    loss = F32((per * w).sum(dtype=F32) / F32(B))
    onehot = np.zeros_like(p)
    onehot[np.arange(B), lab] = 1
    dlogits = (p - onehot) * (w / F32(B))[:, None]
    return loss, dlogits.astype(F32)


def sgd_apply(params, grads, lr):
    """In-place plain SGD; None grads (frozen layers) are skipped."""
    lr = F32(lr)
    for p, g in zip(params, grads):
        if g is not None:
            p -= lr * g


class AdamState:
    """Keras Adam slot variables + iteration counter for one model."""

    def __init__(self, params, lr, beta1=0.9, beta2=0.999, eps=1e-7):
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(beta1), float(beta2), float(eps)
        self.t = 0
        self.m = [np.zeros_like(p, dtype=F32) for p in params]
        self.v = [np.zeros_like(p, dtype=F32) for p in params]


def adam_apply(params, grads, st):
    """TF2 `ResourceApplyAdam` form: alpha = lr*sqrt(1-b2^t)/(1-b1^t);
    m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= alpha*m/(sqrt(v)+eps)."""
    st.t += 1
    b1, b2 = st.beta1, st.beta2
    alpha = F32(st.lr * np.sqrt(1.0 - b2 ** st.t) / (1.0 - b1 ** st.t))
    one_m_b1, one_m_b2, eps = F32(1.0 - b1), F32(1.0 - b2), F32(st.eps)
    for p, g, m, v in zip(params, grads, st.m, st.v):
        if g is None:
            continue
        m += (g - m) * one_m_b1
        v += (g * g - v) * one_m_b2
        p -= (m * alpha) / (np.sqrt(v) + eps)


def fit_mse(params, x, y, lr, epochs, batch_size=None, perms=None, sample_weight=None):
    """Keras ``model.fit(x, y, batch_size, epochs)`` with an SGD optimizer and MSE
    loss, all layers trainable.  ``perms`` is an [epochs, B] int array giving the
    per-epoch shuffle (Keras shuffles with TF's RNG, which is irreproducible
    here: the oracle *defines* the shuffle -- SURVEY.md 8c).  With
    ``batch_size >= B`` the shuffle only permutes the single batch, so
    ``perms=None`` keeps the natural row order.  Returns the per-epoch losses
    (sample-weighted mean of the batch losses, as Keras' History does)."""
    x = np.asarray(x, dtype=F32)
    y = np.asarray(y, dtype=F32).reshape(x.shape[0], -1)
    B = x.shape[0]
    bs = B if batch_size is None else min(int(batch_size), B)
    hist = []
    for e in range(epochs):
        order = np.arange(B) if perms is None else np.asarray(perms[e])
        tot, cnt = 0.0, 0
        for lo in range(0, B, bs):
            idx = order[lo:lo + bs]
            xb, yb = (x, y) if (bs == B and perms is None) else (x[idx], y[idx])
            pred, cache = forward(params, xb, want_cache=True)
            loss, dout = mse_loss_and_dout(pred, yb)
            grads = backward(params, cache, dout)
            sgd_apply(params, grads, lr)
            tot += float(loss) * len(idx)
            cnt += len(idx)
        hist.append(F32(tot / cnt))
    return hist


def fit_actor_ce(params, adam, x, labels, sample_weight, epochs=1, batch_size=None, perms=None):
    """``actor.fit``/``train_on_batch`` with Adam + sparse CE + sample weights.
    Returns per-epoch losses."""
    x = np.asarray(x, dtype=F32)
    B = x.shape[0]
    bs = B if batch_size is None else min(int(batch_size), B)
    lab = np.asarray(labels).reshape(B)
    w = np.asarray(sample_weight, dtype=F32).reshape(B)
    hist = []
    for e in range(epochs):
        order = np.arange(B) if perms is None else np.asarray(perms[e])
        tot, cnt = 0.0, 0
        for lo in range(0, B, bs):
            idx = order[lo:lo + bs]
            logits, cache = forward(params, x[idx], want_cache=True)
            loss, dlogits = sparse_ce_loss_and_dlogits(logits, lab[idx], w[idx])
            grads = backward(params, cache, dlogits)
            adam_apply(params, grads, adam)
            tot += float(loss) * len(idx)
            cnt += len(idx)
        hist.append(F32(tot / cnt))
    return hist
