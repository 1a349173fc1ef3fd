"""oracle.train — numpy restatement of the TensorFlow-1.12 half of the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

TensorFlow 1.12.3 (requirements.txt:4) is not installable here, so the graph
half of the reference (model forward/backward/Adam) cannot be executed; this
module restates its arithmetic from the reference's model files plus the
published TF-1.12 kernels [EXT]:

  MF graph        model/general_recommender/MF.py:54-72, util/learner.py:19-22,
                  util/tool.py:216-217
  LightGCN graph  model/general_recommender/LightGCN.py:34-78 (adjacency),
                  :132-149 (propagation + layer mean), :156-166 (loss)
  Adam            tf.train.AdamOptimizer [EXT]: python/training/adam.py
                  (_apply_sparse_shared: every row decays/updates each step) and
                  core/kernels/training_ops.cc ApplyAdam (dense)
  softplus        core/kernels/softplus_op.h (thresholded form) [EXT]

Every function takes `dtype`: np.float32 is the restatement, np.float64 its
error-bar twin.  PINNED (round 3) to traces of the reference's own model classes
run under oracle/tf_shim.py — tests/test_tfgraph_golden.py holds every function
here to tests/golden/tfgraph_*.npz (fp64 <= 1e-12; fp32 losses and gradients
<= 1e-6).  TensorFlow itself still cannot run here: the primitives' definitions
and the optimiser update rules remain [EXT] restatements (in the shim, cited).
"""
import numpy as np
import scipy.sparse as sp


# ------------------------------------------------------------------ shared pieces
def tf_softplus(z):
    z = np.asarray(z)
    thr = np.log(np.finfo(z.dtype).eps) + 2.0
    with np.errstate(over="ignore"):
        e = np.exp(z)
        return np.where(z > -thr, z, np.where(z < thr, e, np.log1p(e))).astype(z.dtype)


def bpr_terms(x):
    """loss_b = -log_sigmoid(x) = softplus(-x);  dloss/dx = -sigmoid(-x)."""
    x = np.asarray(x)
    with np.errstate(over="ignore"):
        g = (-1.0 / (1.0 + np.exp(x))).astype(x.dtype)
    return tf_softplus(-x), g


class Adam:
    """Running fp32 powers as TF keeps them (beta1_power *= beta1 after each step)."""

    def __init__(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float32):
        self.dt = dtype
        self.lr, self.b1, self.b2, self.eps = dtype(lr), dtype(beta1), dtype(beta2), dtype(eps)
        self.b1p, self.b2p = dtype(beta1), dtype(beta2)

    def alpha(self):
        one = self.dt(1)
        return self.dt(self.lr * np.sqrt(one - self.b2p) / (one - self.b1p))

    def advance(self):
        self.b1p = self.dt(self.b1p * self.b1)
        self.b2p = self.dt(self.b2p * self.b2)

    def dense(self, var, m, v, g):
        """training_ops ApplyAdam, in place."""
        one = self.dt(1)
        a = self.alpha()
        m += (g - m) * (one - self.b1)
        v += (g * g - v) * (one - self.b2)
        var -= (m * a) / (np.sqrt(v) + self.eps)

    def sparse_swept(self, var, m, v, g_dense):
        """adam.py _apply_sparse_shared with the summed row gradient scattered into
        g_dense (zeros elsewhere): all rows decay and move every step."""
        one = self.dt(1)
        a = self.alpha()
        m *= self.b1
        m += g_dense * (one - self.b1)
        v *= self.b2
        v += (g_dense * g_dense) * (one - self.b2)
        var -= a * m / (np.sqrt(v) + self.eps)


# ------------------------------------------------------------------ BPR-MF
def mf_loss_and_grads(P, Q, users, pos, neg, reg):
    """One batch of MF.py:54-72: returns loss, dense dP, dense dQ (duplicates summed)."""
    dt = P.dtype.type
    p, qi, qj = P[users], Q[pos], Q[neg]
    x = np.sum(p * qi, axis=1, dtype=dt) - np.sum(p * qj, axis=1, dtype=dt)
    lb, g = bpr_terms(x)
    l2 = (np.sum(p * p, dtype=dt) + np.sum(qj * qj, dtype=dt) + np.sum(qi * qi, dtype=dt)) / dt(2)
    loss = np.sum(lb, dtype=dt) + dt(reg) * l2
    gp = g[:, None] * (qi - qj) + dt(reg) * p
    gqi = g[:, None] * p + dt(reg) * qi
    gqj = -g[:, None] * p + dt(reg) * qj
    dP = np.zeros_like(P)
    dQ = np.zeros_like(Q)
    np.add.at(dP, users, gp)
    np.add.at(dQ, pos, gqi)
    np.add.at(dQ, neg, gqj)
    return loss, dP, dQ


def pairwise_terms(kind, y):
    """util/learner.py:19-29 per element + derivative: bpr | hinge (max(y+1,0), as written) | square."""
    if kind == "bpr":
        return bpr_terms(y)
    if kind == "hinge":
        return np.maximum(y + 1, 0).astype(y.dtype), (y + 1 > 0).astype(y.dtype)
    if kind == "square":
        return (1 - y) ** 2, -2 * (1 - y)
    raise Exception("please choose a suitable loss function")


def pointwise_terms(kind, z, x):
    """util/learner.py:31-41 per element + d/dx; cross_entropy is tf.losses.sigmoid_cross_entropy
    (stable form, averaged over the batch by the caller), square is summed."""
    if kind == "cross_entropy":
        return (np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))).astype(x.dtype), \
               (1 / (1 + np.exp(-x)) - z).astype(x.dtype)
    if kind == "square":
        return (z - x) ** 2, -2 * (z - x)
    raise Exception("please choose a suitable loss function")


def mf_general_loss_and_grads(P, Q, users, items, third, reg, pairwise, kind):
    """MF.py:62-72 for every (is_pairwise, loss_function): returns (data loss, reg term, dP, dQ)."""
    dt = P.dtype.type
    dP, dQ = np.zeros_like(P), np.zeros_like(Q)
    p, qi = P[users], Q[items]
    if pairwise:
        qj = Q[third]
        y = np.sum(p * qi, axis=1, dtype=dt) - np.sum(p * qj, axis=1, dtype=dt)
        lb, g = pairwise_terms(kind, y)
        l2 = (np.sum(p * p, dtype=dt) + np.sum(qj * qj, dtype=dt) + np.sum(qi * qi, dtype=dt)) / dt(2)
        np.add.at(dP, users, g[:, None] * (qi - qj) + dt(reg) * p)
        np.add.at(dQ, items, g[:, None] * p + dt(reg) * qi)
        np.add.at(dQ, third, -g[:, None] * p + dt(reg) * qj)
        return np.sum(lb, dtype=dt), dt(reg) * l2, dP, dQ
    z = np.asarray(third, dtype=P.dtype)
    x = np.sum(p * qi, axis=1, dtype=dt)
    lb, g = pointwise_terms(kind, z, x)
    scale = dt(1) / dt(len(x)) if kind == "cross_entropy" else dt(1)
    l2 = (np.sum(p * p, dtype=dt) + np.sum(qi * qi, dtype=dt)) / dt(2)
    np.add.at(dP, users, (g * scale)[:, None] * qi + dt(reg) * p)
    np.add.at(dQ, items, (g * scale)[:, None] * p + dt(reg) * qi)
    return np.sum(lb * scale, dtype=dt), dt(reg) * l2, dP, dQ


class RowOptimizer:
    """TF-1.12 sparse application of learner.py:2-16 for gd / adagrad / rmsprop / momentum: only
    the rows in `rows` move (duplicate row gradients already summed in the dense g)."""

    def __init__(self, kind, lr, shape, dtype=np.float32, momentum=0.9):
        self.kind, self.dt = kind, dtype
        self.lr, self.mom = dtype(lr), dtype(momentum)
        self.s0 = {"gd": None, "adagrad": np.full(shape, 1e-8, dtype), "rmsprop": np.ones(shape, dtype),
                   "momentum": np.zeros(shape, dtype)}[kind]
        self.s1 = np.zeros(shape, dtype) if kind == "rmsprop" else None

    def apply(self, var, g, rows):
        rows = np.unique(rows)
        dt, gr = self.dt, g[rows]
        if self.kind == "gd":
            var[rows] -= self.lr * gr
        elif self.kind == "adagrad":
            self.s0[rows] += gr * gr
            var[rows] -= (self.lr * gr) * (dt(1) / np.sqrt(self.s0[rows]))
        elif self.kind == "rmsprop":
            rho, eps = dt(0.9), dt(1e-10)
            self.s0[rows] = self.s0[rows] * rho + (gr * gr) * (dt(1) - rho)
            self.s1[rows] = self.s1[rows] * dt(0.0) + (dt(1) / np.sqrt(self.s0[rows] + eps)) * self.lr * gr
            var[rows] -= self.s1[rows]
        else:
            self.s0[rows] = self.s0[rows] * self.mom + gr
            var[rows] -= self.s0[rows] * self.lr


def mf_step(P, Q, mP, vP, mQ, vQ, users, pos, neg, reg, adam):
    """sess.run((loss, optimizer)) of MF.py:101 — updates the six arrays in place."""
    loss, dP, dQ = mf_loss_and_grads(P, Q, users, pos, neg, reg)
    adam.sparse_swept(P, mP, vP, dP)
    adam.sparse_swept(Q, mQ, vQ, dQ)
    adam.advance()
    return loss


# ------------------------------------------------------------------ LightGCN
def lightgcn_adjacency(user_idx, item_idx, n_users, n_items, adj_type="pre"):
    """LightGCN.create_adj_mat (LightGCN.py:34-78): fp32 CSR on N = U+I nodes."""
    user_np = np.asarray(user_idx, dtype=np.int32)
    item_np = np.asarray(item_idx, dtype=np.int32)
    ratings = np.ones_like(user_np, dtype=np.float32)
    n = n_users + n_items
    tmp = sp.csr_matrix((ratings, (user_np, item_np + n_users)), shape=(n, n))
    adj = tmp + tmp.T

    def single(a):
        rowsum = np.array(a.sum(1))
        with np.errstate(divide="ignore"):
            d_inv = np.power(rowsum, -1).flatten()
        d_inv[np.isinf(d_inv)] = 0.0
        return sp.diags(d_inv).dot(a).tocoo()

    if adj_type == "plain":
        out = adj
    elif adj_type == "norm":
        out = single(adj + sp.eye(adj.shape[0]))
    elif adj_type == "gcmc":
        out = single(adj)
    elif adj_type == "pre":
        rowsum = np.array(adj.sum(1))
        with np.errstate(divide="ignore"):
            d_inv = np.power(rowsum, -0.5).flatten()
        d_inv[np.isinf(d_inv)] = 0.0
        d_mat = sp.diags(d_inv)
        out = d_mat.dot(adj).dot(d_mat)
    else:
        mean_adj = single(adj)
        out = mean_adj + sp.eye(mean_adj.shape[0])
    out = out.tocsr().astype(np.float32)
    out.sort_indices()
    return out


def spmm_rowwise(A, X):
    """Y = A @ X with per-row ascending-column accumulation, product and sum rounded
    separately in X's dtype (what scipy's csr_matvecs does; TF-CPU's
    SparseTensorDenseMatMul walks the nnz in the same order)."""
    return (A.astype(X.dtype) @ X).astype(X.dtype)


def lightgcn_propagate(A, E0, n_layers):
    """_create_lightgcn_embed (LightGCN.py:132-149): returns (E*, [E0..EL])."""
    dt = E0.dtype.type
    layers = [E0]
    ego = E0
    for _ in range(n_layers):
        ego = spmm_rowwise(A, ego)
        layers.append(ego)
    acc = layers[0].copy()
    for e in layers[1:]:
        acc = acc + e
    return (acc / dt(n_layers + 1)).astype(E0.dtype), layers


def lightgcn_loss_and_grad(A, At, E0, n_users, n_layers, users, pos, neg, reg):
    """Loss (LightGCN.py:156-166) and dLoss/dE0 through the propagation."""
    dt = E0.dtype.type
    Estar, _ = lightgcn_propagate(A, E0, n_layers)
    iu, ii, ij = np.asarray(users), n_users + np.asarray(pos), n_users + np.asarray(neg)
    eu, ei, ej = Estar[iu], Estar[ii], Estar[ij]
    x = np.sum(eu * ei, axis=1, dtype=dt) - np.sum(eu * ej, axis=1, dtype=dt)
    lb, g = bpr_terms(x)
    zu, zi, zj = E0[iu], E0[ii], E0[ij]
    regularizer = (np.sum(zu * zu, dtype=dt) + np.sum(zi * zi, dtype=dt) +
                   np.sum(zj * zj, dtype=dt)) / dt(2)
    mf_loss = np.sum(lb, dtype=dt)
    emb_loss = dt(reg) * regularizer
    Gstar = np.zeros_like(E0)
    np.add.at(Gstar, iu, g[:, None] * (ei - ej))
    np.add.at(Gstar, ii, g[:, None] * eu)
    np.add.at(Gstar, ij, -g[:, None] * eu)
    H = Gstar / dt(n_layers + 1)
    G = H
    for _ in range(n_layers):
        G = H + spmm_rowwise(At, G)
    R = np.zeros_like(E0)
    np.add.at(R, iu, dt(reg) * zu)
    np.add.at(R, ii, dt(reg) * zi)
    np.add.at(R, ij, dt(reg) * zj)
    return mf_loss, emb_loss, (G + R).astype(E0.dtype)


def lightgcn_step(A, At, E0, m, v, n_users, n_layers, users, pos, neg, reg, adam):
    """sess.run(self.opt) of LightGCN.py:178 — E0, m, v updated in place."""
    mf_loss, emb_loss, grad = lightgcn_loss_and_grad(A, At, E0, n_users, n_layers, users, pos,
                                                     neg, reg)
    adam.dense(E0, m, v, grad)
    adam.advance()
    return mf_loss, emb_loss


# ------------------------------------------------------------------ sampler structure
def generate_positive_items(user_pos_dict):
    """_generate_positive_items (data/sampler.py:24-39)."""
    users_list, pos_items_list, user_pos_len = [], [], []
    for user, pos_items in user_pos_dict.items():
        user_pos_len.append([user, len(pos_items)])
        users_list.extend([user] * len(pos_items))
        pos_items_list.extend(pos_items)
    return user_pos_len, users_list, pos_items_list


# ------------------------------------------------------------------ NGCF (alg_type=ngcf)
def ngcf_adjacency(train_matrix, adj_type="norm"):
    """NGCF.get_adj_mat (NGCF.py:299-318): A carries the *rating values* of the train matrix
    (`self.graph = dataset.train_matrix.toarray()`, NGCF.py:40); `norm` = D^-1 (A + I),
    computed in fp64 because sp.eye is fp64, rounded to fp32 when handed to TF."""
    R = sp.csr_matrix(train_matrix, dtype=np.float32)
    U, I = R.shape
    A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)

    def single(a):
        rowsum = np.array(a.sum(1))
        with np.errstate(divide="ignore"):
            d_inv = np.power(rowsum, -1).flatten()
        d_inv[np.isinf(d_inv)] = 0.0
        return sp.diags(d_inv).dot(a).tocoo()

    if adj_type == "plain":
        out = A
    elif adj_type == "norm":
        out = single(A + sp.eye(A.shape[0]))
    elif adj_type == "gcmc":
        out = single(A)
    else:
        out = single(A) + sp.eye(A.shape[0])
    out = out.tocsr().astype(np.float32)
    out.sort_indices()
    return out


def _leaky(x, alpha=0.2):
    return np.where(x > 0, x, x * x.dtype.type(alpha))


def ngcf_forward(A, E0, weights, masks, keep):
    """_create_ngcf_embed (NGCF.py:160-202).  weights: list of (W_gc, b_gc, W_bi, b_bi) per layer;
    masks: list of {0,1} arrays [N, d_out] (the dropout draw is an input, TF's Philox stream is
    not reproducible); keep = 1 - mess_dropout_ratio.  Returns (concat output, cache)."""
    dt = E0.dtype.type
    ego = E0
    outs, cache = [E0], []
    for (Wg, bg, Wb, bb), mask in zip(weights, masks):
        S = spmm_rowwise(A, ego)
        T1 = S @ Wg + bg
        Bi = ego * S
        T2 = Bi @ Wb + bb
        Z = _leaky(T1) + _leaky(T2)
        Zd = (Z / dt(keep)) * mask.astype(E0.dtype)          # tf.nn.dropout: x / keep_prob * mask
        ss = np.sum(Zd * Zd, axis=1, keepdims=True, dtype=dt)
        inv = dt(1) / np.sqrt(np.maximum(ss, dt(1e-12)))      # tf.nn.l2_normalize
        norm = Zd * inv
        cache.append((ego, S, T1, T2, Bi, Zd, ss, inv, norm, mask))
        outs.append(norm)
        ego = Zd
    return np.concatenate(outs, axis=1).astype(E0.dtype), cache


def ngcf_loss_and_grads(A, At, E0, weights, masks, keep, n_users, users, pos, neg, reg):
    """Loss (NGCF.py:91-110) and gradients w.r.t. E0 and every layer weight."""
    dt = E0.dtype.type
    out, cache = ngcf_forward(A, E0, weights, masks, keep)
    iu, ii, ij = np.asarray(users), n_users + np.asarray(pos), n_users + np.asarray(neg)
    eu, ei, ej = out[iu], out[ii], out[ij]
    x = np.sum(eu * ei, axis=1, dtype=dt) - np.sum(eu * ej, axis=1, dtype=dt)
    lb, g = bpr_terms(x)
    l2 = (np.sum(eu * eu, dtype=dt) + np.sum(ei * ei, dtype=dt) + np.sum(ej * ej, dtype=dt)) / dt(2)
    loss = np.sum(lb, dtype=dt) + dt(reg) * l2
    dOut = np.zeros_like(out)
    np.add.at(dOut, iu, g[:, None] * (ei - ej) + dt(reg) * eu)
    np.add.at(dOut, ii, g[:, None] * eu + dt(reg) * ei)
    np.add.at(dOut, ij, -g[:, None] * eu + dt(reg) * ej)
    d0 = E0.shape[1]
    widths = [w[0].shape[1] for w in weights]
    offs = np.cumsum([d0] + widths)
    dEgo = np.zeros((E0.shape[0], widths[-1]), E0.dtype) if weights else None
    wgrads = [None] * len(weights)
    for k in range(len(weights) - 1, -1, -1):
        ego, S, T1, T2, Bi, Zd, ss, inv, norm, mask = cache[k]
        Wg, bg, Wb, bb = weights[k]
        dNorm = dOut[:, offs[k]:offs[k + 1]]
        dot = np.sum(dNorm * norm, axis=1, keepdims=True, dtype=dt)
        dZd = np.where(ss > dt(1e-12), (dNorm - norm * dot) * inv, dNorm * inv) + dEgo
        dZ = dZd * mask.astype(E0.dtype) / dt(keep)
        dT1 = dZ * np.where(T1 > 0, dt(1), dt(0.2))
        dT2 = dZ * np.where(T2 > 0, dt(1), dt(0.2))
        wgrads[k] = (S.T @ dT1, dT1.sum(0, keepdims=True), Bi.T @ dT2, dT2.sum(0, keepdims=True))
        dBi = dT2 @ Wb.T
        dS = dT1 @ Wg.T + dBi * ego
        dEgo = dBi * S + spmm_rowwise(At, dS.astype(E0.dtype))
    dE0 = dOut[:, :d0] + (dEgo if weights else 0)
    return loss, dE0.astype(E0.dtype), wgrads


# ------------------------------------------------------------------ MultiVAE
def _act(name, x):
    if name == "tanh":
        return np.tanh(x)
    if name == "sigmoid":
        return 1 / (1 + np.exp(-x))
    if name == "relu":
        return np.maximum(x, 0)
    if name == "identity":
        return x
    raise NotImplementedError(name)


def _act_grad(name, y, x):
    """derivative w.r.t. the pre-activation x, given the output y"""
    if name == "tanh":
        return 1 - y * y
    if name == "sigmoid":
        return y * (1 - y)
    if name == "relu":
        return (x > 0).astype(x.dtype)
    return np.ones_like(x)


def multivae_forward(X, Wq, bq, Wp, bp, drop_mask, keep, eps, is_training, act="tanh"):
    """q_graph / p_graph / _create_inference (MultiVAE.py:73-124).  X: dense multi-hot [B, I];
    Wq = [W_q0 [I,h], W_q1 [h, 2*z]], Wp = [W_p0 [z,h], W_p1 [h, I]] (two layers each, the shape
    conf/MultiVAE.properties p_dim=[z,h] gives); drop_mask [B, I] in {0,1}; eps [B, z] ~ N(0, 0.01²)
    (both random draws are inputs).  Returns logits, log_softmax, KL, cache."""
    dt = X.dtype.type
    ss = np.sum(X * X, axis=1, keepdims=True, dtype=dt)
    h0 = X / np.sqrt(np.maximum(ss, dt(1e-12)))
    h0 = h0 / dt(keep) * drop_mask.astype(X.dtype)
    a1 = h0 @ Wq[0] + bq[0]
    h1 = _act(act, a1)
    h2 = h1 @ Wq[1] + bq[1]
    z = h2.shape[1] // 2
    mu, logvar = h2[:, :z], h2[:, z:]
    std = np.exp(dt(0.5) * logvar)
    KL = np.mean(np.sum(dt(0.5) * (-logvar + np.exp(logvar) + mu * mu - dt(1)), axis=1, dtype=dt), dtype=dt)
    zs = mu + dt(is_training) * eps * std
    a3 = zs @ Wp[0] + bp[0]
    g1 = _act(act, a3)
    logits = g1 @ Wp[1] + bp[1]
    mx = logits.max(axis=1, keepdims=True)
    lse = mx + np.log(np.sum(np.exp(logits - mx), axis=1, keepdims=True, dtype=dt))
    logsm = logits - lse
    return logits, logsm, KL, (h0, a1, h1, mu, logvar, std, zs, a3, g1)


def multivae_loss_and_grads(X, Wq, bq, Wp, bp, drop_mask, keep, eps, anneal, reg, act="tanh"):
    """neg-ELBO of MultiVAE.py:126-135 and its gradients (all dense, as TF produces them)."""
    dt = X.dtype.type
    B = X.shape[0]
    logits, logsm, KL, (h0, a1, h1, mu, logvar, std, zs, a3, g1) = multivae_forward(
        X, Wq, bq, Wp, bp, drop_mask, keep, eps, 1.0, act)
    neg_ll = -np.mean(np.sum(logsm * X, axis=1, dtype=dt), dtype=dt)
    reg_var = dt(reg) * sum(np.sum(w * w, dtype=dt) / dt(2) for w in list(Wq) + list(Wp))
    loss = neg_ll + dt(anneal) * KL + dt(2) * reg_var
    n = np.sum(X, axis=1, keepdims=True, dtype=dt)
    dlogits = (np.exp(logsm) * n - X) / dt(B)
    gWp1 = g1.T @ dlogits + dt(2 * reg) * Wp[1]
    gbp1 = dlogits.sum(0)
    da3 = (dlogits @ Wp[1].T) * _act_grad(act, g1, a3)
    gWp0 = zs.T @ da3 + dt(2 * reg) * Wp[0]
    gbp0 = da3.sum(0)
    dz = da3 @ Wp[0].T
    dmu = dz + dt(anneal) * mu / dt(B)
    dlogvar = dz * eps * std * dt(0.5) + dt(anneal) * dt(0.5) * (np.exp(logvar) - dt(1)) / dt(B)
    dh2 = np.concatenate([dmu, dlogvar], axis=1)
    gWq1 = h1.T @ dh2 + dt(2 * reg) * Wq[1]
    gbq1 = dh2.sum(0)
    da1 = (dh2 @ Wq[1].T) * _act_grad(act, h1, a1)
    gWq0 = h0.T @ da1 + dt(2 * reg) * Wq[0]
    gbq0 = da1.sum(0)
    return loss, ([gWq0, gWq1], [gbq0, gbq1], [gWp0, gWp1], [gbp0, gbp1]), (neg_ll, KL)


def multivae_general(X, Wq, bq, Wp, bp, drop_mask, keep, eps, anneal, reg, act="tanh", is_training=1.0,
                     want_grads=True):
    """q_graph / p_graph / neg-ELBO of MultiVAE.py:73-135 for ANY p_dim (the reference builds
    q_dims = reversed(p_dim + [I]) layers of arbitrary number and width, :34-36,46-71): Wq / bq the n encoder layers
    (the last with 2z columns: [mu | logvar]), Wp / bp the n decoder layers (the last onto the I items).  Returns
    (loss, (gWq, gbq, gWp, gbp), (neg_ll, KL), logits).  Pinned to the reference class by
    tests/golden/tfgraph_multivae_wide_*.npz (tests/test_tfgraph_golden.py)."""
    dt = X.dtype.type
    B, n = X.shape[0], len(Wq)
    ss = np.sum(X * X, axis=1, keepdims=True, dtype=dt)
    h = X / np.sqrt(np.maximum(ss, dt(1e-12)))
    h = h / dt(keep) * drop_mask.astype(X.dtype)
    q_in, q_out = [], []
    for i in range(n):
        q_in.append(h)
        a = h @ Wq[i] + bq[i]
        if i != n - 1:
            h = _act(act, a)
            q_out.append(h)
    z = a.shape[1] // 2
    mu, logvar = a[:, :z], a[:, z:]
    std = np.exp(dt(0.5) * logvar)
    KL = np.mean(np.sum(dt(0.5) * (-logvar + np.exp(logvar) + mu * mu - dt(1)), axis=1, dtype=dt), dtype=dt)
    zs = mu + dt(is_training) * eps * std
    g = zs
    p_in, p_out = [], []
    for i in range(n):
        p_in.append(g)
        a3 = g @ Wp[i] + bp[i]
        if i != n - 1:
            g = _act(act, a3)
            p_out.append(g)
    logits = a3
    mx = logits.max(axis=1, keepdims=True)
    lse = mx + np.log(np.sum(np.exp(logits - mx), axis=1, keepdims=True, dtype=dt))
    logsm = logits - lse
    neg_ll = -np.mean(np.sum(logsm * X, axis=1, dtype=dt), dtype=dt)
    reg_var = dt(reg) * sum(np.sum(w * w, dtype=dt) / dt(2) for w in list(Wq) + list(Wp))
    loss = neg_ll + dt(anneal) * KL + dt(2) * reg_var
    if not want_grads:
        return loss, None, (neg_ll, KL), logits
    nrow = np.sum(X, axis=1, keepdims=True, dtype=dt)
    d = (np.exp(logsm) * nrow - X) / dt(B)
    gWp, gbp = [None] * n, [None] * n
    for i in range(n - 1, -1, -1):
        gWp[i] = p_in[i].T @ d + dt(2 * reg) * Wp[i]
        gbp[i] = d.sum(0)
        d = d @ Wp[i].T
        if i > 0:
            d = d * _act_grad(act, p_out[i - 1], p_out[i - 1])        # (relu: y > 0 iff x > 0)
    dz = d
    dmu = dz + dt(anneal) * mu / dt(B)
    dlogvar = dz * eps * std * dt(0.5) * dt(is_training) + dt(anneal) * dt(0.5) * (np.exp(logvar) - dt(1)) / dt(B)
    d = np.concatenate([dmu, dlogvar], axis=1)
    gWq, gbq = [None] * n, [None] * n
    for i in range(n - 1, -1, -1):
        gWq[i] = q_in[i].T @ d + dt(2 * reg) * Wq[i]
        gbq[i] = d.sum(0)
        if i > 0:
            d = (d @ Wq[i].T) * _act_grad(act, q_out[i - 1], q_out[i - 1])
    return loss, (gWq, gbq, gWp, gbp), (neg_ll, KL), logits
