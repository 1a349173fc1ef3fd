"""oracle.train_torch — the LightGCN step of oracle.train restated with multi-threaded torch-CPU ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): this is the `nproc`-thread CPU baseline leg of
bench.py (SURVEY §8d "(ii) ... scipy CSR SpMM (1 thread) *and* torch-CPU (torch.set_num_threads
(nproc)), report the faster").  Same graph as LightGCN.py:132-166 + TF-1.12 dense ApplyAdam; fp32;
checked against oracle.train.lightgcn_step in tests/test_oracle_cpu.py.
"""
import numpy as np
import torch


class TorchLightGCN:
    def __init__(self, A_csr, E0, n_users, n_layers, lr, reg, threads, dtype=np.float32):
        """dtype=np.float64: the fp64 twin of the same graph (the error bar of long parity runs, where the
        one-thread scipy restatement would take minutes)"""
        torch.set_num_threads(int(threads))
        a = A_csr.tocsr().astype(dtype)
        a.sort_indices()
        self.A = torch.sparse_csr_tensor(torch.from_numpy(a.indptr.astype(np.int64)),
                                         torch.from_numpy(a.indices.astype(np.int64)),
                                         torch.from_numpy(a.data), size=a.shape)
        self.E = torch.from_numpy(np.array(E0, dtype=dtype))
        self.m, self.v = torch.zeros_like(self.E), torch.zeros_like(self.E)
        self.U, self.L, self.lr, self.reg = int(n_users), int(n_layers), float(lr), float(reg)
        self.b1, self.b2, self.eps = 0.9, 0.999, 1e-8
        self.b1p, self.b2p = np.float32(0.9), np.float32(0.999)

    def step(self, users, pos, neg):
        """One sess.run(opt) (LightGCN.py:178); `A` symmetric ('pre').  Returns (mf_loss, emb_loss)."""
        E, L, U = self.E, self.L, self.U
        iu = torch.as_tensor(np.asarray(users, np.int64))
        ii = torch.as_tensor(np.asarray(pos, np.int64)) + U
        ij = torch.as_tensor(np.asarray(neg, np.int64)) + U
        acc, ego = E.clone(), E
        for _ in range(L):
            ego = torch.sparse.mm(self.A, ego)
            acc += ego
        Es = acc / float(L + 1)
        eu, ei, ej = Es[iu], Es[ii], Es[ij]
        x = (eu * ei).sum(1) - (eu * ej).sum(1)
        mf_loss = torch.nn.functional.softplus(-x).sum()
        g = -torch.sigmoid(-x)
        zu, zi, zj = E[iu], E[ii], E[ij]
        emb_loss = self.reg * 0.5 * ((zu * zu).sum() + (zi * zi).sum() + (zj * zj).sum())
        Gs = torch.zeros_like(E)
        Gs.index_add_(0, iu, g[:, None] * (ei - ej))
        Gs.index_add_(0, ii, g[:, None] * eu)
        Gs.index_add_(0, ij, -g[:, None] * eu)
        H = Gs / float(L + 1)
        G = H
        for _ in range(L):
            G = H + torch.sparse.mm(self.A, G)
        G.index_add_(0, iu, self.reg * zu)
        G.index_add_(0, ii, self.reg * zi)
        G.index_add_(0, ij, self.reg * zj)
        alpha = float(np.float32(self.lr) * np.sqrt(np.float32(1) - self.b2p) / (np.float32(1) - self.b1p))
        self.m += (G - self.m) * (1.0 - self.b1)
        self.v += (G * G - self.v) * (1.0 - self.b2)
        E -= (self.m * alpha) / (self.v.sqrt() + self.eps)
        self.b1p = np.float32(self.b1p * np.float32(self.b1))
        self.b2p = np.float32(self.b2p * np.float32(self.b2))
        return float(mf_loss), float(emb_loss)
