"""The algebra behind csrc/lagg.hip, in float64 numpy (no GPU): the edge-attention aggregation of one molecule and view
(reference layers.py:82-92 with the masks of layers.py:294-304)

    U[i,j]  = sigma(w[type(i,j)]) adj[i,j] + sigma(self_r) m_i [i == j] + 1e-9 [no bond (i,j)]      over the N_pad columns
    A^[i,j] = m_i U[i,j] / sum_j' U[i,j'] ,      Y'[i,:] = sum_j A^[i,j] P[j,:]

evaluated as a gather over the bonds plus ONE rank-one term per molecule, its transpose, and the closed-form gradients of the attention
parameters -- exactly the three formulas the kernel's row records implement:

    forward      y_i  = sc_i ( sum_e (s_e - 1e-9) P[src_e] + r m_i P[i] + 1e-9 S ) ,            S = sum_{j < nat} P[j]
    transposed   dP_j = sum_e s'_src (s_e - 1e-9) Z[src_e] + r s'_j Z[j] + 1e-9 G ,              G = sum_i s'_i Z[i] ,  s'_i = m_i / rowsum_i
    parameters   d w[c] = sum_{bonds (i,j) of type c} s'_i s_ij (1 - s_ij) ( <Z_i, P_j> - <Z_i, Y'_i> ) ,   d self_r likewise on the diagonal

against the dense operator and against a float64 finite-difference of it."""
import numpy as np
import pytest


def _molecule(rng, n, n_pad, n_types, extra_bonds):
    adj = np.zeros((n, n), dtype=bool)
    for i in range(1, n):                                  # a random tree, then a few ring closures
        j = rng.integers(0, i)
        adj[i, j] = adj[j, i] = True
    for _ in range(extra_bonds):
        i, j = rng.integers(0, n, 2)
        if i != j:
            adj[i, j] = adj[j, i] = True
    btype = np.zeros((n, n), dtype=int)
    t = rng.integers(1, n_types + 1, (n, n))
    t = np.triu(t, 1)
    t = t + t.T                                            # symmetric bond types
    btype[adj] = t[adj]
    return adj, btype


def _dense(adj, btype, w, self_r, n_pad):
    n = adj.shape[0]
    sig = 1.0 / (1.0 + np.exp(-w))
    r = 1.0 / (1.0 + np.exp(-self_r))
    m = (adj.sum(1) > 0).astype(float)                    # row mask: atoms with a bond (layers.py:294-304)
    U = np.where(adj, sig[btype], 1e-9)                   # filler on every non-bond column of the molecule ...
    U = U + np.diag(r * m)
    rowsum = U.sum(1) + 1e-9 * (n_pad - n)                # ... and on the padding columns (their features are zero)
    A = (m / rowsum)[:, None] * U
    return A, U, rowsum, m, sig, r


@pytest.mark.parametrize('n,n_pad,extra', [(19, 132, 2), (64, 64, 30), (7, 40, 0), (256, 256, 200)])
def test_list_form_equals_the_dense_operator(n, n_pad, extra):
    rng = np.random.default_rng(n)
    adj, btype = _molecule(rng, n, n_pad, 4, extra)
    if n == 7:
        adj[3, :] = adj[:, 3] = False                      # an isolated atom inside the molecule
        btype[3, :] = btype[:, 3] = 0
    w, self_r = rng.normal(size=5), 0.3
    F = 12
    P, Z = rng.normal(size=(n, F)), rng.normal(size=(n, F))
    A, U, rowsum, m, sig, r = _dense(adj, btype, w, self_r, n_pad)
    Y = A @ P
    # ---- forward: gather over the bonds + self + rank-one filler ------------------------------------------------------------------
    S = P.sum(0)
    y = np.zeros_like(P)
    sc = m / rowsum
    for i in range(n):
        acc = r * m[i] * P[i] + 1e-9 * S
        for j in np.nonzero(adj[i])[0]:
            acc = acc + (sig[btype[i, j]] - 1e-9) * P[j]
        y[i] = sc[i] * acc
    assert np.abs(y - Y).max() <= 1e-13 * max(1.0, np.abs(Y).max())
    # ---- transposed ----------------------------------------------------------------------------------------------------------------
    G = (sc[:, None] * Z).sum(0)
    dP = np.zeros_like(P)
    for j in range(n):
        acc = r * sc[j] * m[j] * Z[j] + 1e-9 * G           # (U's diagonal carries r m_j: the weight is s'_j r m_j)
        for i in np.nonzero(adj[:, j])[0]:
            acc = acc + sc[i] * (sig[btype[i, j]] - 1e-9) * Z[i]
        dP[j] = acc
    assert np.abs(dP - A.T @ Z).max() <= 1e-13 * max(1.0, np.abs(A.T @ Z).max())
    # ---- parameter gradients: closed form against the dense chain rule -------------------------------------------------------------
    rowdot = (Z * Y).sum(1)                                # <Z_i, Y'_i> = sum_l dA^[i,l] A^[i,l]
    dw = np.zeros_like(w)
    dr = 0.0
    for i in range(n):
        for j in np.nonzero(adj[i])[0]:
            s = sig[btype[i, j]]
            dw[btype[i, j]] += sc[i] * s * (1.0 - s) * (Z[i] @ P[j] - rowdot[i])
        dr += sc[i] * m[i] * (Z[i] @ P[i] - rowdot[i])
    dr *= r * (1.0 - r)

    def loss(w_, self_r_):
        A_, *_ = _dense(adj, btype, w_, self_r_, n_pad)
        return ((A_ @ P) * Z).sum()
    h = 1e-6
    for c in range(1, 5):
        e = np.zeros_like(w)
        e[c] = h
        fd = (loss(w + e, self_r) - loss(w - e, self_r)) / (2 * h)
        assert abs(fd - dw[c]) <= 1e-6 * max(1.0, abs(dw).max()), (c, fd, dw[c])
    fd = (loss(w, self_r + h) - loss(w, self_r - h)) / (2 * h)
    assert abs(fd - dr) <= 1e-6 * max(1.0, abs(dr))


def test_filler_as_one_rank_one_term_is_what_fp32_cannot_do_term_by_term():
    """Why the list form is also the more ACCURATE one at large molecules: a row's 250 filler terms of 1e-9 summed one by one in fp32
    next to O(1) bond weights are lost entirely; as one term 1e-9 * S they are kept to fp32 rounding."""
    rng = np.random.default_rng(5)
    n = 256
    P = rng.normal(size=n).astype(np.float32) + np.float32(3.0)     # one column, mean 3: S = 768
    bonds = np.array([0.41, 0.37, 0.52], dtype=np.float32)
    acc = np.float32(0.0)
    for b, j in zip(bonds, (3, 9, 27)):
        acc = np.float32(acc + b * P[j])
    one_by_one = acc
    for j in range(n):
        one_by_one = np.float32(one_by_one + np.float32(1e-9) * P[j])
    rank_one = np.float32(acc + np.float32(1e-9) * P.sum(dtype=np.float32))
    exact = float(acc) + 1e-9 * float(P.astype(np.float64).sum())
    assert abs(float(rank_one) - exact) < abs(float(one_by_one) - exact) or float(one_by_one) == float(acc)
    assert abs(float(rank_one) - exact) <= 2e-7 * abs(exact)
