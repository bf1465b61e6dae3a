"""Block algebra of k_bge_soft_mf (kernels_bge_soft_mf.h) in numpy, float64: the upper-block (U-form) right-looking factorisation of
W + I with node j ordered last, the block rows of T = L^-1 from products of the forms X^T Y (register form) and T_jj X (diagonal
inverse as the left operand), and the quantities the epilogue takes from them -- against numpy's inverse of the same matrices.
Run: python tests/tools/bge_soft_blocked_emulation.py"""
import numpy as np

rng = np.random.default_rng(0)


def blocked(Wm, NB):
    """Wm = A - I (n x n, n = 16 NB).  Returns pivots - 1, T = L^-1 (dense, lower)."""
    B = 16
    blk = lambda X, i, j: X[B * i:B * i + B, B * j:B * j + B]
    acc = {(i, j): blk(Wm, i, j).copy() for i in range(NB) for j in range(i, NB)}   # upper blocks
    U, T, dm1 = {}, {}, np.zeros(16 * NB)
    for k in range(NB):
        S = acc[(k, k)].copy()                       # symmetric, diagonal stored minus one
        L = np.zeros((B, B))
        for t in range(B):
            pivm1 = S[t, t]; piv = 1 + pivm1; inv = 1 / np.sqrt(piv)
            dm1[B * k + t] = pivm1
            l = np.where(np.arange(B) > t, S[:, t] * inv, 0.0); l[t] = piv * inv
            L[:, t] = l
            for c in range(t + 1, B):
                S[:, c] -= l * l[c]
        Tkk = np.zeros((B, B))
        for q in range(B - 1, -1, -1):
            a = (np.arange(B) == q).astype(float)
            for m in range(q + 1, B):
                a -= Tkk[:, m] * L[m, q]
            Tkk[:, q] = a / L[q, q]
        T[(k, k)] = Tkk
        for i in range(k + 1, NB):
            U[(k, i)] = Tkk @ acc[(k, i)]            # U_ki = L_kk^-1 A_ki   (A operand: T_kk rows; B operand: the block in C layout)
        for i in range(k + 1, NB):
            for j in range(i, NB):
                acc[(i, j)] -= U[(k, i)].T @ U[(k, j)]   # register form X^T Y
        for i in range(k):                           # block row k of T
            Sacc = sum(U[(l, k)].T @ T[(l, i)] for l in range(i, k))
            T[(k, i)] = -Tkk @ Sacc
    Td = np.zeros((B * NB, B * NB))
    for (j, i), v in T.items():
        Td[B * j:B * j + B, B * i:B * i + B] = v
    return dm1, Td


for d in (5, 16, 17, 33, 50, 64):
    NB = (d + 15) // 16
    n = 16 * NB
    X = rng.normal(size=(100, d)); R = X.T @ X / 10 + np.eye(d)
    for j in (0, d // 2, d - 1):
        p = rng.uniform(0, 1, d); p[j] = 0
        var = np.arange(d); var[j], var[d - 1] = d - 1, j          # variable at each position: j last
        pp = p[var].copy(); pp[d - 1] = 1.0
        Wm = np.zeros((n, n))
        Wm[:d, :d] = np.outer(pp, pp) * (R[np.ix_(var, var)] - np.eye(d))
        dm1, T = blocked(Wm, NB)
        # reference quantities
        Dm = np.diag(p); Mpa = np.eye(d) + Dm @ (R - np.eye(d)) @ Dm
        b = p * R[j]
        Minv = np.linalg.inv(Mpa)
        y_ref = Minv @ b; s_ref = R[j, j] - b @ y_ref
        ld_ref = np.linalg.slogdet(Mpa)[1]
        # from the blocked factorisation
        s = 1 + dm1[d - 1]
        ld = np.log(1 + dm1[:d - 1]).sum()
        rows = np.arange(n)[:, None]; cols = np.arange(n)[None, :]
        offd = ((T ** 2) * ((rows > cols) & (rows != d - 1))).sum(axis=0)
        one_minus = dm1 / (1 + dm1) - offd                     # 1 - (M_pa^-1)_cc by position
        y_pos = -T[d - 1] * np.sqrt(s)
        e = []
        e.append(abs(s - s_ref) / s_ref); e.append(abs(ld - ld_ref))
        for ps in range(d - 1):
            v = var[ps]
            e.append(abs(one_minus[ps] - (1 - Minv[v, v])))
            e.append(abs(y_pos[ps] - y_ref[v]))
        print(f"d={d} j={j}: max err {max(e):.2e}")
        assert max(e) < 1e-9
print("ok")
