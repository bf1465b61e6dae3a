"""Graph helpers on adjacency matrices (numpy, host side).  Mirrors dibs/graph_utils.py of the reference
without igraph: a graph is its adjacency matrix (``g[i, j] = 1`` iff ``i -> j``)."""
import numpy as np


def acyclic_constr_nograd(mat, n_vars):
    """h(G) = tr((I + G/d)^d) - d in float32 by binary powering (reference: dibs/graph_utils.py:8-28).
    Evaluated in float32 on purpose: the metrics' DAG filter ``h == 0`` (dibs/metrics.py:72) inherits the
    float32 resolution of the reference (long cycles fall below it)."""
    m = np.eye(n_vars, dtype=np.float32) + np.float32(1.0 / n_vars) * np.asarray(mat, np.float32)
    return np.float32(np.trace(np.linalg.matrix_power(m, n_vars)) - np.float32(n_vars))


def elwise_acyclic_constr_nograd(mats, n_vars):
    return np.array([acyclic_constr_nograd(m, n_vars) for m in np.asarray(mats)], dtype=np.float32)


def mat_is_dag(mat):
    """Exact test by repeatedly peeling sources (Kahn)."""
    a = (np.asarray(mat) != 0).astype(np.int64)
    alive = np.ones(a.shape[0], bool)
    while alive.any():
        indeg = a[alive][:, alive].sum(0)
        src = np.where(indeg == 0)[0]
        if src.size == 0:
            return False
        idx = np.where(alive)[0][src]
        alive[idx] = False
    return True


def topological_order(mat):
    a = (np.asarray(mat) != 0).astype(np.int64)
    d = a.shape[0]
    indeg = a.sum(0)
    order, ready = [], [j for j in range(d) if indeg[j] == 0]
    while ready:
        j = ready.pop(0)
        order.append(j)
        for c in np.where(a[j])[0]:
            indeg[c] -= 1
            if indeg[c] == 0:
                ready.append(int(c))
    if len(order) != d:
        raise ValueError("graph has a cycle")
    return order


def graph_to_mat(g):
    return np.asarray(g)


def mat_to_graph(mat):
    return np.asarray(mat)


def adjmat_to_str(mat, max_len=40):
    mat = np.asarray(mat)
    parts, seen = [], set()
    for u, v in zip(*np.where(mat == 1)):
        if mat[v, u] == 1:
            if (u, v) not in seen:
                seen.add((v, u))
                parts.append(f"{u}--{v}")
        else:
            parts.append(f"{u}->{v}")
    s = "  ".join(parts)
    if len(s) > max_len:
        return s[:max_len] + " ... "
    return s or "<empty graph>"
