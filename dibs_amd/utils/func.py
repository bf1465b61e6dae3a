"""numpy counterparts of the small array helpers of the reference (dibs/utils/func.py:8-18, 100-125).  The device path has its
own closed forms; these exist so that user code importing them keeps working.  (`sel`, `leftsel`, `mask_topk`,
func.py:21-97, are unused in the reference and not provided.)"""
import numpy as np

from .tree import tree_leaves


def expand_by(arr, n):
    """append n singleton axes (func.py:8-18)"""
    arr = np.asarray(arr)
    return arr.reshape(arr.shape + (1,) * n)


def squared_norm_pytree(x, y):
    """sum over all leaves of ||x_leaf - y_leaf||^2 (func.py:100-114)"""
    return float(sum(np.sum((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)
                     for a, b in zip(tree_leaves(x), tree_leaves(y))))


def zero_diagonal(g):
    """zero the diagonal of the trailing [d, d] axes (func.py:117-125)"""
    g = np.array(g, copy=True)
    d = g.shape[-1]
    g[..., np.arange(d), np.arange(d)] = 0
    return g
