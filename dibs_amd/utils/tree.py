"""Minimal pytree helpers (nested lists / tuples of numpy arrays, flattened depth-first like jax).
The engine stores theta as one flat float32 row per particle; these convert to and from the nested
structure the reference returns (e.g. stax parameters ``[(W1, b1), (), (W2, b2)]``)."""
import numpy as np


def tree_leaves(t):
    if t is None:
        return []
    if isinstance(t, (list, tuple)):
        out = []
        for c in t:
            out += tree_leaves(c)
        return out
    return [np.asarray(t)]


def tree_map(f, t):
    if isinstance(t, list):
        return [tree_map(f, c) for c in t]
    if isinstance(t, tuple):
        return tuple(tree_map(f, c) for c in t)
    return f(t)


def tree_unflatten_like(template, leaves):
    it = iter(leaves)

    def go(t):
        if isinstance(t, list):
            return [go(c) for c in t]
        if isinstance(t, tuple):
            return tuple(go(c) for c in t)
        return next(it)
    return go(template)


def tree_select(t, mask):
    return tree_map(lambda a: np.asarray(a)[mask], t)


def tree_mul(t, c):
    return tree_map(lambda a: np.asarray(a) * c, t)


def tree_shapes(t):
    return tree_map(lambda a: np.asarray(a).shape, t)
