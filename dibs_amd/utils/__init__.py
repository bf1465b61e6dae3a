from .tree import tree_leaves, tree_unflatten_like, tree_select, tree_mul, tree_shapes  # noqa: F401
