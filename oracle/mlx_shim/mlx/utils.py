"""mlx.utils stand-in (TEST INFRASTRUCTURE): the tree helpers of MLX, restated from their documentation."""
from __future__ import annotations


def tree_map(fn, tree, *rest, is_leaf=None):
    if is_leaf is not None and is_leaf(tree):
        return fn(tree, *rest)
    if isinstance(tree, (list, tuple)):
        T = type(tree)
        return T(tree_map(fn, c, *(r[i] for r in rest), is_leaf=is_leaf) for i, c in enumerate(tree))
    if isinstance(tree, dict):
        return {k: tree_map(fn, c, *(r[k] for r in rest), is_leaf=is_leaf) for k, c in tree.items()}
    return fn(tree, *rest)


def tree_flatten(tree, prefix="", is_leaf=None, destination=None):
    out = []

    def rec(t, p):
        if is_leaf is not None and is_leaf(t):
            out.append((p, t))
        elif isinstance(t, (list, tuple)):
            for i, c in enumerate(t):
                rec(c, f"{p}.{i}" if p else str(i))
        elif isinstance(t, dict):
            for k, c in t.items():
                rec(c, f"{p}.{k}" if p else str(k))
        else:
            out.append((p, t))

    rec(tree, prefix.lstrip("."))
    return out


def tree_unflatten(items):
    if isinstance(items, dict):
        items = list(items.items())
    root = {}
    for key, v in items:
        parts = key.split(".")
        d = root
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v

    def fix(d):
        if isinstance(d, dict):
            d = {k: fix(v) for k, v in d.items()}
            if d and all(k.isdigit() for k in d):
                return [d[str(i)] for i in range(len(d))]
        return d

    return fix(root)


def tree_reduce(fn, tree, initializer=None, is_leaf=None):
    acc = initializer
    for _, leaf in tree_flatten(tree, is_leaf=is_leaf):
        acc = leaf if acc is None else fn(acc, leaf)
    return acc
