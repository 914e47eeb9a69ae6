"""Symbol allocator and small validation helpers for the TN front-end.

Host-side bookkeeping only (never on the device path).  Behaviour follows the
reference so that ``TN.einsum_str`` text is identical:

* symbol table      - reference contractn/utils.py:66-81 (inverse of
  ``opt_einsum.get_symbol``: a-z, A-Z, then ``chr(idx + 140)``)
* gap-filling rule  - reference contractn/utils.py:44-63
* error factories   - reference contractn/utils.py:84-116
"""
from functools import lru_cache

_ASCII_SYMBOLS = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
_ASCII_INDEX = {c: i for i, c in enumerate(_ASCII_SYMBOLS)}
_UNICODE_SHIFT = 140


def get_symbol(idx):
    """Symbol number ``idx`` of the einsum alphabet (52 ASCII letters, then unicode)."""
    if idx < 0:
        raise ValueError("symbol index must be non-negative")
    if idx < len(_ASCII_SYMBOLS):
        return _ASCII_SYMBOLS[idx]
    return chr(idx + _UNICODE_SHIFT)


@lru_cache(maxsize=None)
def symbol_idx(symbol):
    """Inverse of :func:`get_symbol`."""
    assert_valid_symbol(symbol)
    idx = _ASCII_INDEX.get(symbol)
    if idx is None:
        idx = ord(symbol) - _UNICODE_SHIFT
    assert idx >= 0 and get_symbol(idx) == symbol
    return idx


def get_new_symbols(old_symbols, num_new):
    """``num_new`` unused symbols: holes below the current maximum first, then fresh ones."""
    used = {symbol_idx(s) for s in old_symbols}
    assert len(used) == len(old_symbols)
    picked = []
    top = max(used) if used else -1
    cand = 0
    while len(picked) < num_new and cand < top:
        if cand not in used:
            picked.append(cand)
        cand += 1
    nxt = top + 1
    while len(picked) < num_new:
        picked.append(nxt)
        nxt += 1
    return tuple(get_symbol(i) for i in picked)


def assert_valid_tensor(tensor):
    assert hasattr(tensor, "ndim")
    assert hasattr(tensor, "shape")


def assert_valid_symbol(symbol):
    assert isinstance(symbol, str)
    assert len(symbol) == 1


def opposite_node(edge_id, node):
    """The other endpoint of a networkx edge key ``(u, v, k)``."""
    u, v = edge_id[0], edge_id[1]
    assert node in (u, v)
    return v if node == u else u


def _canon_edges(edges):
    out = []
    for e in edges:
        head = tuple(sorted(e[:2]))
        out.append(head + tuple(e[2:]))
    return sorted(out)


def edge_set_equality(edgeset1, edgeset2):
    """Undirected comparison of two collections of networkx edge ids."""
    return _canon_edges(set(edgeset1)) == _canon_edges(set(edgeset2))


_FULL_NODE_NAMES = {"dense": "dense", "clone": "duplicate", "hyper": "copy", "input": "input"}


def node_attr_error(owner_type, attr_name, node_name, actual_type):
    """ValueError raised when a node-type specific attribute is read on the wrong node type."""
    return ValueError(
        f"Only {_FULL_NODE_NAMES[owner_type]} nodes have {attr_name} attributes "
        f"(node '{node_name}' has node type '{actual_type}')"
    )
