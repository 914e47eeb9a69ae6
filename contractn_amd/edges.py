"""Edge handle of the TN front-end (host bookkeeping, not on the device path).

Mirrors the public surface of reference contractn/edges.py:4-50 (``name``,
``nodes``, ``symbol``, ``dim``, ``var_dim``, ``dangler``, ``dict``); the edge
metadata lives in the networkx edge attribute dict under the same keys
(``"symbol"``, ``"dim"``, ``"tn_edge"``) so graph-level introspection written
against the reference keeps working.
"""
from .utils import assert_valid_symbol


class Edge:
    __slots__ = ("tn", "name")

    def __init__(self, parent_tn, nx_id, dim, symbol):
        assert isinstance(nx_id, tuple) and len(nx_id) == 3
        assert isinstance(dim, int)
        assert_valid_symbol(symbol)
        self.tn = parent_tn
        self.name = nx_id
        attrs = parent_tn.G.edges[nx_id]
        attrs["dim"] = dim
        attrs["symbol"] = symbol
        attrs["tn_edge"] = self
        parent_tn._symbol_use[symbol] += 1

    @property
    def G(self):
        return self.tn.G

    @property
    def dict(self):
        return self.tn.G.edges[self.name]

    @property
    def nodes(self):
        nodes = self.tn.G.nodes
        return (nodes[self.name[0]]["tn_node"], nodes[self.name[1]]["tn_node"])

    @property
    def symbol(self):
        return self.dict["symbol"]

    @property
    def dim(self):
        return self.dict["dim"]

    @property
    def var_dim(self):
        return self.dim < 0

    @property
    def dangler(self):
        return any(n.dangler for n in self.nodes)

    def __repr__(self):
        return f"Edge(name={self.name}, symbol={self.symbol!r}, dim={self.dim})"
