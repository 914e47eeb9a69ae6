"""Edge handle of the TN front-end (host bookkeeping, not on the device path).

An ``Edge`` is a light view onto one networkx multigraph edge ``(u, v, key)``.
All state lives in the graph's edge-attribute dict under the keys the
reference uses (``"symbol"``, ``"dim"``, ``"tn_edge"``; reference
contractn/edges.py:25-30), so code that inspects ``tn.G`` directly keeps
working; the handle itself only stores the owning network and the edge id.
The public surface mirrors reference contractn/edges.py:4-50.
"""
from .utils import assert_valid_symbol

_DIM, _SYM, _SELF = "dim", "symbol", "tn_edge"


class Edge:
    __slots__ = ("tn", "name")

    def __init__(self, parent_tn, nx_id, dim, symbol):
        if not (isinstance(nx_id, tuple) and len(nx_id) == 3):
            raise AssertionError("an edge id is a (u, v, key) triple")
        if not isinstance(dim, int):
            raise AssertionError("edge dimension must be an int (-1 = not yet determined)")
        assert_valid_symbol(symbol)
        self.tn, self.name = parent_tn, nx_id
        record = parent_tn.G.edges[nx_id]
        record.update({_DIM: dim, _SYM: symbol, _SELF: self})
        parent_tn._symbol_use[symbol] += 1  # the TN keeps symbol usage counts incrementally

    # -- graph access ---------------------------------------------------------
    @property
    def G(self):
        return self.tn.G

    @property
    def dict(self):
        """The networkx attribute dict backing this edge."""
        return self.tn.G.edges[self.name]

    def _endpoint(self, which):
        return self.tn.G.nodes[self.name[which]]["tn_node"]

    @property
    def nodes(self):
        """The two Node objects joined by the edge (a self loop lists its node twice)."""
        return (self._endpoint(0), self._endpoint(1))

    def other(self, node):
        """The endpoint that is not ``node`` (``node`` itself for a self loop)."""
        first, second = self.nodes
        return second if node is first else first

    # -- metadata -------------------------------------------------------------
    @property
    def symbol(self):
        return self.dict[_SYM]

    @property
    def dim(self):
        return self.dict[_DIM]

    @property
    def var_dim(self):
        """True while the bond dimension is still undetermined (-1)."""
        return self.dict[_DIM] < 0

    @property
    def dangler(self):
        """True for an open leg, i.e. when one endpoint is a dangling placeholder node."""
        return self._endpoint(0).dangler or self._endpoint(1).dangler

    def __repr__(self):
        return f"Edge(name={self.name}, symbol={self.symbol!r}, dim={self.dim})"
