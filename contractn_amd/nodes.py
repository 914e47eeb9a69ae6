"""Node handle of the TN front-end, including the copy-tensor (hyperedge) encoding.

Host bookkeeping only.  Public surface follows reference contractn/nodes.py:42-328.
The part that matters for the device path is the copy-node encoding
(reference nodes.py:63-64, 79-84, 229-230): a copy node owns *no tensor*; all
of its legs carry one einsum symbol, so it never becomes an operand and the
engine sees a label shared by >2 operands (lowered to batch/gather-multiply
kernels, never to a materialised delta tensor).
"""
from math import prod

from .edges import Edge
from .utils import assert_valid_tensor, node_attr_error, opposite_node

NODE_TYPES = ("dense", "clone", "hyper", "input", "dangler")

# node type -> (mandatory kwargs, optional kwargs); reference nodes.py:19-39
_NODE_KWARGS = {
    "dense": ({"tensor"}, set()),
    "clone": ({"base_node"}, set()),
    "hyper": ({"degree"}, {"dim"}),
    "input": ({"shape", "var_axes"}, set()),
    "dangler": (set(), set()),
}


def check_node_args(node_type, kwargs):
    if node_type not in _NODE_KWARGS:
        raise TypeError(f"Unknown node_type '{node_type}'")
    need, maybe = _NODE_KWARGS[node_type]
    given = set(kwargs)
    for missing in sorted(need - given):
        raise TypeError(f"Argument '{missing}' missing, needed for node_type '{node_type}'")
    for extra in sorted(given - need - maybe):
        raise TypeError(f"Argument '{extra}' not recognized for node_type '{node_type}'")


class Node:
    def __init__(self, parent_tn, node_type, nx_name, edge_symbols, **kwargs):
        check_node_args(node_type, kwargs)
        assert nx_name in parent_tn.G
        self.tn = parent_tn
        self.name = nx_name
        n_legs = len(edge_symbols)
        info = self.dict
        info["node_type"] = node_type
        info["tn_node"] = self
        info["edge_names"] = [None] * n_legs

        if node_type == "hyper":
            assert len(set(edge_symbols)) <= 1
            degree = kwargs["degree"]
            assert degree > 0, "Hyperedge nodes must have positive degree"
            assert n_legs == degree
            dim = kwargs.get("dim")
            assert dim is None or isinstance(dim, int)
            info["degree"], info["dim"] = degree, dim
        else:
            assert len(set(edge_symbols)) == n_legs
            if node_type == "dense":
                info["tensor"] = kwargs["tensor"]
                assert n_legs == info["tensor"].ndim
            elif node_type == "clone":
                base = kwargs["base_node"]
                if not isinstance(base, Node):
                    base = parent_tn.G.nodes[base]["tn_node"]
                assert base.node_type == "dense"
                assert n_legs == base.tensor.ndim
                info["base_node"] = base
            elif node_type == "input":
                shape, var_axes = tuple(kwargs["shape"]), tuple(kwargs["var_axes"])
                assert n_legs == len(shape)
                assert len(set(var_axes)) == len(var_axes)
                assert all(0 <= ax < n_legs for ax in var_axes)
                info["_shape"], info["var_axes"] = shape, var_axes

        if node_type != "dangler":
            for leg, sym in enumerate(edge_symbols):
                info["edge_names"][leg] = parent_tn._new_dangler(self, leg, sym)

    # -- identity -----------------------------------------------------------
    @property
    def G(self):
        return self.tn.G

    @property
    def dict(self):
        return self.tn.G.nodes[self.name]

    @property
    def node_type(self):
        return self.dict["node_type"]

    @property
    def dangler(self):
        return self.node_type == "dangler"

    @property
    def is_copy(self):
        return self.node_type == "hyper"

    # -- legs ---------------------------------------------------------------
    @property
    def edge_names(self):
        if self.dangler:
            return list(self.G.edges(self.name, keys=True))
        return self.dict["edge_names"]

    @property
    def edges(self):
        edges = self.G.edges
        return tuple(edges[en]["tn_edge"] for en in self.edge_names)

    @property
    def edge_symbols(self):
        edges = self.G.edges
        return tuple(edges[en]["symbol"] for en in self.edge_names)

    @property
    def symbol(self):
        """Symbol of the single edge of a dangling node."""
        assert self.dangler
        (_, _, sym), = self.G.edges(self.name, data="symbol")
        return sym

    def _dang_name(self, idx):
        other = opposite_node(self.dict["edge_names"][idx], self.name)
        assert self.G.nodes[other]["node_type"] == "dangler", "edge is already connected"
        return other

    @property
    def neighbors(self):
        nodes = self.G.nodes
        return tuple(nodes[opposite_node(e, self.name)]["tn_node"] for e in self.edge_names)

    def __getitem__(self, key):
        return self.G.edges[self.edge_names[key]]["tn_edge"]

    def index(self, edge):
        if isinstance(edge, Edge):
            edge = edge.name
        return self.edge_names.index(edge)

    # -- shape --------------------------------------------------------------
    @property
    def shape(self):
        kind = self.node_type
        if kind == "dense":
            return tuple(self.dict["tensor"].shape)
        if kind == "clone":
            return tuple(self.dict["base_node"].tensor.shape)
        if kind == "hyper":
            dim = self.dict["dim"]
            return (-1 if dim is None else dim,) * self.dict["degree"]
        if kind == "input":
            var = self.dict["var_axes"]
            return tuple(-1 if i in var else d for i, d in enumerate(self.dict["_shape"]))
        return (-1,)

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        if self.dangler:
            return None
        shape = self.shape
        return None if any(d < 0 for d in shape) else prod(shape)

    @property
    def numel(self):
        return prod(self.tensor.shape) if self.node_type == "dense" else 0

    # -- type specific attributes ------------------------------------------
    def _typed(self, owner, attr):
        if self.node_type != owner:
            raise node_attr_error(owner, attr, self.name, self.node_type)
        return self.dict[attr]

    @property
    def tensor(self):
        return self._typed("dense", "tensor")

    @tensor.setter
    def tensor(self, array):
        self._typed("dense", "tensor")
        assert_valid_tensor(array)
        assert array.ndim == self.ndim
        self.dict["tensor"] = array

    @property
    def base_node(self):
        return self._typed("clone", "base_node")

    @property
    def degree(self):
        return self._typed("hyper", "degree")

    @property
    def dim(self):
        return self._typed("hyper", "dim")

    @property
    def var_axes(self):
        return self._typed("input", "var_axes")

    def __repr__(self):
        return f"Node(name={self.name}, node_type={self.node_type}, degree={self.ndim})"
