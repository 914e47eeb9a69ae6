"""``TN``: NetworkX-backed tensor-network front-end that dispatches ``contract()`` to the HIP engine.

API kept from reference contractn/ctn.py (``TN``, ``add_dense_node`` :130,
``add_duplicate_node`` :141, ``add_copy_node`` :154, ``add_input_node`` :167,
``connect_nodes`` :179, ``remove_edge(s)`` :206-229, ``nodes``/``edges``
:315-347, ``make_contract_fun`` :349, ``contract`` :389, ``einsum_str`` :411,
``params`` :418, counters :430-474).  The graph layout (one ``_dangler_k`` graph
node per open leg, metadata in the networkx attribute dicts) is the same, so
``tn.G`` can be inspected exactly as before.

Deliberate deviations:
* copy-node clusters share ONE symbol on *all* their legs, including legs that
  end on dense nodes created before the copy node (reference ctn.py:271-313
  renames only copy/dangler edges and emits ``"a->bb"`` in that case);
* ``connect_nodes`` accepts node names (reference ctn.py:184-192 crashes);
* clone nodes can be contracted (reference einsum.py:152 tests the wrong tag);
* the set of used symbols is kept incrementally (reference rescans all edges
  on every call, O(N^2) network construction).
"""
from collections import Counter

import networkx as nx

from .edges import Edge
from .einsum import contract, make_arg_packer, make_einstring
from .nodes import Node
from .utils import assert_valid_symbol, assert_valid_tensor, get_new_symbols


class TN:
    """Generic tensor network with copy nodes (hyperedges) and weight sharing."""

    def __init__(self):
        self.G = nx.MultiGraph()
        self._dang_id = 0
        self._symbol_use = Counter()  # symbol -> number of edges carrying it
        self._n_cores = 0

    # ------------------------------------------------------------------ nodes
    def _init_node(self, node_type, name, edge_symbols, **kwargs):
        assert node_type != "dangler"
        name = self._new_node_name(name)
        self.G.add_node(name)
        try:
            node = Node(self, node_type, name, edge_symbols, **kwargs)
        except Exception:
            self.G.remove_node(name)
            raise
        self._n_cores += 1
        return node

    def _new_dangler(self, parent, idx, edge_symbol):
        nx_id = f"_dangler_{self._dang_id}"
        assert nx_id not in self.G and parent.name in self.G
        self._dang_id += 1
        self.G.add_node(nx_id)
        dangler = Node(self, "dangler", nx_id, (edge_symbol,))
        return self._init_edge(parent, dangler, idx, 0, edge_symbol)

    def _new_node_name(self, name=None):
        if name is None:
            name = f"node_{self.num_cores}"
        assert isinstance(name, str)
        if self.G.has_node(name):
            raise TypeError(f"Node name '{name}' already in use")
        return name

    def _new_edge_symbols(self, node_type, degree, edge_symbols=None):
        if edge_symbols is not None:
            assert len(edge_symbols) == degree
            for es in edge_symbols:
                assert_valid_symbol(es)
                if self._symbol_use[es] > 0:
                    raise TypeError(f"Edge symbol '{es}' already in use")
            return tuple(edge_symbols)
        assert node_type != "dangler" and degree >= 0
        if degree == 0:
            return ()
        if node_type == "hyper":  # one symbol repeated on every leg
            return get_new_symbols(self.edge_symbols, 1) * degree
        return get_new_symbols(self.edge_symbols, degree)

    def add_dense_node(self, tensor, name=None, edge_symbols=None):
        """Add a dense core tensor (``ndim``/``shape`` array-like)."""
        assert_valid_tensor(tensor)
        syms = self._new_edge_symbols("dense", tensor.ndim, edge_symbols)
        return self._init_node("dense", name, syms, tensor=tensor)

    def add_duplicate_node(self, base_node, name=None, edge_symbols=None):
        """Add a clone of a dense node (shares its tensor)."""
        if not isinstance(base_node, Node):
            assert base_node in self.G
            base_node = self[base_node]
        syms = self._new_edge_symbols("clone", base_node.ndim, edge_symbols)
        return self._init_node("clone", name, syms, base_node=base_node)

    def add_copy_node(self, degree, dim=None, name=None, edge_symbols=None):
        """Add a copy (hyperedge) node: no tensor, one symbol on all ``degree`` legs."""
        if degree <= 0:
            raise AssertionError("Hyperedge nodes must have positive degree")
        if isinstance(edge_symbols, str):
            edge_symbols = (edge_symbols,) * degree
        if edge_symbols is not None:
            assert len(edge_symbols) == degree and len(set(edge_symbols)) == 1
            assert_valid_symbol(edge_symbols[0])
            if self._symbol_use[edge_symbols[0]] > 0:
                raise TypeError(f"Edge symbol '{edge_symbols[0]}' already in use")
            syms = tuple(edge_symbols)
        else:
            syms = self._new_edge_symbols("hyper", degree)
        return self._init_node("hyper", name, syms, degree=degree, dim=dim)

    def add_input_node(self, shape, var_shape_axes=(), name=None, edge_symbols=None):
        """Add a placeholder node whose tensor is supplied at contraction time."""
        syms = self._new_edge_symbols("input", len(shape), edge_symbols)
        return self._init_node("input", name, syms, shape=shape, var_axes=var_shape_axes)

    # ------------------------------------------------------------------ edges
    def _init_edge(self, node1, node2, idx1, idx2, edge_symbol):
        assert isinstance(node1, Node) and isinstance(node2, Node)
        assert node1 in self and node2 in self
        shape1, shape2 = node1.shape, node2.shape
        assert -len(shape1) <= idx1 < len(shape1)
        assert -len(shape2) <= idx2 < len(shape2)

        # variable (-1) dimensions adopt the size of their partner
        dim1, dim2 = shape1[idx1], shape2[idx2]
        if dim1 < 0 or dim2 < 0:
            new_dim = max(dim1, dim2)
        else:
            assert dim1 == dim2, f"incompatible dimensions {dim1} and {dim2}"
            new_dim = dim1

        n1, n2 = node1.name, node2.name
        edge_id = (n1, n2, self.G.add_edge(n1, n2))
        Edge(self, edge_id, new_dim, edge_symbol)
        if not node1.dangler:
            node1.dict["edge_names"][idx1] = edge_id
        if not node2.dangler:
            node2.dict["edge_names"][idx2] = edge_id

        for node in (node1, node2):
            if node.is_copy:
                self._cleanup_edge_symbols(node)
                break
        return edge_id

    def _set_edge_symbol(self, edge_id, symbol):
        attrs = self.G.edges[edge_id]
        old = attrs["symbol"]
        if old == symbol:
            return
        self._symbol_use[old] -= 1
        if self._symbol_use[old] <= 0:
            del self._symbol_use[old]
        self._symbol_use[symbol] += 1
        attrs["symbol"] = symbol

    def _drop_edges(self, edge_ids):
        for eid in edge_ids:
            sym = self.G.edges[eid]["symbol"]
            self._symbol_use[sym] -= 1
            if self._symbol_use[sym] <= 0:
                del self._symbol_use[sym]

    def connect_nodes(self, node1, node2, index1, index2, edge_symbol=None):
        """Join two dangling legs into a bond (reference ctn.py:179-204)."""
        if not isinstance(node1, Node):
            node1 = self[node1]
        if not isinstance(node2, Node):
            node2 = self[node2]
        es1, es2 = node1.edge_symbols[index1], node2.edge_symbols[index2]
        if edge_symbol is None:
            edge_symbol = min(es1, es2)
        elif self._symbol_use[edge_symbol] > 0:
            assert edge_symbol in (es1, es2)

        dang1, dang2 = node1._dang_name(index1), node2._dang_name(index2)
        assert dang1 != dang2
        for dang in (dang1, dang2):
            self._drop_edges(list(self.G.edges(dang, keys=True)))
            self.G.remove_node(dang)
        self._init_edge(node1, node2, index1, index2, edge_symbol)

    def _remove_edges(self, edge_set):
        assert all(isinstance(e, Edge) for e in edge_set)
        freed = []  # (node, leg index) for both ends of every removed bond
        removed = []
        for e in edge_set:
            assert e.name in self.G.edges
            if e.dangler:
                continue
            removed.append(e.name)
            n1, n2 = e.nodes
            if n1 is n2:  # self loop: two distinct legs of one node
                legs = [i for i, en in enumerate(n1.dict["edge_names"]) if en == e.name]
                freed.extend((n1, i) for i in legs)
            else:
                freed.append((n1, n1.index(e)))
                freed.append((n2, n2.index(e)))
        self._drop_edges(removed)
        self.G.remove_edges_from(removed)

        new_symbols = get_new_symbols(self.edge_symbols, len(freed))
        for (node, leg), sym in zip(freed, new_symbols):
            node.dict["edge_names"][leg] = self._new_dangler(node, leg, sym)
        if any(node.is_copy for node, _ in freed):
            self._cleanup_edge_symbols()

    def remove_edge(self, edge):
        """Break one bond, leaving two dangling legs."""
        assert isinstance(edge, (Edge, tuple))
        if isinstance(edge, tuple):
            edge = self.G.edges[edge]["tn_edge"]
        self._remove_edges([edge])

    def remove_edges_from(self, edge_set):
        """Break several bonds; edges that already dangle are left unchanged."""
        edges, seen = [], set()
        for e in tuple(edge_set):
            assert isinstance(e, (Edge, tuple))
            if isinstance(e, tuple):
                e = self.G.edges[e]["tn_edge"]
            if e.name not in seen:
                seen.add(e.name)
                edges.append(e)
        self._remove_edges(edges)

    # ------------------------------------------------- copy-tensor symbol merge
    def _copy_clusters(self, seed=None):
        """Connected clusters of copy nodes (joined directly or through nothing else)."""
        G = self.G
        hyper = [n for n, t in G.nodes(data="node_type") if t == "hyper"]
        sub = G.subgraph(hyper)
        if seed is not None:
            return [nx.node_connected_component(sub, seed.name)]
        return list(nx.connected_components(sub))

    def _cleanup_edge_symbols(self, naughty_node=None):
        """Give every leg of a connected copy-node cluster one shared symbol.

        Reference ctn.py:271-313 picks the most common incident symbol; we do the
        same but (a) rename *every* incident leg, including bonds to dense nodes,
        and (b) hand out a fresh symbol if the favourite is also used outside the
        cluster (clusters split by ``remove_edge``).
        """
        if naughty_node is not None:
            assert naughty_node.is_copy
        for cluster in self._copy_clusters(naughty_node):
            incident = list(self.G.edges(cluster, keys=True, data="symbol"))
            if not incident:
                continue
            tally = Counter()
            seen = set()
            for u, v, k, sym in incident:
                key = (min(u, v), max(u, v), k)
                if key in seen:
                    continue
                seen.add(key)
                tally[sym] += 1
            best, n_inside = tally.most_common(1)[0]
            if self._symbol_use[best] > n_inside:
                free = [s for s, c in tally.items() if self._symbol_use[s] == c]
                best = min(free) if free else get_new_symbols(self.edge_symbols, 1)[0]
            assert_valid_symbol(best)
            for u, v, k, _ in incident:
                self._set_edge_symbol((u, v, k), best)

    # -------------------------------------------------------------- iteration
    def nodes(self, as_iter=False, copy_nodes=True, danglers=False):
        """Node objects in insertion order (reference ctn.py:315-336)."""
        it = (
            d["tn_node"]
            for _, d in self.G.nodes.data()
            if (danglers or d["node_type"] != "dangler")
            and (copy_nodes or d["node_type"] != "hyper")
        )
        return it if as_iter else tuple(it)

    def edges(self, as_iter=False):
        it = (e for _, _, e in self.G.edges(data="tn_edge"))
        return it if as_iter else tuple(it)

    # ------------------------------------------------------------ contraction
    def make_contract_fun(self, optimize="auto", split_format=False):
        """Compile the TN once; returns ``contract_fun(params, inputs)``.

        Same contract as reference ctn.py:349-387.  The einsum string and operand
        packer are frozen here; the closure hands them to :func:`contract`, whose
        plan cache keeps the native plan handle alive across calls.
        """
        einstr = self.einsum_str
        arg_packer = make_arg_packer(self)

        def contract_fun(params, inputs):
            operands = arg_packer(params, inputs)
            return contract(einstr, *operands, optimize=optimize, split_format=split_format)

        contract_fun.einsum_str = einstr
        contract_fun.arg_packer = arg_packer
        return contract_fun

    def make_batched_contract_fun(self, replicas, optimize="auto", dtype=None, device=0):
        """Throughput form of :meth:`make_contract_fun`: ``fun(params_list, inputs_list)`` contracts
        ``replicas`` independent parameter sets of this network in ONE launch sequence (every
        pairwise step is a single kernel over all replicas) and returns ``(t_hat, log_scale)``
        stacked along a leading replica axis (split format).  A dependent chain of steps cannot
        fill the GPU with one network; this is how the headline throughput is reached."""
        import numpy as np

        from .einsum import BatchedContraction

        einstr = self.einsum_str
        arg_packer = make_arg_packer(self)
        state = {}

        def batched_contract_fun(params_list, inputs_list=None):
            assert len(params_list) == replicas
            inputs_list = inputs_list if inputs_list is not None else [()] * replicas
            sets = [arg_packer(p, i) for p, i in zip(params_list, inputs_list)]
            if "bc" not in state:
                shapes = [tuple(o.shape) for o in sets[0]]
                dt = dtype or np.result_type(*[np.asarray(o).dtype for o in sets[0]])
                dt = np.float32 if dt == np.float32 else np.float64
                state["bc"] = BatchedContraction(einstr, shapes, dt, optimize=optimize, replicas=replicas,
                                                 device=device)
            return state["bc"].run_host(sets)

        batched_contract_fun.einsum_str = einstr
        return batched_contract_fun

    def contract(self, inputs=(), optimize="auto", split_format=False):
        """Contract the network to a dense tensor (reference ctn.py:389-409)."""
        fun = self.make_contract_fun(optimize=optimize, split_format=split_format)
        return fun(self.params, inputs)

    @property
    def einsum_str(self):
        return make_einstring(self)

    @property
    def params(self):
        return tuple(
            n.tensor for n in self.nodes(as_iter=True, copy_nodes=False) if n.node_type == "dense"
        )

    # --------------------------------------------------------------- counters
    def _count(self, kind):
        return sum(1 for _, t in self.G.nodes(data="node_type") if t == kind)

    @property
    def num_dense(self):
        return self._count("dense")

    @property
    def num_duplicate(self):
        return self._count("clone")

    @property
    def num_copy(self):
        return self._count("hyper")

    @property
    def num_input(self):
        return self._count("input")

    @property
    def num_cores(self):
        return self._n_cores

    @property
    def edge_symbols(self):
        return {s for s, c in self._symbol_use.items() if c > 0}

    def __contains__(self, node):
        if isinstance(node, Node):
            node = node.name
        return node in self.G

    def __getitem__(self, name):
        assert name in self.G
        return self.G.nodes[name]["tn_node"]
