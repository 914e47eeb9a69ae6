"""contractn_amd: MI355X-native drop-in for the ``TN.contract()`` hot path of ContracTN.

``TN``, ``Node``, ``Edge`` and ``contract`` carry the reference's public names (reference
contractn/__init__.py); ``BatchedContraction`` (R networks per launch sequence) and
``contractn_amd.dist`` (multi-GPU slicing / joins) are the additions of this engine.
"""
from . import engine  # noqa: F401  (ctypes binding of the C ABI; no GPU needed to import)
from .ctn import TN
from .edges import Edge
from .einsum import BatchedContraction, clear_caches, contract, destabilize, stabilize
from .nodes import Node

__version__ = "0.1.0"
__all__ = ["TN", "Node", "Edge", "contract", "stabilize", "destabilize", "BatchedContraction", "clear_caches", "engine", "__version__"]
