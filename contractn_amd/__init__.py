"""contractn_amd: MI355X-native drop-in for ContracTN's ``TN.contract()`` hot path.

Same public names as reference contractn/__init__.py:1-6.
"""
from .ctn import TN
from .edges import Edge
from .einsum import contract
from .nodes import Node

__all__ = ["TN", "Node", "Edge", "contract"]
