"""Build-owned stand-in for the 8 names the reference imports from ``opt_einsum``.

TEST INFRASTRUCTURE, used ONLY by oracle/gen_golden.py inside the build
container (opt_einsum is not installed and cannot be: no network).  It lets the
reference's unmodified files import and run so that golden vectors can be
captured; it never travels with the product path and nothing under
``contractn_amd/`` imports it.  Path *choice* is the only logic here that is
not the reference's: explicit paths are honoured as given, otherwise the
operands are contracted left to right.  All arithmetic is NumPy's.
"""
from . import backends, contract as _contract_mod  # noqa: F401
from .contract import contract_path, get_symbol  # noqa: F401
