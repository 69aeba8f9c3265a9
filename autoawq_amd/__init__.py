"""autoawq_amd -- MI355X-native AWQ int4 weight-only matmul path (drop-in for awq/modules/linear/*)."""
__version__ = "0.1.0"

from .modules.linear import WQLinear_GEMM, WQLinear_GEMV, WQLinear_GEMVFast, WQLinearMMFunction  # noqa: F401
