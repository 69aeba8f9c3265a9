from .gemm import WQLinear_GEMM, WQLinearMMFunction  # noqa: F401
from .gemv import WQLinear_GEMV  # noqa: F401
from .gemv_fast import WQLinear_GEMVFast  # noqa: F401
