from .gemm import WQLinear_GEMM, WQLinearMMFunction  # noqa: F401
from .gemv import WQLinear_GEMV  # noqa: F401
