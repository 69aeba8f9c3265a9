from .gemm import WQLinear_GEMM, WQLinearMMFunction  # noqa: F401
