"""Drop-in module classes (linear/: WQLinear_GEMM / _GEMV / _GEMVFast; fused/: decoder blocks) over libawq_hip.so."""
