"""Host-side helpers: bit-exact packers (packing.py), layout conversion (convert.py), buffer fusion (fused_utils.py)."""
