"""CPU oracle for the AWQ int4 matmul path -- TEST INFRASTRUCTURE ONLY (see awq_oracle.c)."""
