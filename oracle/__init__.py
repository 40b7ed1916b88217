"""CPU oracle for the transformer forward pass — TEST INFRASTRUCTURE ONLY (see gl3_oracle.c header)."""
