"""CPU oracle for the identification hot path -- TEST INFRASTRUCTURE ONLY (see ss_oracle.c)."""
