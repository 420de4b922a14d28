"""CPU oracle for the PP / P3M gravity path: test infrastructure only (see pp_oracle.h)."""
