"""CPU oracle: test infrastructure only (see oracle/epik_oracle.h)."""
