"""CPU oracle (test infrastructure only -- see loftr_oracle.py header)."""
