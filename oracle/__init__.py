"""CPU oracle (test infrastructure only) -- see vlsat_oracle.py header."""
