"""CPU oracle (test infrastructure).  See oracle/meshanything_oracle.py and oracle/README.md."""
