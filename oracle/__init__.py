"""CPU oracle (test infrastructure). See oracle/matpbr_oracle.c."""
