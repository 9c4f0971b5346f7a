"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/tsim_oracle.cpp header). PARITY UNPINNED."""
