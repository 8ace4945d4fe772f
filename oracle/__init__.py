"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/secphase_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  PARITY UNPINNED: restatement of the reference path,
not the reference itself.
"""
