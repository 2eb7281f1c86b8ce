"""CPU oracle for the flat-search hot path.  TEST INFRASTRUCTURE ONLY — see oracle/flat_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
