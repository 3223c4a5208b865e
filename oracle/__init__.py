"""ORACLE — test infrastructure only (CPU restatement; PARITY UNPINNED, see ref_core.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
