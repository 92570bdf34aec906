"""TEST INFRASTRUCTURE: CPU oracle for the tree-likelihood hot path (see oracle/phyoracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
