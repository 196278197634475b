"""CPU oracle for the NeRF-SH hot path. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package. The product path (plenoctree_amd/) never does.
"""
