"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see hessenberg_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (starneig_amd) never does.
"""
from .oracle import *  # noqa: F401,F403
