"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- see oracle/rome_oracle.h.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (rome.jl_amd/) never does.
"""
from .ro import *  # noqa: F401,F403
