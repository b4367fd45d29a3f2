"""CPU oracle for the RealNVP hot path -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Never imported by probaforms_amd (the product path has no CPU fallback).
"""
from .oracle import (CvaeOracle, CvaeShape, Oracle, Shape, build, cvae_flat_from_state, default_masks,  # noqa: F401
                     flat_from_state_dict)
