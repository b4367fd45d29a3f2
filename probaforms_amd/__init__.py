"""probaforms_amd -- the conditional RealNVP hot path of hse-cs/probaforms, rebuilt for MI355X.

    from probaforms_amd.models import RealNVP      # mirrors `from probaforms.models import RealNVP`

Only the path named in BASELINE.json is implemented: the affine
coupling stack forward/inverse/backward as hand-written HIP kernels behind the reference's
sklearn-style ``RealNVP.fit(X, C)`` / ``.sample(C)`` API.  There is no CPU fallback.
"""
__version__ = "0.1.0"
