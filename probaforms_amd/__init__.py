"""probaforms_amd -- the conditional RealNVP hot path of hse-cs/probaforms, rebuilt for MI355X.

    from probaforms_amd.models import RealNVP      # mirrors `from probaforms.models import RealNVP`

Only the path named in BASELINE.json is implemented: the affine
coupling stack forward/inverse/backward as hand-written HIP kernels behind the reference's
sklearn-style ``RealNVP.fit(X, C)`` / ``.sample(C)`` API.  There is no CPU fallback.
"""
__version__ = "0.1.0"


def install_as_probaforms():
    """Make `from probaforms.models import RealNVP` (the reference's import path, README.md:48) resolve
    to this package: registers `probaforms`, `probaforms.models` and the model modules in sys.modules.
    Call it before anything imports the reference; it refuses to shadow an already imported one."""
    import sys
    import types
    from . import models
    from .models import cvae, interfaces, nflow, realnvp
    existing = sys.modules.get("probaforms")
    if existing is not None and getattr(existing, "__probaforms_amd__", False) is False:
        raise RuntimeError("a different `probaforms` package is already imported (%s)"
                           % getattr(existing, "__file__", "?"))
    pkg = types.ModuleType("probaforms")
    pkg.__probaforms_amd__ = True
    pkg.__path__ = []                      # a package, with no importable submodules of its own
    pkg.models = models
    sys.modules["probaforms"] = pkg
    sys.modules["probaforms.models"] = models
    for name, mod in (("realnvp", realnvp), ("nflow", nflow), ("interfaces", interfaces), ("cvae", cvae)):
        sys.modules["probaforms.models." + name] = mod
    return pkg
