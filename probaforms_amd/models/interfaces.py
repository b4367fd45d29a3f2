"""Base class shared by the generative models.

Mirrors `probaforms.models.interfaces.GenModel`
(/root/reference/probaforms/models/interfaces.py:6-43): an ``nn.Module`` whose ``fit`` and
``sample`` do nothing, so that code which walks ``GenModel.__subclasses__()`` (the reference's
tests/test_models.py:6-10) discovers the models here the same way.
"""
import torch.nn as nn


class GenModel(nn.Module):
    """Conditional generative model: ``fit(X, C)`` learns p(X | C), ``sample(C)`` draws from it."""

    def __init__(self):
        super().__init__()

    def fit(self, X, C):
        """X: array [n, var_size]; C: array [n, cond_size] or None.  No-op in the base class."""
        return None

    def sample(self, C):
        """C: array [n, cond_size] or an int (number of draws).  No-op in the base class."""
        return None
