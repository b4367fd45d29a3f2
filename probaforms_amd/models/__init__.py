"""`from probaforms_amd.models import RealNVP` mirrors `from probaforms.models import RealNVP`
(/root/reference/probaforms/models/__init__.py:1, README.md:48).  Only the RealNVP path is in
scope plus its first "next" row, CVAE; ConditionalWGAN / ConditionalNormal
are not rebuilt."""
from .interfaces import GenModel
from .nflow import InvertibleLayer, NormalizingFlow, StandardNormalPrior
from .realnvp import RealNVP, RealNVPLayer, gen_network
from .cvae import CVAE, Decoder, Encoder

__all__ = ['RealNVP', 'CVAE', 'Encoder', 'Decoder', 'RealNVPLayer', 'NormalizingFlow', 'InvertibleLayer', 'GenModel', 'gen_network',
           'StandardNormalPrior']
