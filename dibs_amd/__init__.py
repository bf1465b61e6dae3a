"""dibs_amd -- MI355X-native SVGD engine behind the MarginalDiBS / JointDiBS API of larslorch/dibs.

    from dibs_amd.inference import MarginalDiBS, JointDiBS
    from dibs_amd.target import make_linear_gaussian_equivalent_model
    from dibs_amd import random            # PRNGKey / split with jax.random stream semantics
"""
__version__ = "0.1.0"
