"""dibs_amd -- MI355X-native SVGD engine behind the MarginalDiBS / JointDiBS API of larslorch/dibs."""
__version__ = "0.1.0"
