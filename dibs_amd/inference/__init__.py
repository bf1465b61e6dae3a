from .dibs import DiBS  # noqa: F401
from .svgd import MarginalDiBS, JointDiBS  # noqa: F401
