from .graph import ErdosReniDAGDistribution, ScaleFreeDAGDistribution, UniformDAGDistributionRejection  # noqa: F401
from .linearGaussian import LinearGaussian, BGe  # noqa: F401
from .nonlinearGaussian import DenseNonlinearGaussian  # noqa: F401
