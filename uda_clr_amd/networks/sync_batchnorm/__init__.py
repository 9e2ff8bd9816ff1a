from .batchnorm import BatchNorm2d  # noqa: F401  (TransNorm2d; the reference's ``--use_TN`` normalisation)
