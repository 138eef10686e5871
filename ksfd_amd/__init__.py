"""ksfd_amd -- MI355X-native implicit Keller-Segel finite-difference stepper (hot path of leonavery/KSFD)."""
from .config import ProblemConfig  # noqa: F401
