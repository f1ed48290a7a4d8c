"""varlociraptor_amd — MI355X-native per-locus Bayesian likelihood engine (hot path of
`varlociraptor call variants`).  See DESIGN.md; the C ABI is include/vlr.h."""
from . import abi  # noqa: F401
from .batch import CallResults, PileupBatch  # noqa: F401
from .scenario import Scenario, Sample, Species, Contamination, Inheritance, single_sample, tumor_normal  # noqa: F401
