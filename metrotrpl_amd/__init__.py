"""metrotrpl_amd -- MI355X engine for MetroTRPL's likelihood hot path (forward_solver / laplace / sim_utils drop-ins,
batched likelihood, Metropolis / parallel-tempering drivers) above the C ABI of include/metrotrpl_amd.h."""
from ._lib import configure_runtime  # noqa: F401  (imported = called once: GPU_MAX_HW_QUEUES, unless MTRPL_KEEP_HW_QUEUES=1)
