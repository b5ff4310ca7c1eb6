"""zkevm_specs_amd — MI355X-native constraint-evaluation engine for the zkEVM spec circuits."""
import os as _os

# one hardware queue per concurrent circuit session (see csrc/zkevm_hip.hip zk_default_hw_queues): must be in the environment
# before the HIP runtime's first call, which torch makes lazily — importing this package first is early enough
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
