"""zkevm_specs_amd — MI355X-native constraint-evaluation engine for the zkEVM spec circuits."""
