"""Stated score tolerances of the parity tests, by (head, operand format): max |score - oracle score| per clip.
Shared by tests/test_forward_gpu.py and the smoke check of __graft_entry__.py."""
TOL = {("sls", "f16"): 1e-3, ("sls", "bf16"): 3e-2, ("sae", "f16"): 5e-3, ("sae", "bf16"): 3e-2}
LAYER_RMS = {"f16": 2.5e-3, "bf16": 1.6e-2}   # rms error of a layer output / rms of the layer output
