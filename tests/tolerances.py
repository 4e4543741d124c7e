"""Stated score tolerances of the parity tests, by (head, operand format): max |score - oracle score| per clip.
Shared by tests/test_forward_gpu.py and the smoke check of __graft_entry__.py."""
TOL = {("sls", "f16"): 1e-3, ("sls", "bf16"): 3e-2, ("sae", "f16"): 5e-3, ("sae", "bf16"): 3e-2}
LAYER_RMS = {"f16": 2.5e-3, "bf16": 1.6e-2}   # rms error of a layer output / rms of the layer output
# The in-tree head at the benchmark shape (201 frames): the 2048-clip fixture and the eight weight seeds.  Its bound is
# an observed one: the top-128-of-4096 selection turns the backbone's error into changed members of the selection
# (the auto-encoder's own operand rounding is not what is felt: oracle/precision_ablate.py --head sae gives the same
# error with the encoder's operands exact, rms 1.18e-3 against 1.13e-3 over 64 clips).  Clips of fewer frames pool
# fewer selections and sit closer to TOL (4.8e-3 at 149 frames).
SAE_F16_201_FRAMES = 4e-3
SAE_F16_PER_SEED_64_CLIPS = 3.5e-3
