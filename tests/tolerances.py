"""Stated tolerances of the fp32 device engine against the numpy oracle, shared by the Clipped-PPO agent
tests so that the document and the tests cannot drift apart.  Measured at the full C2 size on the MI355X
(tests/test_ppo_full_size.py, `pytest -s` prints the worst deviations): V(s) 5e-7 absolute, standardised advantages
3.4e-4 relative / 6e-7 absolute, minibatch losses 1e-6 relative, gradient norm 5e-8 relative, weights after three Adam
steps 1.5e-8 absolute.  The bounds below leave room for a different fp32 summation order (split-K factors and tile
shapes are tuning parameters), not for a different algorithm.

Integer / byte results (replay indices, sampled actions, gathered frames, PER trees) are compared bit for bit and do not
use these."""
OUT = dict(rtol=1e-4, atol=2e-6)        # network outputs: V(s), action probabilities, policy means
ADVANTAGE = dict(rtol=1e-3, atol=2e-5)  # (A - mean) / std: the fp32 error of V(s) divided by a std of O(0.1 .. 1)
LOSS = dict(rtol=2e-4, atol=2e-6)       # minibatch or epoch-mean losses, gradient norms
WEIGHTS = dict(rtol=2e-4, atol=1e-6)    # every weight after the Adam steps of a training phase
