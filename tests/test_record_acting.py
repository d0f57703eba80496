"""V(s) and the action probabilities recorded BY the acting steps (ClippedPPOAgent.RECORD_WHILE_ACTING,
ClippedPPONet.act_and_record) against the two whole-dataset passes they replace: fill_advantages' value pass
(clipped_ppo_agent.py:161-170) and the old-policy pass (:238-241, hoisted per the reference's own TODO-perf).  The weights
do not change between a rollout and its training phase and networks['main'].sync() (:326) makes the old policy the acting
policy, so the passes recompute the same numbers — up to the fp32 summation order of another tiling (batch n_env instead of
a chunk of the dataset).  Bounds: tests/tolerances.py (the bounds the same quantities have against the oracle)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rollout(dev, n_env, L, playing, batch, flag=True):
    from test_ppo_agent import _make
    random.seed(5); np.random.seed(5)
    agent = _make(dev, n_env, L, playing, batch, 1, seed=0)
    agent.RECORD_WHILE_ACTING = flag          # (instance attribute: this agent only, for its whole life)
    return agent


def _fill(agent, recorded):
    import torch
    from coach_amd import _rlx
    n = agent.memory.num_transitions()
    rows = agent.memory.dataset_rows()
    with _rlx.KernelTimer(4096) as t:
        agent._fill_advantages_device(n, rows, n, recorded)
    torch.cuda.synchronize()
    return {k: getattr(agent, k)[:n].clone() for k in ("ds_value", "ds_old_probs", "ds_adv", "ds_vtarget")}, \
        [nm for nm, _ in t.records]


@pytest.mark.parametrize("n_env,L,playing,batch", [(64, 4, 256, 64), (8, 5, 40, 8), (64, 32, 2048, 64)])
def test_recorded_columns_equal_the_dataset_passes(rlx, dev, n_env, L, playing, batch):
    import torch
    import tolerances as TOL
    agent = _rollout(dev, n_env, L, playing, batch)
    assert agent._records_acting()
    for _ in range(L):
        agent.act()
    assert not agent._rec_missing and agent._should_train()
    agent.networks["main"].update_target(1.0)                        # train(): networks['main'].sync() (:326)
    rec, k_rec = _fill(agent, True)
    ref, k_ref = _fill(agent, False)
    # the recorded form launches no product at all; the other one runs both towers over the dataset
    assert not any(("gemm" in k or "conv" in k) for k in k_rec), k_rec
    assert any(("gemm" in k or "conv" in k) for k in k_ref), k_ref
    dv = float((rec["ds_value"] - ref["ds_value"]).abs().max())
    dp = float((rec["ds_old_probs"] - ref["ds_old_probs"]).abs().max())
    da = float((rec["ds_adv"] - ref["ds_adv"]).abs().max())
    print("n_env %d L %d: |dV| %.2e  |dprobs| %.2e  |dadv| %.2e (adv std 1)" % (n_env, L, dv, dp, da))
    assert torch.allclose(rec["ds_value"], ref["ds_value"], **TOL.OUT)
    assert torch.allclose(rec["ds_old_probs"], ref["ds_old_probs"], **TOL.OUT)
    assert torch.allclose(rec["ds_adv"], ref["ds_adv"], **TOL.ADVANTAGE)
    assert torch.allclose(rec["ds_vtarget"], ref["ds_vtarget"], **TOL.ADVANTAGE)
    # probabilities the actions were drawn from: rows sum to one, the stored action has positive probability
    p = rec["ds_old_probs"]
    assert torch.allclose(p.sum(1), torch.ones_like(p[:, 0]), atol=1e-5)
    a = agent.ds_action[:p.shape[0]].long()
    assert float(p.gather(1, a.view(-1, 1)).min()) > 0


def test_a_step_of_another_phase_switches_the_recording_off_for_that_rollout(rlx, dev):
    """a stored step that did not go through act_and_record (here: one HEATUP step) leaves no V(s) / probabilities in its
    rows: the training phase of that rollout runs the dataset passes; the next rollout records again"""
    from coach_amd.core_types import RunPhase
    agent = _rollout(dev, 8, 5, 40, 8)
    agent.phase = RunPhase.HEATUP
    agent.act()
    agent.phase = RunPhase.TRAIN
    assert agent._rec_missing
    res = None
    while res is None:
        agent.act()
        res = agent.train()
    assert not agent._rec_missing                                    # post_training_commands: a new rollout
    assert any(k[0] == "fill" and k[3] is False for k in agent._warm), sorted(map(str, agent._warm))
    res = None
    while res is None:
        agent.act()
        res = agent.train()
    assert any(k[0] == "fill" and k[3] is True for k in agent._warm), sorted(map(str, agent._warm))


def test_an_iteration_with_and_without_the_recording_agree(rlx, dev):
    """the same seeds with the flag off and on: sampled actions of the first rollout identical or differing only where
    a uniform draw sits inside the fp32 noise of a CDF boundary (none expected in 256 draws), trained weights equal to
    the bounds the weights have against the oracle"""
    import torch
    import tolerances as TOL
    out = {}
    for flag in (False, True):
        agent = _rollout(dev, 64, 4, 256, 64, flag)
        assert agent._records_acting() == flag
        acts, res = [], None
        while res is None:
            agent.act()
            acts.append(agent.actions.clone())
            res = agent.train()
        out[flag] = (torch.stack(acts), agent.networks["main"].params.weights.clone(), torch.stack([r.clone() for r in res]))
        agent.check_status()
    assert torch.equal(out[False][0], out[True][0])
    assert torch.allclose(out[False][1], out[True][1], **TOL.WEIGHTS)
    assert torch.allclose(out[False][2][:, :5], out[True][2][:, :5], **TOL.LOSS)
