"""Agent target arithmetic — numpy restatement (pinned: golden targets.npz produced by calling the
reference's DQNAgent / DDQNAgent / DDPGAgent / TD3Agent .learn_from_batch with stand-in networks)."""
import numpy as np


def dqn_targets(q_next_target, q_online, actions, rewards, game_overs, discount, q_next_online=None):
    """DQNAgent.learn_from_batch (agents/dqn_agent.py:92-103); q_next_online given -> DDQN
    (agents/ddqn_agent.py:43).  Returns (TD_targets fp32 [B,A], TD_errors fp64 [B])."""
    selector = q_next_target if q_next_online is None else q_next_online
    selected_actions = np.argmax(selector, 1)                               # :78-79 / ddqn :43
    TD_targets = np.array(q_online, dtype=np.float32, copy=True)
    TD_errors = []
    for i in range(len(actions)):
        # batch.rewards()[i] is np.float64 and (1.0 - np.bool_) is np.float64, so the whole
        # expression is evaluated in fp64 left to right (the fp32 Q value is promoted)
        new_target = np.float64(rewards[i]) + \
            (1.0 - np.float64(game_overs[i])) * discount * q_next_target[i][selected_actions[i]]
        TD_errors.append(np.abs(new_target - TD_targets[i, actions[i]]))
        TD_targets[i, actions[i]] = new_target
    return TD_targets, np.array(TD_errors, dtype=np.float64)


def ac_td_targets(rewards, game_overs, q_next, discount, use_non_zero_discount_for_terminal_states=False,
                  clip_critic_targets=None):
    """DDPG / TD3 / SAC bootstrapped targets (agents/ddpg_agent.py:156-164, td3_agent.py:171-180,
    soft_actor_critic_agent.py:265-266).  rewards/game_overs (B,), q_next (B,1) fp32."""
    r = np.asarray(rewards, dtype=np.float64)[:, None]
    go = np.asarray(game_overs, dtype=np.float64)[:, None]
    if use_non_zero_discount_for_terminal_states:
        t = r + discount * q_next
    else:
        t = r + (1.0 - go) * discount * q_next
    if clip_critic_targets:
        t = np.clip(t, *clip_critic_targets)
    return t


def td3_smooth_actions(next_actions, noise, noise_clipping, low, high):
    """TD3 target-policy smoothing (agents/td3_agent.py:162-165); `noise` = the un-clipped
    np.random.normal draws."""
    nz = np.asarray(noise).clip(-noise_clipping, noise_clipping)
    return np.clip(next_actions + nz, low, high)                 # spaces.py:379
