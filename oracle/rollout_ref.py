"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Line-by-line restatement of the reference's rollout generator `segment()` (orchestrator.py:42-118) on numpy with a
plain list standing in for `agent.rb` -- used to check the product's `loop.segment` row for row.  The reference's
own function cannot be imported (orchestrator.py needs tensordict / gymnasium / wandb, which are absent) and ships no
fixtures: PARITY UNPINNED beyond this restatement.
"""
import numpy as np


class ListBuffer:
    def __init__(self):
        self.rows = []

    def extend(self, td):
        n = len(td["observations"])
        for k in range(n):
            self.rows.append({key: np.array(np.asarray(val)[k]) for key, val in td.items()})

    def __len__(self):
        return len(self.rows)


def segment_ref(env, agent, seed, segment_len, learning_starts, action_repeat):
    obs, _ = env.reset(seed=seed)                                    # :53
    obs = np.asarray(obs, dtype=np.float32)                          # :54
    actions = None                                                   # :55
    t = 0
    r = 0
    while True:
        if r % action_repeat == 0:                                   # :62
            if agent.timesteps_so_far < learning_starts:             # :64
                actions = env.action_space.sample()                  # :65
            else:
                actions = agent.predict({"observations": obs}, explore=True)   # :67-75
        if t > 0 and t % segment_len == 0:                           # :77
            yield                                                    # :78
        next_obs, rewards, terminations, truncations, infos = env.step(actions)   # :81
        next_obs = np.asarray(next_obs, dtype=np.float32)            # :83
        real_next_obs = next_obs.copy()                              # :84
        for idx, trunc in enumerate(np.array(truncations)):          # :86
            if trunc:
                real_next_obs[idx] = np.asarray(infos["final_observation"][idx], dtype=np.float32)   # :88-89
        rewards = np.asarray(rewards, dtype=np.float32)[:, None]     # :91-94
        terminations = np.asarray(terminations, dtype=bool)[:, None]   # :95-98
        agent.rb.extend({                                            # :100-113
            "observations": obs,
            "next_observations": real_next_obs,
            "actions": np.asarray(actions, dtype=np.float32),
            "rewards": rewards,
            "terminations": terminations,
            "dones": terminations,
        })
        obs = next_obs                                               # :115
        t += 1
        r += 1
