"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Restatement of the evaluation side of the reference's training loop, used to check the product's `loop.episode`,
`loop.Evaluator` and `loop.evaluate` on a synthetic vector env:

* `episode_ref`      orchestrator.py:121-246 (the trajectory lists of `need_lists=True`; pixels left out)
* `EvalBlockRef`     orchestrator.py:303-305 (rolling buffers), :354-380 (eval block: `eval_steps` episodes, float32 means
                     over the window, best-model bookkeeping), :319-322 and :392-397 (speed with burn-in and eval time excluded)
* `evaluate_ref`     orchestrator.py:436-476 (episode loop and the final means)

The reference's own functions cannot be imported (orchestrator.py needs tensordict / gymnasium / wandb / tqdm, absent
here) and it ships no fixtures: PARITY UNPINNED beyond this restatement.  torch's float32 `.mean()` of the reference is
stated with numpy float32 (same IEEE type; summation order can differ in the last bit, so comparisons use rtol 1e-6).
"""
from collections import deque

import numpy as np


def episode_ref(env, agent, seed, need_lists=False):
    rng = np.random.default_rng(seed)                                   # :136

    def randomize_seed():
        return seed + rng.integers(2**32 - 1, size=1).item()            # :138-140

    obs_list, next_obs_list, actions_list, rewards_list, terminations_list, dones_list = [], [], [], [], [], []   # :142-147
    ob, _ = env.reset(seed=randomize_seed())                            # :152
    if need_lists:
        obs_list.append(ob)                                             # :156-157
    while True:
        action = agent.predict({"observations": np.asarray(ob, np.float32)}, explore=False)    # :165-173
        new_ob, reward, termination, truncation, infos = env.step(action)                      # :175
        done = termination or truncation                                                       # :180  (one-env arrays)
        if need_lists:                                                                         # :182-193
            next_obs_list.append(new_ob)
            actions_list.append(action)
            rewards_list.append(reward)
            terminations_list.append(termination)
            dones_list.append(done)
            if not done:
                obs_list.append(new_ob)
        ob = new_ob                                                                            # :196
        if "final_info" in infos:                                                              # :198
            for info in infos["final_info"]:
                ep_len = float(info["episode"]["l"].item())
                ep_ret = float(info["episode"]["r"].item())
            if need_lists:
                out = {"observations": np.array(obs_list), "actions": np.array(actions_list),
                       "next_observations": np.array(next_obs_list), "rewards": np.array(rewards_list),
                       "terminations": np.array(terminations_list), "dones": np.array(dones_list),
                       "length": np.array(ep_len), "return": np.array(ep_ret)}
            else:
                out = {"length": np.array(ep_len), "return": np.array(ep_ret)}
            yield out
            obs_list, next_obs_list, actions_list, rewards_list, terminations_list, dones_list = [], [], [], [], [], []
            ob, _ = env.reset(seed=randomize_seed())                                           # :238
            if need_lists:
                obs_list.append(ob)


class EvalBlockRef:
    def __init__(self, cfg, eval_env, agent, clock):
        self.cfg, self.agent, self.clock = cfg, agent, clock
        self.ep_gen = episode_ref(eval_env, agent, cfg.seed)            # :293-294
        maxlen = 20 * cfg.eval_steps                                    # :303
        self.len_buff, self.ret_buff = deque(maxlen=maxlen), deque(maxlen=maxlen)
        self.start_time, self.measure_burnin, self.time_spent_eval = None, None, 0
        self.saved = []

    def loop_top(self):                                                 # :319-322
        a, c = self.agent, self.cfg
        if self.start_time is None and a.timesteps_so_far >= (c.measure_burnin + c.learning_starts):
            self.start_time = self.clock()
            self.measure_burnin = a.timesteps_so_far

    def eval_block(self):                                               # :354-405
        a, c = self.agent, self.cfg
        eval_start = self.clock()
        for _ in range(c.eval_steps):
            ep = next(self.ep_gen)
            self.len_buff.append(ep["length"])
            self.ret_buff.append(ep["return"])
        metrics = {"length": float(np.array(list(self.len_buff)).astype(np.float32).mean()),
                   "return": float(np.array(list(self.ret_buff)).astype(np.float32).mean())}
        out = {"timestep": a.timesteps_so_far, **metrics}
        if (new_best := metrics["return"]) > a.best_eval_ep_ret:
            a.best_eval_ep_ret = new_best
            self.saved.append(a.timesteps_so_far)                       # agent.save(ckpt_dir, sfx="best")
        self.time_spent_eval += self.clock() - eval_start
        if self.start_time is not None:
            out["speed"] = (a.timesteps_so_far - self.measure_burnin) / (self.clock() - self.start_time - self.time_spent_eval)
        return out


def evaluate_ref(cfg, env, agent):
    ep_gen = episode_ref(env, agent, cfg.seed, need_lists=cfg.gather_trajectories)             # :433-434
    len_list, ret_list, eps = [], [], []
    for _ in range(cfg.num_episodes):
        ep = next(ep_gen)
        len_list.append(ep["length"])
        ret_list.append(ep["return"])
        eps.append(ep)
    return {"length": float(np.array(len_list).astype(np.float32).mean()),
            "return": float(np.array(ret_list).astype(np.float32).mean())}, eps
