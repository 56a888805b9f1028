"""The frozen style discriminator the task-level rollout scores the behaviour policy's motion with
(tsc/rsl_rl/algorithms/discriminator.py:12-118): the behaviour tree's network (same parameter names, so its checkpoint entry
`disc` loads) without the task / frame weighting of the inputs, with the input normaliser as a member."""
import types

from quadrupedal_agility_amd.rsl_rl.algorithms.discriminator import Discriminator as _BbcDiscriminator


class Discriminator(_BbcDiscriminator):
    def __init__(self, input_dim, num_disc_obs, dim_c, dt, disc_loss_function, reward_i_normalizer, reward_i_coef, reward_us_coef,
                 reward_ss_coef, reward_t_coef, disc_obs_len, hidden_units, normalizer, device):
        env = types.SimpleNamespace(task_obs_weight_decay=False, task_obs_weight=1.0, task_obs_weight_dev=None)
        super().__init__(env, input_dim, num_disc_obs, dim_c, dt, disc_loss_function, reward_i_normalizer, reward_i_coef, reward_us_coef,
                         reward_ss_coef, reward_t_coef, disc_obs_len, disc_obs_len, 0.0, hidden_units, device)
        self.normalizer = normalizer

    def predict_disc_reward(self, reward_t, obs, obs_disc):
        """:71-118 -- labels from the last dim_c + 1 columns of the behaviour observation row"""
        return super().predict_disc_reward(reward_t, obs, obs_disc, normalizer=self.normalizer)
