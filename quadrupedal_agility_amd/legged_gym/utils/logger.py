"""Play-time logger: same surface as bbc/legged_gym/utils/logger.py:7-135 (log_state(s), log_rewards, reset,
plot_states, plot_dof_pos, print_rewards).  Plotting needs matplotlib, which is optional here: without it the
plot calls dump the logged series to an .npz next to the working directory instead of opening windows."""
from collections import defaultdict

import numpy as np


class Logger:
    def __init__(self, dt):
        self.state_log = defaultdict(list)
        self.rew_log = defaultdict(list)
        self.dt = dt
        self.num_episodes = 0
        self.plot_process = None

    def log_state(self, key, value):
        self.state_log[key].append(value)

    def log_states(self, dict):
        for key, value in dict.items():
            self.log_state(key, value)

    def log_rewards(self, dict, num_episodes):
        for key, value in dict.items():
            if "rew" in key:
                self.rew_log[key].append(float(value) * num_episodes)
        self.num_episodes += num_episodes

    def reset(self):
        self.state_log.clear()
        self.rew_log.clear()

    # ------------------------------------------------------------------ output
    def series(self):
        """name -> (T,) or (T, k) array of everything logged so far"""
        return {k: np.asarray(v) for k, v in self.state_log.items() if len(v)}

    def _time(self):
        n = max((len(v) for v in self.state_log.values()), default=0)
        return np.linspace(0, n * self.dt, n)

    def plot_states(self, path="play_states.npz"):
        try:
            import matplotlib.pyplot as plt
        except Exception:
            np.savez(path, time=self._time(), **self.series())
            print(f"matplotlib not available: wrote the logged series to {path}")
            return
        log, t = self.series(), self._time()
        panels = [("base_vel_x", "command_x", "Base velocity x [m/s]"), ("base_vel_y", "command_y", "Base velocity y [m/s]"),
                  ("base_vel_yaw", "command_yaw", "Base velocity yaw [rad/s]"), ("base_vel_z", None, "Base velocity z [m/s]"),
                  ("contact_forces_z", None, "Vertical contact forces [N]"), ("dof_torque", None, "Joint torque [Nm]")]
        fig, axs = plt.subplots(2, 3, figsize=(14, 7))
        for ax, (meas, cmd, title) in zip(axs.flat, panels):
            if meas in log:
                ax.plot(t[:len(log[meas])], log[meas], label="measured")
            if cmd and cmd in log:
                ax.plot(t[:len(log[cmd])], log[cmd], label="commanded")
            ax.set(xlabel="time [s]", title=title)
        fig.tight_layout()
        plt.show()

    def plot_dof_pos(self, path="play_dof_pos.npz"):
        try:
            import matplotlib.pyplot as plt
        except Exception:
            keep = {k: v for k, v in self.series().items() if k.startswith("dof_pos")}
            np.savez(path, time=self._time(), **keep)
            print(f"matplotlib not available: wrote the joint series to {path}")
            return
        log, t = self.series(), self._time()
        fig, axs = plt.subplots(4, 3, figsize=(12, 10))
        for j, ax in enumerate(axs.flat):
            if "dof_pos" in log:
                ax.plot(t[:len(log["dof_pos"])], log["dof_pos"][:, j], label="measured")
            if "dof_pos_target" in log:
                ax.plot(t[:len(log["dof_pos_target"])], log["dof_pos_target"][:, j], label="target")
            ax.set(xlabel="time [s]", ylabel="Position [rad]", title=f"DOF {j}")
        fig.tight_layout()
        plt.show()

    def mean_rewards(self):
        n = max(self.num_episodes, 1)
        return {k: float(np.sum(v)) / n for k, v in self.rew_log.items()}

    def print_rewards(self):
        print("Average rewards per second:")
        for key, mean in self.mean_rewards().items():
            print(f" - {key}: {mean}")
        print(f"Total number of episodes: {self.num_episodes}")
