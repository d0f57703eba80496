"""Parameter schedules — host-side mirror of rl_coach/schedules.py (same classes, same stepping
arithmetic; :22-125).  The hot path only reads ``current_value`` once per env-step / sample."""
import numpy as np


class Schedule(object):
    def __init__(self, initial_value):
        self.initial_value = initial_value
        self.current_value = initial_value

    def step(self):
        raise NotImplementedError("")


class ConstantSchedule(Schedule):                       # schedules.py:31-36
    def step(self):
        pass


class LinearSchedule(Schedule):                         # schedules.py:39-63
    def __init__(self, initial_value, final_value, decay_steps):
        super().__init__(initial_value)
        self.final_value = final_value
        self.decay_steps = decay_steps
        self.decay_delta = (initial_value - final_value) / float(decay_steps)

    def step(self):
        self.current_value -= self.decay_delta
        if self.final_value < self.initial_value:
            self.current_value = np.clip(self.current_value, self.final_value, self.initial_value)
        if self.final_value > self.initial_value:
            self.current_value = np.clip(self.current_value, self.initial_value, self.final_value)


class ExponentialSchedule(Schedule):                    # schedules.py:96-125
    def __init__(self, initial_value, final_value, decay_coefficient):
        super().__init__(initial_value)
        self.final_value = final_value
        self.decay_coefficient = decay_coefficient
        self.current_step = 0
        if decay_coefficient < 1 and final_value > initial_value:
            raise ValueError("The final value should be lower than the initial value when the decay coefficient < 1")
        if decay_coefficient > 1 and initial_value > final_value:
            raise ValueError("The final value should be higher than the initial value when the decay coefficient > 1")

    def step(self):
        self.current_value *= self.decay_coefficient
        if self.final_value < self.initial_value:
            self.current_value = np.clip(self.current_value, self.final_value, self.initial_value)
        if self.final_value > self.initial_value:
            self.current_value = np.clip(self.current_value, self.initial_value, self.final_value)
        self.current_step += 1
