"""Parameter schedules (epsilon, noise, PER beta, PPO clip rescaler).

Same class names, constructor arguments and per-step arithmetic as rl_coach/schedules.py:22-125 —
a schedule is advanced ONE increment per `step()` (repeated subtraction / multiplication, not a
closed form, so the float sequence matches the reference's) and clamped between its end points —
organised here as one base class with a per-schedule `_advance` rule.
"""


class Schedule(object):
    def __init__(self, initial_value, final_value=None):
        self.initial_value = initial_value
        self.final_value = initial_value if final_value is None else final_value
        self.current_value = initial_value
        self._lo = min(self.initial_value, self.final_value)
        self._hi = max(self.initial_value, self.final_value)

    def _advance(self, value):
        raise NotImplementedError("a schedule defines how its value moves in one step")

    def step(self):
        v = self._advance(self.current_value)
        if self._lo != self._hi:                      # clamp between the end points (np.clip in the reference)
            v = self._lo if v < self._lo else (self._hi if v > self._hi else v)
        self.current_value = v


class ConstantSchedule(Schedule):                         # schedules.py:31-36
    def __init__(self, initial_value):
        super().__init__(initial_value)

    def step(self):
        pass


class LinearSchedule(Schedule):                           # schedules.py:39-63
    """initial_value -> final_value in decay_steps equal decrements."""

    def __init__(self, initial_value, final_value, decay_steps):
        super().__init__(initial_value, final_value)
        self.decay_steps = decay_steps
        self.decay_delta = (initial_value - final_value) / float(decay_steps)

    def _advance(self, value):
        return value - self.decay_delta


class ExponentialSchedule(Schedule):                      # schedules.py:96-125
    """value *= decay_coefficient per step, stopping at final_value."""

    def __init__(self, initial_value, final_value, decay_coefficient):
        if decay_coefficient < 1 and final_value > initial_value:
            raise ValueError("The final value should be lower than the initial value when the decay coefficient < 1")
        if decay_coefficient > 1 and initial_value > final_value:
            raise ValueError("The final value should be higher than the initial value when the decay coefficient > 1")
        super().__init__(initial_value, final_value)
        self.decay_coefficient = decay_coefficient
        self.current_step = 0

    def _advance(self, value):
        self.current_step += 1
        return value * self.decay_coefficient
