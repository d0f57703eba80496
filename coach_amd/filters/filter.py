"""InputFilter — mirror of rl_coach/filters/filter.py:224-350 for device batches."""
from collections import OrderedDict


class InputFilter(object):
    def __init__(self, observation_filters=None, reward_filters=None, is_a_reference_filter=False):
        self._observation_filters = observation_filters or {}          # {key: OrderedDict(name -> filter)}
        self._reward_filters = reward_filters or OrderedDict()
        self.i_am_a_reference_filter = is_a_reference_filter

    def add_observation_filter(self, observation_name, filter_name, filter, add_as_the_first_filter=False):
        d = self._observation_filters.setdefault(observation_name, OrderedDict())
        d[filter_name] = filter
        if add_as_the_first_filter:
            d.move_to_end(filter_name, last=False)                     # filter.py:394-396

    def add_reward_filter(self, filter_name, filter, add_as_the_first_filter=False):
        self._reward_filters[filter_name] = filter
        if add_as_the_first_filter:
            self._reward_filters.move_to_end(filter_name, last=False)

    def filter_observation(self, key, observation, update_internal_state=True):
        """observation: device tensor with a leading batch (env) dimension."""
        for f in self._observation_filters.get(key, {}).values():
            observation = f.filter(observation, update_internal_state=update_internal_state)
        return observation

    def filter_reward(self, reward, out=None):
        for f in self._reward_filters.values():
            reward = f.filter(reward, out=out)
        return reward

    def reset(self):
        for d in self._observation_filters.values():
            for f in d.values():
                f.reset()


class NoInputFilter(InputFilter):                        # filter.py:353-358
    pass
